"""
The PACKED matrix-core kernel (``mfma_pack_kernel`` in csrc/mfma_scan.hip: two row tiles per f32 accumulator, folded as
packed f16 with v_pk_minimum3_f16) against the oracle and against the unpacked kernel.

What the packing could break, and what is therefore exercised here on the GPU:
  * the extremes of a dot product (-64 .. +64): rows and queries of all ones / all zeros / one bit, thresholds at both ends;
  * an all-zero 64-bit query (+64 would not fit the 7-bit high half): such a batch must stay on the unpacked kernel;
  * masked prefixes (codes of 1..7 bytes: the same kernel with a partial word);
  * every mode (single self-tightening pass, threshold levels, range-limited collect) and table sizes that are no
    multiple of the 128 rows a wave takes per step;
  * a self-tightening pass cut into stretches whose last one is shorter than ``mfma_min_rows``.
"""

import numpy as np
import pytest

from oracle import np_within, oracle_topk

pytestmark = pytest.mark.gpu

ONES = np.uint64(0xFFFFFFFFFFFFFFFF)


def _table(engine, rng, n, nbytes=8, extremes=True):
    mask = np.uint64((0xFFFFFFFFFFFFFFFF << (8 * (8 - nbytes))) & 0xFFFFFFFFFFFFFFFF)
    words = rng.integers(0, 2**64, size=(n, 1), dtype=np.uint64) & mask
    if extremes:
        words[rng.integers(0, n, size=8), 0] = ONES & mask
        words[rng.integers(0, n, size=8), 0] = np.uint64(0)
        words[rng.integers(0, n, size=8), 0] = np.uint64(1) << np.uint64(63)
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(5)
    t = engine.open_table(0, 1, nbytes)
    t.add(keys, words)
    return t, keys, words, mask


def _queries(rng, words, nq, mask, zero_query):
    q = rng.integers(0, 2**64, size=(nq, 1), dtype=np.uint64) & mask
    q[0, 0] = ONES & mask                       # dot products -popc .. 0
    q[1, 0] = (np.uint64(1) << np.uint64(63)) & mask
    q[2, 0] = (ONES ^ np.uint64(1) << np.uint64(63)) & mask
    q[3:13] = words[rng.integers(0, len(words), size=10)] ^ (np.uint64(9) << np.uint64(60)) & mask
    if zero_query:
        q[13, 0] = 0
    else:
        q[q[:, 0] == 0, 0] = np.uint64(1) << np.uint64(63)
    return q


def _expect(keys, words, q, k, nbytes):
    return oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=nbytes)


def _assert_equal(got, exp, what):
    for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
        np.testing.assert_array_equal(g, e, err_msg=f"{what}: {name}")


@pytest.fixture
def forced(hip_engine):
    hip_engine.set_option("mfma", 1)
    hip_engine.set_option("mfma_pack", 1)
    hip_engine.set_option("mfma_min_queries", 1)
    hip_engine.set_option("mfma_min_rows", 1)
    yield hip_engine
    hip_engine.set_option("mfma_min_queries", 17)
    hip_engine.set_option("mfma_min_rows", 65536)


# (65..128 queries: chunks of three and four groups, their own instantiations; 160: five groups, the general loop with an odd count;
#  2 M rows: every wave takes more than one stretch of four steps, so the accumulators carried across steps change hands)
@pytest.mark.parametrize("n,k,nq,nbytes", [(127, 5, 20, 8), (129, 10, 33, 8), (70_003, 10, 64, 8), (300_001, 100, 40, 8), (200_000, 10, 1024, 8),
                                           (50_000, 10, 48, 1), (50_000, 10, 48, 3), (50_000, 10, 48, 5), (90_001, 20, 48, 7),
                                           (150_001, 10, 70, 8), (150_001, 10, 96, 8), (150_001, 10, 128, 8), (100_003, 10, 160, 8),
                                           (2_000_003, 10, 32, 8), (2_000_003, 10, 64, 8), (2_000_003, 10, 96, 8), (2_000_003, 10, 128, 8)])
def test_packed_kernel_vs_oracle_and_unpacked(forced, n, k, nq, nbytes):
    rng = np.random.default_rng(4242 + n + nbytes)
    t, keys, words, mask = _table(forced, rng, n, nbytes)
    try:
        q = _queries(rng, words, nq, mask, zero_query=False)
        before = forced.stats()
        got = t.search(q, None, k)
        after = forced.stats()
        assert after["mfma_pack_launches"] > before["mfma_pack_launches"], "the batch did not run on the packed kernel"
        _assert_equal(got, _expect(keys, words, q, k, nbytes), "packed")
        forced.set_option("mfma_pack", 0)
        try:
            again = t.search(q, None, k)
            assert forced.stats()["mfma_pack_launches"] == after["mfma_pack_launches"]
        finally:
            forced.set_option("mfma_pack", 1)
        _assert_equal(again, got, "unpacked against packed")
        # the level design on the packed kernel
        forced.set_option("self_tighten", 0)
        try:
            levels = t.search(q, None, k)
        finally:
            forced.set_option("self_tighten", 1)
        _assert_equal(levels, got, "levels against the single pass")
    finally:
        t.drop()


def test_an_all_zero_64_bit_query_keeps_the_batch_off_the_packed_kernel(forced):
    rng = np.random.default_rng(808)
    t, keys, words, mask = _table(forced, rng, 80_000)
    try:
        q = _queries(rng, words, 40, mask, zero_query=True)
        before = forced.stats()
        got = t.search(q, None, 10)
        after = forced.stats()
        assert after["mfma_launches"] > before["mfma_launches"] and after["mfma_pack_launches"] == before["mfma_pack_launches"]
        _assert_equal(got, _expect(keys, words, q, 10, 8), "all-zero query")
        # a 40-bit table compares at most 40 bits: +40 fits, the packed kernel keeps the batch
        t5, keys5, words5, mask5 = _table(forced, rng, 80_000, nbytes=5)
        try:
            q5 = _queries(rng, words5, 40, mask5, zero_query=True)
            got5 = t5.search(q5, None, 10)
            assert forced.stats()["mfma_pack_launches"] > after["mfma_pack_launches"]
            _assert_equal(got5, _expect(keys5, words5, q5, 10, 5), "all-zero 40-bit query")
        finally:
            t5.drop()
    finally:
        t.drop()


@pytest.mark.parametrize("radius,nq", [(0, 24), (1, 24), (12, 24), (31, 24), (32, 24), (63, 24), (64, 24), (1, 50), (12, 80), (14, 128), (32, 100), (12, 200)])
def test_range_limited_searches_on_the_packed_kernel(forced, radius, nq):
    """Collect mode under a GIVEN threshold, up to the radius that admits every row (thr = 64 - popc(q): both ends of a half)."""
    rng = np.random.default_rng(99 + radius + nq)
    n = 3_000 if radius >= 31 else 150_000
    t, keys, words, mask = _table(forced, rng, n)
    try:
        q = _queries(rng, words, nq, mask, zero_query=False)
        k = 4096 if radius >= 31 else 64
        before = forced.stats()
        gk, gh, gp, gc = t.search_within(q, None, k, radius)
        assert forced.stats()["mfma_pack_launches"] > before["mfma_pack_launches"]
        for i in range(len(q)):
            ek, eh, _ = np_within(words, 8, keys, q[i], 8, k, radius)     # one query: (keys, hamming, prefix bits) of the <= k hits
            assert int(gc[i]) == len(ek), (i, gc[i], len(ek))
            np.testing.assert_array_equal(gk[i, : len(ek)], ek)
            np.testing.assert_array_equal(gh[i, : len(ek)], eh)
    finally:
        t.drop()


def test_self_pass_in_stretches_with_a_short_last_one(hip_engine):
    """
    ADVICE r2: with several chunks of queries the single pass walks cache-sized stretches; a last stretch shorter than
    ``mfma_min_rows`` used to fall through to the XOR + popcount launcher, which has no self-tightening mode.
    256-bit codes (chunks of 256 queries), 1 MB stretches of 4 x 8 192 rows ... and a table that leaves a 24 k-row tail.
    """
    rng = np.random.default_rng(31337)
    n, nq, k, nbytes = 3 * 32_768 * 3 + 24_000, 600, 10, 32
    words = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(1)
    q = rng.integers(0, 2**64, size=(nq, 4), dtype=np.uint64)
    q[:50] = words[rng.integers(0, n, size=50)]
    t = hip_engine.open_table(0, 1, nbytes)
    try:
        t.add(keys, words)
        hip_engine.set_option("stretch_mb", 1)
        hip_engine.set_option("mfma_stretch_factor", 1)
        try:
            before = hip_engine.stats()
            got = t.search(q, None, k)
            after = hip_engine.stats()
        finally:
            hip_engine.set_option("stretch_mb", 128)
            hip_engine.set_option("mfma_stretch_factor", 3)
        launches = after["scan_launches"] - before["scan_launches"]
        assert launches > 2, "the pass was expected to run in several stretches"
        assert after["scan_mfma_launches"] - before["scan_mfma_launches"] == launches, "a stretch of the single pass left the matrix cores"
        assert after["self_retries"] == before["self_retries"] and after["fallback_queries"] == before["fallback_queries"]
        _assert_equal(got, oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=nbytes), "stretched single pass")
    finally:
        t.drop()
