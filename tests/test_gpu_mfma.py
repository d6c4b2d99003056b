"""
The matrix-core (FP4 MFMA) form of the scan (``csrc/mfma_scan.hip``) against the oracle.

By default the engine sends a launch to the matrix cores only for batches of > 16 queries over >= 65 536 rows, which
most parity cases are too small to reach.  Here the thresholds are dropped to 1 query / 1 row so that EVERY scan launch
of the re-run parity, range-limited, fuzz and search_many cases takes that path (all code lengths 1..32 bytes, masked
prefixes, mixed-length NPHD segments, 64- and 128-bit keys, tails that are not a multiple of 64 rows, overflow
fallback), and a large-batch case runs under the default thresholds.
"""

import numpy as np
import pytest

import test_gpu_fuzz as fuzz
import test_gpu_many as many
import test_gpu_parity as parity
import test_gpu_within as within
from oracle import oracle_topk

pytestmark = pytest.mark.gpu


@pytest.fixture
def forced(hip_engine):
    """Every scan launch on the matrix cores; restored afterwards."""
    hip_engine.set_option("mfma", 1)
    hip_engine.set_option("mfma_min_queries", 1)
    hip_engine.set_option("mfma_min_rows", 1)
    before = hip_engine.stats()["mfma_launches"]
    yield hip_engine
    ran = hip_engine.stats()["mfma_launches"] - before
    hip_engine.set_option("mfma_min_queries", 17)
    hip_engine.set_option("mfma_min_rows", 65536)
    assert ran > 0, "the case never reached the MFMA kernel"


@pytest.mark.parametrize("n,k,nq", [(100, 10, 5), (5000, 10, 33), (70000, 10, 40), (300000, 100, 17), (200001, 10, 300)])
def test_hamming64_random(forced, n, k, nq):
    parity.test_hamming64_random_vs_oracle(forced, n, k, nq)


def test_hamming64_heavy_ties(forced):
    parity.test_hamming64_heavy_ties(forced)


@pytest.mark.parametrize("key_words", [1, 2])
def test_identical_codes_overflow_fallback(forced, key_words):
    parity.test_all_identical_codes_take_the_exact_fallback(forced, key_words)


@pytest.mark.parametrize("nbytes", [1, 4, 8, 16, 24, 32, 13])
def test_fixed_lengths_128bit_keys(forced, nbytes):
    parity.test_hamming_fixed_lengths_128bit_keys(forced, nbytes)


def test_nphd_mixed_lengths(forced):
    parity.test_nphd_mixed_lengths_vs_oracle(forced)


def test_golden_kats(forced):
    parity.test_golden_hamming_kats(forced)


@pytest.mark.parametrize("nbytes,key_words", [(8, 1), (8, 2), (16, 2), (32, 2), (13, 1)])
def test_within_fixed_length(forced, nbytes, key_words):
    within.test_within_fixed_length_vs_numpy(forced, nbytes, key_words)


def test_within_nphd_prefix_match(forced):
    within.test_within_nphd_mixed_lengths_is_the_prefix_match(forced)


def test_within_overflow_fallback(forced):
    within.test_within_radius_beyond_the_candidate_buffer_takes_the_exact_fallback(forced)


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_sequences(forced, seed):
    fuzz.test_random_operation_sequences(forced, seed)


def test_search_many(forced):
    many.test_search_many_equals_the_oracle_in_every_order(forced)


@pytest.mark.parametrize("k,key_words", [(4096, 1), (1000, 2)])
def test_largest_k_on_the_matrix_cores(forced, k, key_words):
    """The engine's k ceiling (simprint-sized result lists): candidate handling and select under the matrix-core scan."""
    rng = np.random.default_rng(k)
    n, nq = 60_000, 40
    words = rng.integers(0, 2**64, size=(n, 2), dtype=np.uint64)
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(1)
    if key_words == 2:
        keys = np.stack([rng.integers(1, 500, size=n).astype(np.uint64), keys], axis=1)
    q = words[rng.integers(0, n, size=nq)] ^ np.uint64(0x11)
    t = forced.open_table(0, key_words, 16)
    try:
        t.add(keys, words)
        got = t.search(q, None, k)
        exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=16)
        for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
            np.testing.assert_array_equal(g, e, err_msg=name)
    finally:
        t.drop()


@pytest.mark.parametrize("nbytes,metric", [(8, 0), (32, 1), (20, 0)])
def test_large_batch_under_default_thresholds(hip_engine, nbytes, metric):
    """1 000 queries over 400 003 rows: levels and collect pass run on the matrix cores, bit-exact against the oracle."""
    rng = np.random.default_rng(31 + nbytes)
    n, nq, k = 400_003, 1000, 10
    mw = (nbytes + 7) // 8
    words = parity._rand_words(rng, n, mw, nbytes)
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(3)
    q = parity._rand_words(rng, nq, mw, nbytes)
    near = rng.integers(0, n, size=nq // 2)
    q[: nq // 2] = words[near]
    q[: nq // 2, 0] ^= np.uint64(1) << rng.integers(40, 64, size=nq // 2).astype(np.uint64)      # one bit off a stored code
    lens = np.full(n, nbytes, dtype=np.uint8) if metric else None
    qlens = np.full(nq, nbytes, dtype=np.uint8) if metric else None
    t = hip_engine.open_table(metric, 1, nbytes)
    try:
        t.add(keys, words, lens)
        before = hip_engine.stats()["mfma_launches"]
        got = t.search(q, qlens, k)
        assert hip_engine.stats()["mfma_launches"] > before
        exp = oracle_topk(metric, keys, words, lens, q, qlens, k, fixed_nbytes=0 if metric else nbytes)
        for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
            np.testing.assert_array_equal(g, e, err_msg=name)
        # the XOR + popcount kernel returns the same bits
        hip_engine.set_option("mfma", 0)
        try:
            other = t.search(q, qlens, k)
        finally:
            hip_engine.set_option("mfma", 1)
        for g, e, name in zip(other, exp, ("keys", "hamming", "prefix_bits", "count")):
            np.testing.assert_array_equal(g, e, err_msg="valu " + name)
    finally:
        t.drop()


# ---------------------------------------------------------------------------------------------------------------------
# k <= 512 on the matrix cores is ONE pass with self-tightening thresholds (MODE_SELF); larger k and the option
# self_tighten=0 take threshold levels + picks.  Everything above ran the default; the level design stays covered here, and
# the two must agree with each other and with the oracle where the single pass is most exposed: clustered codes (big tie
# classes, thresholds that collapse to 0 within the first steps), planted duplicates, k at both ends of its range.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture
def forced_levels(forced):
    forced.set_option("self_tighten", 0)
    yield forced
    forced.set_option("self_tighten", 1)


@pytest.mark.parametrize("n,k,nq", [(70000, 10, 40), (300000, 100, 17)])
def test_hamming64_random_with_levels(forced_levels, n, k, nq):
    parity.test_hamming64_random_vs_oracle(forced_levels, n, k, nq)


def test_nphd_mixed_lengths_with_levels(forced_levels):
    parity.test_nphd_mixed_lengths_vs_oracle(forced_levels)


@pytest.mark.parametrize("seed", range(2))
def test_fuzz_sequences_with_levels(forced_levels, seed):
    fuzz.test_random_operation_sequences(forced_levels, seed)


@pytest.mark.parametrize("nbytes,k", [(8, 1), (8, 10), (8, 512), (16, 100), (29, 37)])
def test_self_tightening_pass_equals_levels_and_oracle(hip_engine, nbytes, k):
    rng = np.random.default_rng(977 + nbytes + k)
    n, nq = 1_000_000 + 77, 96   # beyond 8 x the 65 536-row bootstrap sample: the level design needs a level before its collect pass
    mw = (nbytes + 7) // 8
    # a third of the rows are noisy copies of 12 base codes (0-3 flipped bits), the rest random
    words = parity._rand_words(rng, n, mw, nbytes)
    bases = parity._rand_words(rng, 12, mw, nbytes)
    clustered = rng.random(n) < 0.33
    pick = rng.integers(0, 12, size=n)
    noisy = bases[pick].copy()
    for _ in range(3):      # up to three flipped bits per copy (a flip beyond the code's length is masked away again)
        noisy[:, 0] ^= (rng.random(n) < 0.5).astype(np.uint64) << rng.integers(0, 64, size=n).astype(np.uint64)
    words[clustered] = noisy[clustered]
    words = fuzz._mask(words, nbytes)
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(11)
    q = np.concatenate([bases, words[rng.integers(0, n, size=nq - 12 - 30)], parity._rand_words(rng, 30, mw, nbytes)])
    t = hip_engine.open_table(0, 1, nbytes)
    try:
        t.add(keys, words)
        before = hip_engine.stats()
        single_pass = t.search(q, None, k)
        after = hip_engine.stats()
        assert after["mfma_launches"] > before["mfma_launches"] and after["level_launches"] == before["level_launches"], "not the single pass"
        hip_engine.set_option("self_tighten", 0)
        try:
            levels = t.search(q, None, k)
            assert hip_engine.stats()["level_launches"] > after["level_launches"], "not the level design"
        finally:
            hip_engine.set_option("self_tighten", 1)
        exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=nbytes)
        for g, l, e, name in zip(single_pass, levels, exp, ("keys", "hamming", "prefix_bits", "count")):
            np.testing.assert_array_equal(g, e, err_msg="single pass: " + name)
            np.testing.assert_array_equal(l, e, err_msg="levels: " + name)
    finally:
        t.drop()


def test_an_overflowed_single_pass_is_answered_again_by_the_levels(hip_engine):
    """
    The single pass never prunes its candidate lists (~k ln(n / sample) entries + ties + the first steps' flood).  With the
    buffer shrunk to 640 entries and a 256-row bootstrap sample (every wave starts under a threshold that ~4 % of the rows
    pass) it overflows where the level design, which prunes after every level, still fits: the batch must come back exact,
    through ONE retry and without the per-query exact fallback.
    """
    rng = np.random.default_rng(5150)
    n, nq, k = 300_000, 64, 10
    words = rng.integers(0, 2**64, size=(n, 1), dtype=np.uint64)
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(1)
    q = rng.integers(0, 2**64, size=(nq, 1), dtype=np.uint64)
    t = hip_engine.open_table(0, 1, 8)
    try:
        t.add(keys, words)
        hip_engine.set_option("candidate_cap", 256)
        hip_engine.set_option("self_boot_rows", 256)
        hip_engine.set_option("self_boot_per_k", 0)         # (the sample otherwise grows with k)
        try:
            before = hip_engine.stats()
            got = t.search(q, None, k)
            after = hip_engine.stats()
        finally:
            hip_engine.set_option("candidate_cap", 16384)
            hip_engine.set_option("self_boot_rows", 65536)
            hip_engine.set_option("self_boot_per_k", 1024)
        delta = {x: after[x] - before[x] for x in ("self_retries", "fallback_queries", "mfma_launches", "mfma_pack_launches", "level_launches", "scan_launches")}
        assert after["self_retries"] == before["self_retries"] + 1, f"the single pass was expected to overflow its lists: {delta}"
        assert after["fallback_queries"] == before["fallback_queries"], "the level design was expected to fit its lists"
        exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=8)
        for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
            np.testing.assert_array_equal(g, e, err_msg=name)
    finally:
        t.drop()
