"""
``isccsearch_search_many``: several searches, one synchronisation.  Every request must return exactly what the
ORACLE returns for it (an ``OracleTable`` mirror holds the same rows as each device table), whichever internal
path it takes: deferred (single-segment table), ordinary (multi-segment / mixed query lengths / empty table), or
the exact fallback after a candidate-list overflow -- and in whichever ORDER the paths are mixed: the ordinary
pipeline stages its results in the same pinned block the deferred requests were copied into.
"""

import os

import numpy as np
import pytest

from oracle_engine import OracleTable
from test_gpu_parity import METRIC_HAMMING, METRIC_NPHD, _mask_to_len, _rand_words

# the whole GPU tier can be re-run with the matrix cores off (ISCC_HIP_OPTS="mfma=0", tests/conftest.py)
MFMA_ON = "mfma=0" not in os.environ.get("ISCC_HIP_OPTS", "")
needs_matrix_cores = pytest.mark.skipif(not MFMA_ON, reason="asserts what the matrix-core kernels do (single pass, hinted start, packed launches)")

pytestmark = pytest.mark.gpu


def _same(a, b, tag):
    for x, y, name in zip(a, b, ("keys", "hamming", "prefix_bits", "count")):
        np.testing.assert_array_equal(x, y, err_msg=f"{tag}: {name}")


class Mirrored:
    """A device table and an oracle table holding the same rows."""

    def __init__(self, engine, metric, key_words, max_bytes):
        self.hip = engine.open_table(metric, key_words, max_bytes)
        self.ref = OracleTable(metric, key_words, max_bytes)

    def add(self, keys, words, nbytes=None):
        self.hip.add(keys, words, nbytes)
        self.ref.add(keys, words, nbytes)

    def expect(self, q, nb, k, radius):
        return self.ref.search(q, nb, k) if radius is None else self.ref.search_within(q, nb, k, radius)

    def drop(self):
        self.hip.drop()


def _run_and_check(engine, plan, tag):
    got = engine.search_many([(m.hip, q, nb, k, r) for m, q, nb, k, r in plan])
    for i, ((m, q, nb, k, r), g) in enumerate(zip(plan, got)):
        _same(g, m.expect(q, nb, k, r), f"{tag} request {i}")
    return got


def test_search_many_equals_the_oracle_in_every_order(hip_engine):
    rng = np.random.default_rng(77)
    t1 = Mirrored(hip_engine, METRIC_HAMMING, 1, 8)
    t2 = Mirrored(hip_engine, METRIC_NPHD, 1, 32)
    t3 = Mirrored(hip_engine, METRIC_HAMMING, 2, 16)
    t4 = Mirrored(hip_engine, METRIC_NPHD, 1, 32)      # one segment only: deferred even though it is an NPHD table
    empty = Mirrored(hip_engine, METRIC_HAMMING, 1, 8)
    try:
        n = 30000
        w1 = rng.integers(0, 2**64, size=(n, 1), dtype=np.uint64)
        w1[:400] = np.uint64(0xABCDEF0123456789)
        t1.add(rng.permutation(n).astype(np.uint64) + np.uint64(1), w1)
        lens = rng.choice([8, 16, 32], size=n).astype(np.uint8)
        w2 = _rand_words(rng, n, 4, lens)
        t2.add(np.arange(1, n + 1, dtype=np.uint64), w2, lens)
        w3 = rng.integers(0, 4, size=(n, 2), dtype=np.uint64) * np.uint64(0x0F0F0F0F0F0F0F0F)
        t3.add(np.stack([rng.integers(1, 50, size=n).astype(np.uint64), rng.permutation(n).astype(np.uint64)], axis=1), w3)
        w4 = _rand_words(rng, n, 4, 32)
        t4.add(np.arange(1, n + 1, dtype=np.uint64), w4, np.full(n, 32, dtype=np.uint8))

        q1 = np.concatenate([w1[:3], rng.integers(0, 2**64, size=(2, 1), dtype=np.uint64)])
        q2 = _mask_to_len(w2[[5, 77, 1234]].copy(), np.array([16, 16, 16], dtype=np.uint8))
        q2_mixed = _mask_to_len(w2[[5, 77]].copy(), np.array([8, 32], dtype=np.uint8))
        q3 = w3[:4].copy()
        q4 = w4[[9, 99]].copy()
        q4[:, 3] ^= np.uint64(7)
        nb16, nbmix, nb32 = np.full(3, 16, np.uint8), np.array([8, 32], np.uint8), np.full(2, 32, np.uint8)
        requests = [
            (t1, q1, None, 10, None),          # deferred
            (t2, q2, nb16, 7, None),           # multi-segment table: ordinary path
            (t3, q3, None, 1000, 0),           # deferred, range-limited, 128-bit keys
            (t1, q1[:2], None, 3, 2),          # deferred, same table again
            (t2, q2_mixed, nbmix, 5, None),    # mixed query lengths: ordinary path
            (empty, q1[:1], None, 4, None),    # empty table
            (t4, q4, nb32, 20, None),          # single-segment NPHD table: deferred
            (t1, w1[:1], None, 10, 0),         # 400 collisions asked for 10: still fits the candidate list
        ]
        _run_and_check(hip_engine, requests, "mixed")
        # the orders ADVICE r1 names: an ordinary (multi-segment) request AHEAD of a deferred one, and a large
        # ordinary request (nq * k records from offset 0 of the staging block) between two small deferred ones
        _run_and_check(hip_engine, [requests[1], requests[0]], "ordinary-then-deferred")
        _run_and_check(hip_engine, [requests[4], requests[6], requests[1], requests[2]], "ordinary-deferred-ordinary-deferred")
        q2_many = _mask_to_len(w2[rng.integers(0, n, size=40)].copy(), np.full(40, 16, dtype=np.uint8))
        big_ordinary = (t2, q2_many, np.full(40, 16, np.uint8), 300, None)
        _run_and_check(hip_engine, [(t1, q1[:1], None, 3, None), big_ordinary, (t4, q4, nb32, 5, None), (t1, q1, None, 10, None)],
                       "deferred-small / ordinary-large / deferred")
        # search_assets' own shape: a mixed-length META table ahead of a single-length CONTENT table (index.py::_search_units)
        _run_and_check(hip_engine, [(t2, q2_mixed[:1], nbmix[:1], 100, None), (t4, q4[:1], nb32[:1], 100, None), (t1, q1[:1], None, 4096, 0)],
                       "search_assets shape")
        assert hip_engine.search_many([]) == []
        with pytest.raises(ValueError):
            hip_engine.search_many([(t1.hip, q1, None, 0, None)])
    finally:
        for t in (t1, t2, t3, t4, empty):
            t.drop()


def test_search_many_overflow_takes_the_exact_fallback(hip_engine):
    n = 120000
    t = Mirrored(hip_engine, METRIC_HAMMING, 1, 8)
    small = Mirrored(hip_engine, METRIC_HAMMING, 1, 8)
    try:
        words = np.full((n, 1), 0x1111111111111111, dtype=np.uint64)
        words[::3] ^= np.uint64(1)
        t.add(np.arange(n, dtype=np.uint64)[::-1].copy() + np.uint64(5), words)
        rng = np.random.default_rng(4)
        sw = rng.integers(0, 2**64, size=(5000, 1), dtype=np.uint64)
        small.add(np.arange(5000, dtype=np.uint64) + np.uint64(1), sw)
        q = np.array([[0x1111111111111111], [0x1111111111111110], [0x7777777777777777]], dtype=np.uint64)
        before = hip_engine.stats()["fallback_queries"]
        # the overflowed requests rerun through the ordinary path AFTER every deferred result has been handed out,
        # the healthy deferred request behind them included
        _run_and_check(hip_engine, [(t, q, None, 10, None), (t, q, None, 50, 0), (t, q[:1], None, 5, None), (small, sw[:4], None, 7, None)], "overflow")
        assert hip_engine.stats()["fallback_queries"] > before
    finally:
        t.drop()
        small.drop()


def test_small_batches_speculate_on_the_previous_k_th_distance_and_stay_exact(hip_engine):
    """
    A batch of a few queries over a one-segment table is first tried as ONE range-limited pass under the k-th distance the
    previous such search ended at (+ 2) and verified (`spec_hits` / `spec_misses`): a hit when the radius holds k rows for every
    query, the ordinary path when it does not (a query far from everything after queries inside a cluster) or when a list
    overflows (a cluster of 40 000 near-duplicates inside the radius).  Every answer equals the oracle's.
    """
    from oracle import oracle_topk

    rng = np.random.default_rng(77)
    n, k = 400_000, 10
    words = rng.integers(0, 2**64, size=(n, 1), dtype=np.uint64)
    small, big = np.uint64(0x0123456789ABCDEF), np.uint64(0xFEDCBA9876543210)
    picks = rng.choice(n, size=43_000, replace=False)
    words[picks[:3_000], 0] = small ^ (np.uint64(1) << rng.integers(0, 64, size=3_000).astype(np.uint64))     # 3 000 rows 1 bit off `small`
    words[picks[3_000:], 0] = big ^ (np.uint64(1) << rng.integers(0, 64, size=40_000).astype(np.uint64))      # 40 000 rows 1 bit off `big`
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(1)
    t = hip_engine.open_table(0, 1, 8)
    try:
        t.add(keys, words)

        def ask(q):
            q = np.asarray(q, dtype=np.uint64).reshape(-1, 1)
            before = hip_engine.stats()
            got = t.search(q, None, k)
            after = hip_engine.stats()
            exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=8)
            for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
                np.testing.assert_array_equal(g, e, err_msg=name)
            return after["spec_hits"] - before["spec_hits"], after["spec_misses"] - before["spec_misses"], int(got[1].max())

        r = rng.integers(0, 2**64, size=16, dtype=np.uint64)
        assert ask(r[:1])[:2] == (0, 0)                              # nothing to go by yet
        hits, misses, worst = ask(r[1:2])
        assert (hits, misses) == (1, 0) and worst >= 10              # the previous search's k-th distance + 2 holds this one's
        assert ask(r[2:6])[:2] == (0, 0)                             # a batch of four: another size class, its own hint
        assert ask(r[2:6])[:2] == (1, 0)
        assert ask([small])[:2] == (1, 0)                            # 3 000 near-duplicates inside a random query's radius: they fit
        assert ask(r[6:7])[:2] == (1, 0)                             # ... and the hint decayed by ONE bit, not to their distance
        assert ask([big])[:2] == (0, 1)                              # 40 000 rows within the radius: the list overflows -> ordinary path
        assert ask(r[7:8])[:2] == (0, 1)                             # `big` ended at distance 1: radius 3 holds nothing of a random query
        assert ask(r[8:9])[:2] == (0, 0)                             # two misses in a row: the next two batches do not speculate
        assert ask(r[9:10])[:2] == (0, 0)
        assert ask(r[10:11])[:2] == (1, 0)
        assert ask([small ^ np.uint64(3)])[:2] == (1, 0)             # the same 3 000 rows at distance 1 or 3
        # the per-unit searches of one search_assets request (isccsearch_search_many) speculate the same way, verified after their
        # one synchronisation; a request whose pass missed is rerun on the ordinary path (which does not speculate again)
        def ask_many(qs):
            reqs = [(t, np.asarray([q], dtype=np.uint64).reshape(1, 1), None, k, None) for q in qs]
            before = hip_engine.stats()
            outs = hip_engine.search_many(reqs)
            after = hip_engine.stats()
            for (_, q, _, _, _), got in zip(reqs, outs):
                exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=8)
                for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
                    np.testing.assert_array_equal(g, e, err_msg=name)
            return after["spec_hits"] - before["spec_hits"], after["spec_misses"] - before["spec_misses"]

        assert ask_many(r[:2]) == (2, 0)
        assert ask_many([small]) == (1, 0)
        assert ask_many([big]) == (0, 1)                             # overflow; the rerun seeds radius 3
        assert ask_many(r[2:4]) == (0, 2)                            # both were enqueued under radius 3: misses two and three in a row
        assert ask_many(r[4:8]) == (0, 0)                            # ... six batches of back-off, four of them here
        assert ask_many(r[8:10]) == (0, 0)
        assert ask_many(r[10:12]) == (2, 0)
        hip_engine.set_option("speculate", 0)
        try:
            assert ask(r[:3])[:2] == (0, 0)
        finally:
            hip_engine.set_option("speculate", 1)
    finally:
        t.drop()


@needs_matrix_cores
@pytest.mark.parametrize("nq", [17, 32, 64, 100, 128])
def test_matrix_core_batches_speculate_too(hip_engine, nq):
    """
    Up to ``spec_max_queries`` (128) queries: the speculative pass of a batch of 9 or more runs on the packed matrix-core kernel
    in collect mode (chunks of one to four groups).  Hit, miss (one query inside a cluster that overflows its list) and the
    batch one above the limit (ordinary path), each against the oracle.
    """
    from oracle import oracle_topk

    rng = np.random.default_rng(1000 + nq)
    n, k = 300_000, 10
    words = rng.integers(0, 2**64, size=(n, 1), dtype=np.uint64)
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(1)
    near = words[rng.integers(0, n, size=4 * nq), 0] ^ (np.uint64(1) << rng.integers(0, 64, size=4 * nq).astype(np.uint64))
    # k + 2 copies near every `near` query: their k-th distance is <= 2
    extra = np.repeat(near, k + 2) ^ (np.uint64(1) << rng.integers(0, 64, size=4 * nq * (k + 2)).astype(np.uint64))
    # and a cluster of 30 000 rows one bit off `centre`: more than a candidate list holds
    centre = np.uint64(0x5A5A5A5A12345678)
    cluster = centre ^ (np.uint64(1) << rng.integers(0, 64, size=30_000).astype(np.uint64))
    words = np.concatenate([words, extra.reshape(-1, 1), cluster.reshape(-1, 1)])
    keys = np.concatenate([keys, np.arange(len(extra) + len(cluster), dtype=np.uint64) + np.uint64(n + 1)])
    t = hip_engine.open_table(0, 1, 8)
    try:
        t.add(keys, words)

        def ask(q):
            q = np.ascontiguousarray(np.asarray(q, dtype=np.uint64).reshape(-1, 1))
            before = hip_engine.stats()
            got = t.search(q, None, k)
            after = hip_engine.stats()
            exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=8)
            for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
                np.testing.assert_array_equal(g, e, err_msg=name)
            return after["spec_hits"] - before["spec_hits"], after["spec_misses"] - before["spec_misses"], after["mfma_pack_launches"] - before["mfma_pack_launches"]

        random_q = rng.integers(1, 2**64, size=3 * nq, dtype=np.uint64)
        assert ask(random_q[:nq])[:2] == (0, 0)                      # the first search of this size class: nothing to go by
        hits, misses, packed = ask(random_q[nq : 2 * nq])
        assert (hits, misses) == (1, 0) and packed >= 1              # random queries after random queries: one packed collect pass
        assert ask(near[:nq])[:2] == (1, 0)                          # all within the radius, far below it
        mixed = random_q[:nq].copy()
        mixed[nq // 2] = centre                                      # 30 000 rows within the radius of one query: its list overflows
        assert ask(mixed)[:2] == (0, 1)
        assert ask(random_q[2 * nq : 3 * nq])[:2] == (1, 0)          # the ordinary rerun re-seeded the radius; one miss does not back off
        hip_engine.set_option("spec_max_queries", nq - 1)            # one above the limit: the single pass, started under the hint
        try:
            assert ask(random_q[:nq])[:2] == (1, 0)
            hip_engine.set_option("self_hint", 0)
            assert ask(random_q[:nq])[:2] == (0, 0)                  # ... or under its bootstrap sample
        finally:
            hip_engine.set_option("spec_max_queries", 128)
            hip_engine.set_option("self_hint", 1)
    finally:
        t.drop()


@needs_matrix_cores
@pytest.mark.parametrize("nq,k", [(200, 10), (1024, 10), (300, 100)])
def test_large_batches_start_their_single_pass_under_the_hint(hip_engine, nq, k):
    """
    Batches above ``spec_max_queries`` keep the single self-tightening pass but start it under the k-th distance the previous
    batch of their size ended at (+ 2) instead of a bootstrap sample (option ``self_hint``).  Verified like the speculative pass:
    a hint that is too tight for some query (random queries after near-duplicates) or a list that overflows (a query inside a
    30 000-row cluster) sends the batch through the ordinary pass.  Every answer equals the oracle's.
    """
    from oracle import oracle_topk

    rng = np.random.default_rng(5000 + nq + k)
    n = 300_000
    words = rng.integers(0, 2**64, size=(n, 1), dtype=np.uint64)
    near = words[rng.integers(0, n, size=nq), 0] ^ (np.uint64(1) << rng.integers(0, 64, size=nq).astype(np.uint64))
    extra = np.repeat(near, k + 2) ^ (np.uint64(1) << rng.integers(0, 64, size=nq * (k + 2)).astype(np.uint64))      # k-th distance of `near` <= 2
    centre = np.uint64(0x5A5A5A5A12345678)
    cluster = centre ^ (np.uint64(1) << rng.integers(0, 64, size=30_000).astype(np.uint64))
    words = np.concatenate([words, extra.reshape(-1, 1), cluster.reshape(-1, 1)])
    keys = rng.permutation(len(words)).astype(np.uint64) + np.uint64(1)
    t = hip_engine.open_table(0, 1, 8)
    try:
        t.add(keys, words)

        def ask(q):
            q = np.ascontiguousarray(np.asarray(q, dtype=np.uint64).reshape(-1, 1))
            before = hip_engine.stats()
            got = t.search(q, None, k)
            after = hip_engine.stats()
            exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=8)
            for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
                np.testing.assert_array_equal(g, e, err_msg=name)
            return after["spec_hits"] - before["spec_hits"], after["spec_misses"] - before["spec_misses"]

        r = rng.integers(1, 2**64, size=8 * nq, dtype=np.uint64)
        far = np.array([bin(int(x) ^ int(centre)).count("1") >= 28 for x in r])      # (a random query NEAR the cluster would overflow by itself)
        r = r[far][: 4 * nq]
        assert len(r) == 4 * nq
        assert ask(r[:nq]) == (0, 0)                                 # bootstrap sample; seeds the hint
        assert ask(r[nq : 2 * nq]) == (1, 0)                         # started under the hint
        with_centre = r[2 * nq : 3 * nq].copy()
        with_centre[7] = centre
        assert ask(with_centre) == (0, 1)                            # a list overflows under the hint -> ordinary pass (whose own retry repairs it)
        assert ask(r[:nq]) == (1, 0)                                 # (one miss does not back off)
        assert ask(near) == (1, 0)                                   # near-duplicates: far inside the hint, which decays by one bit
        hip_engine.set_option("self_hint", 0)
        assert ask(near) == (0, 0)
        hip_engine.set_option("self_hint", 1)
    finally:
        hip_engine.set_option("self_hint", 1)
        t.drop()


@needs_matrix_cores
def test_a_hint_that_is_too_tight_is_noticed(hip_engine):
    """Near-duplicate queries seed a hint of a few bits; the random queries that follow find fewer than k rows under it: miss, rerun, exact."""
    from oracle import oracle_topk

    rng = np.random.default_rng(909)
    n, nq, k = 200_000, 200, 10
    words = rng.integers(0, 2**64, size=(n, 1), dtype=np.uint64)
    near = words[rng.integers(0, n, size=nq), 0]
    extra = np.repeat(near, k + 2) ^ (np.uint64(1) << rng.integers(0, 64, size=nq * (k + 2)).astype(np.uint64))
    words = np.concatenate([words, extra.reshape(-1, 1)])
    keys = rng.permutation(len(words)).astype(np.uint64) + np.uint64(1)
    t = hip_engine.open_table(0, 1, 8)
    try:
        t.add(keys, words)

        def ask(q):
            q = np.ascontiguousarray(np.asarray(q, dtype=np.uint64).reshape(-1, 1))
            before = hip_engine.stats()
            got = t.search(q, None, k)
            after = hip_engine.stats()
            exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=8)
            for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
                np.testing.assert_array_equal(g, e, err_msg=name)
            return after["spec_hits"] - before["spec_hits"], after["spec_misses"] - before["spec_misses"]

        assert ask(near) == (0, 0)                                   # seeds the hint at <= 1 + 2 bits
        assert ask(rng.integers(1, 2**64, size=nq, dtype=np.uint64)) == (0, 1)
        assert ask(rng.integers(1, 2**64, size=nq, dtype=np.uint64)) == (1, 0)
    finally:
        t.drop()


@needs_matrix_cores
@pytest.mark.parametrize("nbytes,n,nq,k", [(8, 150_000, 150, 700), (8, 120_000, 40, 4096), (16, 100_000, 150, 2048), (32, 80_000, 64, 1000)])
def test_the_single_pass_serves_every_k(hip_engine, nbytes, n, nq, k):
    """``self_max_k`` = 4 096: large k takes the single self-tightening pass too (bootstrap sample, then under the hint); both against the oracle."""
    from oracle import oracle_topk

    rng = np.random.default_rng(31 + nbytes + k)
    w = (nbytes + 7) // 8
    words = rng.integers(0, 2**64, size=(n, w), dtype=np.uint64)
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(1)
    t = hip_engine.open_table(0, 1, nbytes)
    try:
        t.add(keys, words)
        for rnd in range(3):
            q = rng.integers(0, 2**64, size=(nq, w), dtype=np.uint64)
            q[: nq // 4] = words[rng.integers(0, n, size=nq // 4)]
            before = hip_engine.stats()
            got = t.search(q, None, k)
            after = hip_engine.stats()
            assert after["level_launches"] == before["level_launches"], "large k fell back to the level design"
            assert (after["spec_hits"] + after["spec_misses"] > before["spec_hits"] + before["spec_misses"]) == (rnd > 0)
            exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=nbytes)
            for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
                np.testing.assert_array_equal(g, e, err_msg=f"round {rnd}: {name}")
    finally:
        t.drop()


@needs_matrix_cores
def test_hints_are_kept_per_prefix_length(hip_engine):
    """
    An NPHD table of 256-bit rows answers 64-bit queries over a 64-bit prefix: their k-th distance (~10 bits) and a 256-bit query's
    (~95) must not share a hint -- alternating them keeps hitting, and every answer equals the oracle's.
    """
    from oracle import oracle_topk

    rng = np.random.default_rng(2468)
    n, k = 120_000, 10
    words = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    lens = np.full(n, 32, dtype=np.uint8)
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(1)
    t = hip_engine.open_table(1, 1, 32)
    try:
        t.add(keys, words, lens)

        def ask(nq, nbytes):
            q = rng.integers(0, 2**64, size=(nq, 4), dtype=np.uint64)
            q[:, (nbytes + 7) // 8 :] = 0
            ql = np.full(nq, nbytes, dtype=np.uint8)
            before = hip_engine.stats()
            got = t.search(q, ql, k)
            after = hip_engine.stats()
            exp = oracle_topk(1, keys, words, lens, q, ql, k)
            for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
                np.testing.assert_array_equal(g, e, err_msg=name)
            return after["spec_hits"] - before["spec_hits"], after["spec_misses"] - before["spec_misses"]

        for nq in (3, 200):
            assert ask(nq, 32) == (0, 0) and ask(nq, 8) == (0, 0)        # each length seeds its own hint
            for _ in range(3):
                assert ask(nq, 32) == (1, 0)
                assert ask(nq, 8) == (1, 0)
    finally:
        t.drop()


def test_tables_of_several_code_lengths_speculate_per_segment(hip_engine):
    """
    An ISCC-UNIT index holds 64 / 128 / 192 / 256-bit codes in ONE NPHD table.  Small batches over it are first tried with every
    segment listing its rows within (previous k-th NPHD) x compared bits + 2, merged and verified; near-duplicates, a query
    inside a cluster that overflows one segment's list, and queries of another length (their own hint) -- all against the oracle.
    """
    from oracle import oracle_topk

    rng = np.random.default_rng(8642)
    n, k = 160_000, 10
    lens = rng.choice([8, 16, 24, 32], size=n).astype(np.uint8)
    words = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    centre = rng.integers(0, 2**64, size=4, dtype=np.uint64)
    words[:30_000] = centre                                            # a cluster: 30 000 rows within a bit of `centre` ...
    words[:30_000, 0] ^= np.uint64(1) << rng.integers(0, 64, size=30_000).astype(np.uint64)
    lens[:30_000] = 32                                                 # ... all of them 256-bit codes
    for j in range(4):
        words[lens.astype(np.int64) <= 8 * j, j] = 0
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(1)
    t = hip_engine.open_table(1, 1, 32)
    try:
        t.add(keys, words, lens)

        def ask(q, nbytes):
            q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, 4).copy()
            q[:, (nbytes + 7) // 8 :] = 0
            ql = np.full(len(q), nbytes, dtype=np.uint8)
            before = hip_engine.stats()
            got = t.search(q, ql, k)
            after = hip_engine.stats()
            exp = oracle_topk(1, keys, words, lens, q, ql, k)
            for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
                np.testing.assert_array_equal(g, e, err_msg=name)
            return after["spec_hits"] - before["spec_hits"], after["spec_misses"] - before["spec_misses"]

        def rnd(nq):
            return rng.integers(0, 2**64, size=(nq, 4), dtype=np.uint64)

        for nq in (1, 5, 200):                                         # (200: above spec_max_queries -- every segment's single pass starts under the hint)
            assert ask(rnd(nq), 32) == (0, 0)                          # seeds the hint of this size and length
            assert ask(rnd(nq), 32) == (1, 0)
            assert ask(rnd(nq), 16) == (0, 0)                          # 128-bit queries: their own hint
            assert ask(rnd(nq), 16) == (1, 0)
            near = words[rng.integers(30_000, n, size=nq)].copy()
            near[:, 0] ^= np.uint64(5)
            assert ask(near, 32) == (1, 0)                             # near-duplicates of stored codes of every length
            assert ask(rnd(nq), 32) == (1, 0)
        q = rnd(5)
        q[2] = centre
        assert ask(q, 32) == (0, 1)                                    # the 256-bit segment's list overflows: ordinary path
        assert ask(rnd(5), 32) == (1, 0)
        hip_engine.set_option("speculate", 0)
        try:
            assert ask(rnd(5), 32) == (0, 0)
        finally:
            hip_engine.set_option("speculate", 1)
    finally:
        t.drop()
