"""
``isccsearch_search_many``: several searches, one synchronisation.  Every request must return exactly what the
single-request entry points return (they are checked against the oracle elsewhere), whichever internal path
it takes: deferred (single-segment table), ordinary (multi-segment / mixed query lengths / empty table), or the
exact fallback after a candidate-list overflow.
"""

import numpy as np
import pytest

from test_gpu_parity import METRIC_HAMMING, METRIC_NPHD, _mask_to_len, _rand_words

pytestmark = pytest.mark.gpu


def _same(a, b, tag):
    for x, y, name in zip(a, b, ("keys", "hamming", "prefix_bits", "count")):
        np.testing.assert_array_equal(x, y, err_msg=f"{tag}: {name}")


def test_search_many_equals_single_calls(hip_engine):
    rng = np.random.default_rng(77)
    t1 = hip_engine.open_table(METRIC_HAMMING, 1, 8)
    t2 = hip_engine.open_table(METRIC_NPHD, 1, 32)
    t3 = hip_engine.open_table(METRIC_HAMMING, 2, 16)
    t4 = hip_engine.open_table(METRIC_NPHD, 1, 32)      # one segment only: deferred even though it is an NPHD table
    empty = hip_engine.open_table(METRIC_HAMMING, 1, 8)
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
        got = hip_engine.search_many(requests)
        for i, ((table, q, nb, k, r), g) in enumerate(zip(requests, got)):
            want = table.search(q, nb, k) if r is None else table.search_within(q, nb, k, r)
            _same(g, want, f"request {i}")
        assert hip_engine.search_many([]) == []
        with pytest.raises(ValueError):
            hip_engine.search_many([(t1, q1, None, 0, None)])
    finally:
        for t in (t1, t2, t3, t4, empty):
            t.drop()


def test_search_many_overflow_takes_the_exact_fallback(hip_engine):
    n = 120000
    t = hip_engine.open_table(METRIC_HAMMING, 1, 8)
    try:
        words = np.full((n, 1), 0x1111111111111111, dtype=np.uint64)
        words[::3] ^= np.uint64(1)
        t.add(np.arange(n, dtype=np.uint64)[::-1].copy() + np.uint64(5), words)
        q = np.array([[0x1111111111111111], [0x1111111111111110], [0x7777777777777777]], dtype=np.uint64)
        before = hip_engine.stats()["fallback_queries"]
        got = hip_engine.search_many([(t, q, None, 10, None), (t, q, None, 50, 0), (t, q[:1], None, 5, None)])
        assert hip_engine.stats()["fallback_queries"] > before
        _same(got[0], t.search(q, None, 10), "plain")
        _same(got[1], t.search_within(q, None, 50, 0), "within")
        _same(got[2], t.search(q[:1], None, 5), "third")
    finally:
        t.drop()
