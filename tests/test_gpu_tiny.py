"""
Small segments answered by ONE launch (``tiny_search_kernel``: option ``tiny_rows``, default 16 384 rows): exact top-k and
range-limited searches against the oracle and against the ordinary path (``tiny_rows = 0``) on the same tables -- Hamming and NPHD
with mixed lengths and odd byte counts, 64- and 128-bit keys, k beyond the row count, ties at the cut, one query and hundreds.
"""

import numpy as np
import pytest

from oracle import np_within, oracle_topk
from test_gpu_parity import METRIC_HAMMING, METRIC_NPHD, _mask_to_len, _rand_words

pytestmark = pytest.mark.gpu


def _searches(engine, t, keys, words, lens, q, qlens, metric, fixed):
    for k in (1, 10, 400, min(len(keys) + 5, 4096)):
        exp = oracle_topk(metric, keys, words, lens, q, qlens, k, fixed_nbytes=fixed)
        for tiny_rows in (16384, 0):
            engine.set_option("tiny_rows", tiny_rows)
            got = t.search(q, qlens, k)
            np.testing.assert_array_equal(got[3], exp[3], err_msg=f"k={k} tiny_rows={tiny_rows}: counts")
            for i in range(q.shape[0]):
                c = int(exp[3][i])
                for g, e, name in zip(got[:3], exp[:3], ("keys", "hamming", "prefix_bits")):
                    np.testing.assert_array_equal(g[i, :c], e[i, :c], err_msg=f"k={k} tiny_rows={tiny_rows} query {i}: {name}")


@pytest.mark.parametrize("n", [1, 7, 300, 5000, 16384])
@pytest.mark.parametrize("key_words", [1, 2])
def test_hamming_tables_of_a_few_thousand_rows(hip_engine, n, key_words):
    rng = np.random.default_rng(1000 * key_words + n)
    for nbytes in (8, 13, 32):
        t = hip_engine.open_table(METRIC_HAMMING, key_words, nbytes)
        try:
            base = _rand_words(rng, 12, (nbytes + 7) // 8, nbytes)
            words = base[rng.integers(0, len(base), size=n)].copy()
            words[:, 0] ^= (rng.integers(0, 4, size=n).astype(np.uint64)) << np.uint64(60)      # coarse distances: ties at every cut
            keys = rng.permutation(n).astype(np.uint64) + np.uint64(9)
            if key_words == 2:
                keys = np.stack([keys % np.uint64(5), keys], axis=1)
            t.add(keys, words)
            for nq in (1, 5, 130):
                q = base[rng.integers(0, len(base), size=nq)].copy()
                q[:, 0] ^= np.uint64(1) << rng.integers(56, 64, size=nq).astype(np.uint64)
                _searches(hip_engine, t, keys, words, None, q, None, METRIC_HAMMING, nbytes)
            # range-limited, the radius of collision lookups and a wider one
            q1 = words[int(rng.integers(0, n))].copy()
            for radius in (0, 3):
                ek, eh, _ = np_within(words, nbytes, keys, q1, nbytes, 50, radius)
                for tiny_rows in (16384, 0):
                    hip_engine.set_option("tiny_rows", tiny_rows)
                    gk, gh, _, gc = t.search_within(q1.reshape(1, -1), None, 50, radius)
                    assert int(gc[0]) == len(eh), (radius, tiny_rows)
                    np.testing.assert_array_equal(gh[0, : len(eh)], eh)
                    np.testing.assert_array_equal(gk[0, : len(eh)], ek)
        finally:
            hip_engine.set_option("tiny_rows", 16384)
            t.drop()


def test_nphd_segments_of_mixed_lengths(hip_engine):
    """Every length is its own segment: some tiny, one beyond ``tiny_rows`` -- one launch each, merged as ever."""
    rng = np.random.default_rng(77)
    n = 30000
    t = hip_engine.open_table(METRIC_NPHD, 1, 32)
    try:
        lens = rng.choice([8, 16, 24, 32, 4, 12], size=n, p=[0.7, 0.1, 0.05, 0.1, 0.03, 0.02]).astype(np.uint8)
        base = rng.integers(0, 2**64, size=(20, 4), dtype=np.uint64)
        words = base[rng.integers(0, len(base), size=n)].copy()
        words[:, 0] ^= rng.integers(0, 8, size=n).astype(np.uint64) << np.uint64(59)
        words = _mask_to_len(words, lens)
        keys = rng.permutation(np.arange(n, dtype=np.uint64) + np.uint64(1))
        t.add(keys, words, lens)
        qlens = np.array([8, 16, 24, 32, 4, 12, 8, 32], dtype=np.uint8)
        q = _mask_to_len(base[rng.integers(0, len(base), size=len(qlens))].copy(), qlens)
        _searches(hip_engine, t, keys, words, lens, q, qlens, METRIC_NPHD, 0)
    finally:
        hip_engine.set_option("tiny_rows", 16384)
        t.drop()
