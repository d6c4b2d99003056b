"""
``select_kernel`` with 1 024-thread blocks (option ``select_wide_from``: large LDS sort buffers, few queries) against the oracle and
against the 256-thread form: the same exact top-k under (distance, key), ties at the cut included, 64- and 128-bit keys, with and
without the tie class fitting the sort buffer.
"""

import numpy as np
import pytest

from oracle import oracle_topk

pytestmark = pytest.mark.gpu


def _table(engine, rng, n, key_words, nbytes, pool):
    """Codes drawn from a small pool with a few bits flipped: distances are coarse and tie classes at the cut are large."""
    words_per = (nbytes + 7) // 8
    base = rng.integers(0, 2**64, size=(pool, words_per), dtype=np.uint64)
    words = base[rng.integers(0, pool, size=n)].copy()
    flips = rng.integers(0, 3, size=n)
    for f in (1, 2):
        rows = np.nonzero(flips >= f)[0]
        words[rows, 0] ^= np.uint64(1) << rng.integers(0, 64, size=rows.shape[0]).astype(np.uint64)
    if nbytes % 8:
        words[:, -1] &= np.uint64((0xFFFFFFFFFFFFFFFF << (8 * (8 - nbytes % 8))) & 0xFFFFFFFFFFFFFFFF)
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(3)
    if key_words == 2:
        keys = np.stack([keys // np.uint64(7), keys], axis=1)
    t = engine.open_table(0, key_words, nbytes)
    t.add(keys, words)
    return t, keys, words, base


@pytest.mark.parametrize("key_words,nbytes", [(1, 8), (2, 16), (2, 32)])
def test_wide_blocks_select_what_the_oracle_selects(hip_engine, key_words, nbytes):
    rng = np.random.default_rng(100 * key_words + nbytes)
    t, keys, words, base = _table(hip_engine, rng, 120_000, key_words, nbytes, pool=40)
    try:
        for nq, k in ((5, 400), (130, 400), (3, 1500), (40, 4096), (9, 257)):
            q = base[rng.integers(0, len(base), size=nq)].copy()
            q[:, 0] ^= np.uint64(1) << rng.integers(0, 64, size=nq).astype(np.uint64)
            exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=nbytes)
            for wide_from in (1024, 2048, 1 << 30):
                hip_engine.set_option("select_wide_from", wide_from)
                got = t.search(q, None, k)
                for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
                    np.testing.assert_array_equal(g, e, err_msg=f"nq={nq} k={k} select_wide_from={wide_from}: {name}")
    finally:
        hip_engine.set_option("select_wide_from", 2048)
        t.drop()
