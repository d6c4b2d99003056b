"""
GPU parity of the range-limited search (``isccsearch_search_within``) and the document frequency
(``isccsearch_doc_freq``) against the numpy restatement ``oracle.np_within`` -- bit-exact keys, Hamming
distances, prefix lengths and counts, through the C-ABI.
"""

import numpy as np
import pytest

from oracle import np_within
from test_gpu_parity import METRIC_HAMMING, METRIC_NPHD, _mask_to_len, _rand_words

pytestmark = pytest.mark.gpu


def _check_within(table, keys, words, nbytes, q_words, q_nbytes, k, r):
    got = table.search_within(q_words, q_nbytes, k, r)
    for q in range(q_words.shape[0]):
        qb = table.max_bytes if q_nbytes is None else int(q_nbytes[q])
        ek, eh, ep = np_within(words, table.max_bytes if nbytes is None else nbytes, keys, q_words[q], qb, k, r)
        c = len(eh)
        assert int(got[3][q]) == c, f"count q={q} r={r}: {int(got[3][q])} != {c}"
        np.testing.assert_array_equal(got[1][q, :c], eh, err_msg=f"hamming q={q} r={r}")
        np.testing.assert_array_equal(got[2][q, :c], ep, err_msg=f"prefix bits q={q} r={r}")
        np.testing.assert_array_equal(got[0][q, :c], ek, err_msg=f"keys q={q} r={r}")
        assert not got[0][q, c:].any() and not got[1][q, c:].any()


@pytest.mark.parametrize("nbytes,key_words", [(8, 1), (8, 2), (16, 2), (32, 2), (13, 1)])
def test_within_fixed_length_vs_numpy(hip_engine, nbytes, key_words):
    """Planted collisions and near-collisions among random rows; radii from 0 up to 'most of the table'."""
    rng = np.random.default_rng(100 + nbytes + key_words)
    n, nq = 40000, 11
    t = hip_engine.open_table(METRIC_HAMMING, key_words, nbytes)
    try:
        mw = t.max_words
        words = _rand_words(rng, n, mw, nbytes)
        pool = _rand_words(rng, 6, mw, nbytes)
        for i in range(0, n, 7):                       # every 7th row is one of six codes ...
            words[i] = pool[int(rng.integers(0, 6))]
            if i % 3 == 0:                             # ... a third of them one or two bits away
                words[i, 0] ^= np.uint64(1) << np.uint64(int(rng.integers(56, 64)))
        if key_words == 2:
            keys = np.stack([rng.integers(1, 300, size=n).astype(np.uint64), rng.permutation(n).astype(np.uint64)], axis=1)
        else:
            keys = rng.permutation(np.arange(n, dtype=np.uint64) * np.uint64(977) + np.uint64(3))
        t.add(keys, words)
        q = np.concatenate([pool, _rand_words(rng, nq - 6, mw, nbytes)])
        for r, k in [(0, 1000), (0, 50), (1, 4096), (2, 300), (8 * nbytes // 4, 64)]:
            _check_within(t, keys, words, None, q, None, k, r)
        # document frequency = distinct first key words among the first dup_limit collisions
        for dup_limit in (1000, 37):
            freq = t.doc_freq(q, None, dup_limit)
            for qi in range(q.shape[0]):
                ek, _, _ = np_within(words, nbytes, keys, q[qi], nbytes, dup_limit, 0)
                want = len(np.unique(ek[:, 0])) if key_words == 2 else len(ek)
                assert int(freq[qi]) == want, (qi, dup_limit)
    finally:
        t.drop()


def test_within_radius_beyond_the_candidate_buffer_takes_the_exact_fallback(hip_engine):
    """A radius that admits most of the table overflows the candidate lists: the fallback must still
    return the k nearest rows WITHIN the radius -- and nothing when the radius admits nothing."""
    rng = np.random.default_rng(9)
    n = 200000
    t = hip_engine.open_table(METRIC_HAMMING, 1, 8)
    try:
        words = rng.integers(0, 2**64, size=(n, 1), dtype=np.uint64)
        words[: n // 2] = np.uint64(0x0F0F0F0F0F0F0F0F)            # 100 000 equal rows
        keys = rng.permutation(np.arange(n, dtype=np.uint64) + np.uint64(1))
        t.add(keys, words)
        q = np.array([[0x0F0F0F0F0F0F0F0F], [0x0F0F0F0F0F0F0F0E], [0xF0F0F0F0F0F0F0F0], [0x123456789ABCDEF0]], dtype=np.uint64)
        before = hip_engine.stats()["fallback_queries"]
        for r, k in [(0, 1000), (1, 10), (40, 100), (64, 4096)]:
            _check_within(t, keys, words, None, q, None, k, r)
        assert hip_engine.stats()["fallback_queries"] > before
    finally:
        t.drop()


def test_within_nphd_mixed_lengths_is_the_prefix_match(hip_engine):
    """NPHD table: radius 0 = bidirectional prefix equality (INSTANCE matching, usearch/index.py:1957-2022)."""
    rng = np.random.default_rng(21)
    n = 30000
    t = hip_engine.open_table(METRIC_NPHD, 1, 32)
    try:
        lens = rng.choice([8, 16, 32, 12], size=n, p=[0.4, 0.3, 0.2, 0.1]).astype(np.uint8)
        words = _rand_words(rng, n, 4, lens)
        base = _rand_words(rng, 3, 4, 32)
        for i in range(0, n, 11):                       # families of codes sharing a prefix at every length
            words[i] = _mask_to_len(base[i % 3: i % 3 + 1], lens[i: i + 1])[0]
            if i % 5 == 0:
                words[i, 0] ^= np.uint64(1)             # differs in bit 63 of word 0 (inside every prefix)
        keys = rng.permutation(np.arange(n, dtype=np.uint64) + np.uint64(1))
        t.add(keys, words, lens)
        qlens = np.array([32, 16, 8, 12, 32, 8], dtype=np.uint8)
        q = _mask_to_len(np.concatenate([base, base[:1], _rand_words(rng, 2, 4, 32)]), qlens)
        for r, k in [(0, 4096), (1, 4096), (0, 5), (3, 100)]:
            _check_within(t, keys, words, lens, q, qlens, k, r)
    finally:
        t.drop()


def test_within_edges(hip_engine):
    t = hip_engine.open_table(METRIC_HAMMING, 2, 8)
    try:
        q = np.array([[5]], dtype=np.uint64)
        keys, ham, pbits, cnt = t.search_within(q, None, 10, 0)          # empty table
        assert cnt.tolist() == [0] and t.doc_freq(q).tolist() == [0]
        t.add(np.array([[1, 1], [1, 2], [2, 1]], dtype=np.uint64), np.array([[5], [5], [7]], dtype=np.uint64))
        keys, ham, pbits, cnt = t.search_within(q, None, 10, 0)
        assert cnt.tolist() == [2] and keys[0, :2].tolist() == [[1, 1], [1, 2]] and t.doc_freq(q).tolist() == [1]
        keys, ham, pbits, cnt = t.search_within(q, None, 10, 1)          # 5 ^ 7 = 2: one bit away
        assert cnt.tolist() == [3] and ham[0, :3].tolist() == [0, 0, 1]
        with pytest.raises(ValueError):
            t.search_within(q, None, 0, 0)
        with pytest.raises(ValueError):
            t.search_within(q, None, 10, 257)
    finally:
        t.drop()


def _freq_model(keys, words, dup_limit):
    """Per row: distinct assets among the first dup_limit rows (ascending key) holding the same code."""
    groups = {}
    for i in range(len(keys)):
        groups.setdefault(words[i].tobytes(), []).append(i)
    out = np.zeros(len(keys), dtype=np.uint32)
    for rows in groups.values():
        if keys.ndim == 2:
            rows_sorted = sorted(rows, key=lambda r: (int(keys[r, 0]), int(keys[r, 1])))[:dup_limit]
            f = len({int(keys[r, 0]) for r in rows_sorted})
        else:
            f = min(len(rows), dup_limit)
        out[rows] = f
    return out


@pytest.mark.parametrize("nbytes,key_words", [(8, 2), (16, 2), (32, 2), (13, 2), (8, 1)])
def test_frequency_column_vs_model(hip_engine, nbytes, key_words):
    """isccsearch_get_freq (sort-based column) == the dict model == isccsearch_doc_freq (collision scan)."""
    rng = np.random.default_rng(500 + nbytes + key_words)
    n = 30000
    t = hip_engine.open_table(METRIC_HAMMING, key_words, nbytes)
    try:
        mw = t.max_words
        pool = _rand_words(rng, 400, mw, nbytes)
        pool[1] = pool[0]
        pool[1, mw - 1] ^= np.uint64(1) << np.uint64(63)     # differs from pool[0] in the LAST word only
        pick = np.minimum(rng.geometric(0.02, size=n) - 1, 399)   # skewed: a few codes are very common
        words = pool[pick]
        words[n // 2:] = _rand_words(rng, n - n // 2, mw, nbytes)  # and half the rows are unique
        if key_words == 2:
            keys = np.stack([rng.integers(1, 60, size=n).astype(np.uint64), rng.permutation(n).astype(np.uint64)], axis=1)
        else:
            keys = rng.permutation(np.arange(n, dtype=np.uint64) * np.uint64(31) + np.uint64(9))
        t.add(keys, words)
        builds = hip_engine.stats()["freq_builds"]
        for dup_limit in (1000, 7):
            want = _freq_model(keys, words, dup_limit)
            got = t.get_freq(keys, dup_limit)
            np.testing.assert_array_equal(got, want)
            sample = rng.integers(0, n, size=40)
            np.testing.assert_array_equal(t.doc_freq(words[sample], None, dup_limit), want[sample])
            # the counted form: the same frequencies and, beside them, how many colliding rows each was taken over
            freq, coll = t.doc_freq_counted(words[sample], None, dup_limit)
            np.testing.assert_array_equal(freq, want[sample])
            np.testing.assert_array_equal(coll, t.search_within(words[sample], None, dup_limit, 0)[3])
        assert hip_engine.stats()["freq_builds"] == builds + 2          # one build per dup_limit, reused by lookups
        t.get_freq(keys[:10], 7)
        assert hip_engine.stats()["freq_builds"] == builds + 2
        # absent keys -> 0; rows changing -> the column is rebuilt
        absent = keys[:3].copy()
        if key_words == 2:
            absent[:, 1] += np.uint64(10**15)
        else:
            absent += np.uint64(10**15)
        assert t.get_freq(absent).tolist() == [0, 0, 0]
        t.remove(keys[:5000])
        keys2, words2 = keys[5000:], words[5000:]
        np.testing.assert_array_equal(t.get_freq(keys2, 1000), _freq_model(keys2, words2, 1000))
        extra_k = keys[:100].copy()
        t.add(extra_k, words2[:100])
        keys3, words3 = np.concatenate([keys2, extra_k]), np.concatenate([words2, words2[:100]])
        np.testing.assert_array_equal(t.get_freq(keys3, 1000), _freq_model(keys3, words3, 1000))
    finally:
        t.drop()


def test_frequency_column_is_for_fixed_length_tables(hip_engine):
    t = hip_engine.open_table(METRIC_NPHD, 1, 32)
    try:
        t.add(np.array([1], dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64), np.array([8], dtype=np.uint8))
        with pytest.raises(ValueError):
            t.get_freq(np.array([1], dtype=np.uint64))
    finally:
        t.drop()


def test_two_shards_within_merged_on_device_equal_the_unsharded_answer(hip_engine):
    """isccsearch_search_within_device per shard + merge_kernel == isccsearch_search_within on the whole table;
    the product ShardedTable (one rank, no process group) goes through the same device path."""
    import torch

    from iscc_search_amd.sharded import HipShardOps, ShardedTable, block_bytes

    rng = np.random.default_rng(31)
    n, nq = 50000, 9
    words = rng.integers(0, 5, size=(n, 2), dtype=np.uint64) * np.uint64(0x1111111111111111)
    keys = np.stack([rng.integers(1, 40, size=n).astype(np.uint64), rng.permutation(n).astype(np.uint64)], axis=1)
    q = np.concatenate([words[:6], rng.integers(0, 2**64, size=(3, 2), dtype=np.uint64)])
    whole = hip_engine.open_table(METRIC_HAMMING, 2, 16)
    shards = [hip_engine.open_table(METRIC_HAMMING, 2, 16) for _ in range(2)]
    try:
        whole.add(keys, words)
        shards[0].add(keys[: n // 3], words[: n // 3])
        shards[1].add(keys[n // 3:], words[n // 3:])
        for r, k in ((0, 1000), (0, 11), (4, 300), (128, 50)):
            exp = whole.search_within(q, None, k, r)
            rec_bytes, blk = block_bytes(nq, k)
            gathered = torch.empty(2 * blk, dtype=torch.uint8, device="cuda:0")
            for i, t in enumerate(shards):
                base = gathered.data_ptr() + i * blk
                t.search_device(q, None, k, base, base + rec_bytes, max_hamming=r)
            torch.cuda.synchronize()
            merged = hip_engine.merge_device(2, nq, k, 2, gathered.data_ptr(), gathered.data_ptr() + rec_bytes, blk, blk)
            for a, b in zip(exp, merged):
                np.testing.assert_array_equal(a, b)
            one = ShardedTable(HipShardOps(whole, "cuda:0")).search_within(q, None, k, r)
            for a, b in zip(exp, one):
                np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(ShardedTable(HipShardOps(whole, "cuda:0")).doc_freq(q, None, 1000), whole.doc_freq(q, None, 1000))
    finally:
        for t in shards + [whole]:
            t.drop()


def test_out_of_memory_is_reported_and_the_engine_stays_usable(hip_engine):
    """A reservation beyond the 288 GB of HBM fails with -ENOMEM -> MemoryError; tables and searches keep working."""
    t = hip_engine.open_table(METRIC_HAMMING, 1, 8)
    try:
        t.add(np.array([1, 2, 3], dtype=np.uint64), np.array([[5], [6], [7]], dtype=np.uint64))
        with pytest.raises(MemoryError):
            t.reserve(8, 60_000_000_000)          # 60 G rows x (8 B code + 8 B key) = 960 GB
        assert t.size == 3
        keys, ham, _, cnt = t.search(np.array([[5]], dtype=np.uint64), None, 2)
        assert cnt.tolist() == [2] and keys[0].tolist() == [1, 3] and ham[0].tolist() == [0, 1]
        t.add(np.array([4], dtype=np.uint64), np.array([[5]], dtype=np.uint64))
        assert t.search_within(np.array([[5]], dtype=np.uint64), None, 10, 0)[3].tolist() == [2]
    finally:
        t.drop()
