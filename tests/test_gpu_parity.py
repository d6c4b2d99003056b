"""
GPU parity: the HIP path (through the C-ABI) against the CPU oracle, bit-exact on keys, hamming
distances, prefix lengths and counts, on the same seeded inputs.
"""

import json
import os

import numpy as np
import pytest

from oracle import oracle_topk, pack_codes

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
METRIC_HAMMING, METRIC_NPHD = 0, 1


def _mask_to_len(words, nbytes):
    """Zero every bit beyond each row's byte length (big-endian packed words)."""
    n, mw = words.shape
    nb = np.broadcast_to(np.asarray(nbytes, dtype=np.int64), (n,))
    out = words.copy()
    for j in range(mw):
        valid = np.clip(nb - 8 * j, 0, 8)  # bytes of word j that belong to the code
        shift = ((8 * (8 - valid)) % 64).astype(np.uint64)
        mask = np.where(valid == 8, ~np.uint64(0), np.where(valid == 0, np.uint64(0), (~np.uint64(0)) << shift))
        out[:, j] &= mask
    return out


def _rand_words(rng, n, max_words, nbytes):
    """Random big-endian packed codes, zero beyond each row's byte length."""
    return _mask_to_len(rng.integers(0, 2**64, size=(n, max_words), dtype=np.uint64), nbytes)


def _check(table, keys, words, nbytes, q_words, q_nbytes, k, metric, fixed_nbytes=0):
    got = table.search(q_words, q_nbytes, k)
    exp = oracle_topk(metric, keys, words, nbytes, q_words, q_nbytes, k, fixed_nbytes=fixed_nbytes)
    np.testing.assert_array_equal(got[3], exp[3], err_msg="counts")
    for q in range(q_words.shape[0]):
        c = int(exp[3][q])
        np.testing.assert_array_equal(got[1][q, :c], exp[1][q, :c], err_msg=f"hamming q={q}")
        np.testing.assert_array_equal(got[2][q, :c], exp[2][q, :c], err_msg=f"prefix bits q={q}")
        np.testing.assert_array_equal(got[0][q, :c], exp[0][q, :c], err_msg=f"keys q={q}")


def test_golden_hamming_kats(hip_engine):
    """Every literal known-answer of the reference's usearch characterisation tests."""
    with open(os.path.join(GOLDEN, "kat_hamming.json")) as f:
        kat = json.load(f)
    for case in kat["cases"]:
        nb = case["ndim"] // 8
        t = hip_engine.open_table(METRIC_HAMMING, 1, nb)
        try:
            if case["rows"]:
                keys = np.array([r[0] for r in case["rows"]], dtype=np.uint64)
                words, _ = pack_codes([bytes(r[1]) for r in case["rows"]], t.max_words)
                t.add(keys, words)
            q, _ = pack_codes([bytes(case["query"])], t.max_words)
            keys_o, ham, pbits, cnt = t.search(q, None, case["count"])
            c = int(cnt[0])
            assert keys_o[0, :c].tolist() == case["expected_keys"], case["source"]
            if case["expected_distances"] is not None:
                assert ham[0, :c].tolist() == case["expected_distances"], case["source"]
            if case.get("strictly_increasing"):
                assert all(a < b for a, b in zip(ham[0, : c - 1], ham[0, 1:c]))
            assert all(p == case["ndim"] for p in pbits[0, :c])
        finally:
            t.drop()


def test_count_zero_is_value_error(hip_engine):
    t = hip_engine.open_table(METRIC_HAMMING, 1, 4)
    try:
        q, _ = pack_codes([bytes([1, 2, 3, 4])], 1)
        with pytest.raises(ValueError, match="`count` must be >= 1"):
            t.search(q, None, 0)
    finally:
        t.drop()


@pytest.mark.parametrize("n,k,nq", [(1, 10, 3), (100, 10, 5), (5000, 10, 33), (70000, 10, 40), (300000, 100, 17)])
def test_hamming64_random_vs_oracle(hip_engine, n, k, nq):
    rng = np.random.default_rng(n * 31 + k)
    t = hip_engine.open_table(METRIC_HAMMING, 1, 8)
    try:
        keys = rng.permutation(np.arange(1, n + 1, dtype=np.uint64) * np.uint64(2654435761))
        words = rng.integers(0, 2**64, size=(n, 1), dtype=np.uint64)
        t.add(keys, words)
        assert t.size == n
        q = rng.integers(0, 2**64, size=(nq, 1), dtype=np.uint64)
        q[0] = words[n // 2]           # exact hit
        q[1 % nq] = words[0] ^ np.uint64(0b1011)
        _check(t, keys, words, None, q, None, k, METRIC_HAMMING)
    finally:
        t.drop()


def test_hamming64_heavy_ties(hip_engine):
    """Few distinct codes: the k-th place always falls inside a large tie class, broken by key."""
    rng = np.random.default_rng(7)
    n, k = 50000, 25
    t = hip_engine.open_table(METRIC_HAMMING, 1, 8)
    try:
        base = rng.integers(0, 2**64, size=8, dtype=np.uint64)
        words = base[rng.integers(0, 8, size=n)].reshape(n, 1)
        keys = rng.permutation(np.arange(n, dtype=np.uint64) + np.uint64(10**12))
        t.add(keys, words)
        q = np.concatenate([base[:4], rng.integers(0, 2**64, size=4, dtype=np.uint64)]).reshape(-1, 1)
        _check(t, keys, words, None, q, None, k, METRIC_HAMMING)
    finally:
        t.drop()


@pytest.mark.parametrize("key_words", [1, 2])
def test_all_identical_codes_take_the_exact_fallback(hip_engine, key_words):
    """Every row equal: the tie class is the whole table, far beyond the candidate buffer AND beyond what
    the fallback collects in one go -> radix select on the key over the table itself."""
    n, k = 300000, 10
    rng = np.random.default_rng(3)
    t = hip_engine.open_table(METRIC_HAMMING, key_words, 8)
    try:
        words = np.full((n, 1), 0xDEADBEEFCAFEF00D, dtype=np.uint64)
        words[::1000] ^= np.uint64(1)                    # a few rows one bit away
        if key_words == 2:
            keys = np.stack([rng.integers(0, 5, size=n).astype(np.uint64), rng.permutation(n).astype(np.uint64) * np.uint64(2**40 + 7)], axis=1)
        else:
            keys = rng.permutation(np.arange(n, dtype=np.uint64) * np.uint64(2**33 + 5) + np.uint64(5))
        t.add(keys, words)
        before = hip_engine.stats()["fallback_queries"]
        q = np.array([[0xDEADBEEFCAFEF00D], [0xDEADBEEFCAFEF00C], [0x0123456789ABCDEF]], dtype=np.uint64)
        _check(t, keys, words, None, q, None, k, METRIC_HAMMING)
        _check(t, keys, words, None, q, None, 700, METRIC_HAMMING)
        assert hip_engine.stats()["fallback_queries"] > before
    finally:
        t.drop()


@pytest.mark.parametrize("nbytes", [1, 4, 8, 16, 24, 32, 13])
def test_hamming_fixed_lengths_128bit_keys(hip_engine, nbytes):
    """Simprint-style tables: fixed ndim, 128-bit composite keys."""
    rng = np.random.default_rng(nbytes)
    n, k, nq = 20000, 40, 9
    t = hip_engine.open_table(METRIC_HAMMING, 2, nbytes)
    try:
        mw = t.max_words
        words = _rand_words(rng, n, mw, nbytes)
        keys = rng.integers(0, 2**64, size=(n, 2), dtype=np.uint64)
        keys[:, 0] >>= np.uint64(50)  # many rows share an asset id: ties resolved on the low word
        keys = np.unique(keys, axis=0)
        n = keys.shape[0]
        words = words[:n]
        t.add(keys, words)
        q = _rand_words(rng, nq, mw, nbytes)
        q[0] = words[5]
        _check(t, keys, words, None, q, None, k, METRIC_HAMMING, fixed_nbytes=nbytes)
    finally:
        t.drop()


def test_nphd_mixed_lengths_vs_oracle(hip_engine):
    """NPHD over 64/128/192/256-bit (and odd) code lengths, queries of every length."""
    rng = np.random.default_rng(11)
    n, k = 60000, 20
    t = hip_engine.open_table(METRIC_NPHD, 1, 32)
    try:
        lens = rng.choice([8, 16, 24, 32, 4, 12, 20], size=n, p=[0.3, 0.2, 0.15, 0.2, 0.05, 0.05, 0.05]).astype(np.uint8)
        words = _rand_words(rng, n, 4, lens)
        keys = rng.permutation(np.arange(n, dtype=np.uint64) + np.uint64(1))
        t.add(keys, words, lens)
        qlens = np.array([8, 16, 24, 32, 4, 12, 8, 32, 20, 16], dtype=np.uint8)
        q = _rand_words(rng, len(qlens), 4, qlens)
        # two thirds of the queries are near-duplicates of stored rows, so short prefixes tie with long ones
        for i in range(len(qlens)):
            if i % 3 == 2:
                continue
            row = words[int(rng.integers(0, n))].copy()
            row[0] ^= np.uint64(1) << np.uint64(63 - i)
            q[i] = row
        q = _mask_to_len(q, qlens)
        _check(t, keys, words, lens, q, qlens, k, METRIC_NPHD)
    finally:
        t.drop()


def test_add_remove_get_contains(hip_engine):
    rng = np.random.default_rng(5)
    n = 3000
    t = hip_engine.open_table(METRIC_NPHD, 1, 32)
    try:
        lens = rng.choice([8, 16, 32], size=n).astype(np.uint8)
        words = _rand_words(rng, n, 4, lens)
        keys = np.arange(100, 100 + n, dtype=np.uint64)
        t.add(keys, words, lens)
        with pytest.raises(KeyError):
            t.add(keys[:1], words[:1], lens[:1])
        assert t.contains(np.array([100, 99, 100 + n - 1, 100 + n], dtype=np.uint64)).tolist() == [True, False, True, False]
        gw, gb = t.get(np.array([150, 7], dtype=np.uint64))
        assert gb.tolist() == [int(lens[50]), 0]
        np.testing.assert_array_equal(gw[0], words[50])
        # remove every third row, then parity again on what is left
        drop = keys[::3]
        assert t.remove(drop) == len(drop)
        assert t.remove(drop) == 0
        keep = np.ones(n, dtype=bool)
        keep[::3] = False
        assert t.size == int(keep.sum())
        q = _rand_words(rng, 6, 4, 32)
        qlens = np.full(6, 32, dtype=np.uint8)
        _check(t, keys[keep], words[keep], lens[keep], q, qlens, 15, METRIC_NPHD)
        # removed keys can come back with new codes
        t.add(drop[:10], words[:10], lens[:10])
        assert t.size == int(keep.sum()) + 10
    finally:
        t.drop()


def test_synthetic_generator_matches_host_formula(hip_engine):
    from oracle import oracle_splitmix64_fill

    n, seed = 100000, 0x1511CC00
    t = hip_engine.open_table(METRIC_HAMMING, 1, 8)
    try:
        t.add_synthetic(8, n, seed)
        words = oracle_splitmix64_fill(n, seed, stride=4).reshape(n, 1)
        keys = np.arange(n, dtype=np.uint64)
        q = words[[17, 4242, 99999]] ^ np.array([[0], [0b111], [1 << 40]], dtype=np.uint64)
        _check(t, keys, words, None, q, None, 10, METRIC_HAMMING)
    finally:
        t.drop()
