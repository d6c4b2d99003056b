"""
The CPU oracle against (a) every literal known-answer the reference's tests hold for this
boundary and (b) an independent numpy restatement.  Runs without a GPU.
"""

import json
import os

import numpy as np
import pytest

from oracle import oracle_topk, oracle_splitmix64_fill, pack_codes, ref_topk, load_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def _run_c(rows, query, count, metric, max_words, fixed_nbytes=0, key128=False):
    if rows:
        codes = [bytes(r[1]) for r in rows]
        words, nb = pack_codes(codes, max_words)
        if key128:
            keys = np.array([[r[0] >> 64, r[0] & (2**64 - 1)] for r in rows], dtype=np.uint64)
        else:
            keys = np.array([r[0] for r in rows], dtype=np.uint64)
    else:
        words = np.zeros((0, max_words), dtype=np.uint64)
        nb = np.zeros(0, dtype=np.uint8)
        keys = np.zeros((0, 2) if key128 else 0, dtype=np.uint64)
    q, qnb = pack_codes([bytes(query)], max_words)
    return oracle_topk(metric, keys, words, nb if metric else None, q, qnb if metric else None, count, fixed_nbytes=fixed_nbytes)


@pytest.mark.parametrize("case", _load("kat_hamming.json")["cases"], ids=lambda c: c["source"][:48])
def test_oracle_matches_reference_hamming_kats(case):
    nb = case["ndim"] // 8
    keys, ham, pbits, cnt = _run_c(case["rows"], case["query"], case["count"], 0, (nb + 7) // 8, fixed_nbytes=nb)
    c = int(cnt[0])
    assert c == len(case["expected_keys"])
    assert keys[0, :c].tolist() == case["expected_keys"]
    if case["expected_distances"] is not None:
        assert ham[0, :c].tolist() == case["expected_distances"]
    if case.get("strictly_increasing"):
        assert all(a < b for a, b in zip(ham[0, : c - 1], ham[0, 1:c]))
    assert all(int(p) == case["ndim"] for p in pbits[0, :c])


def test_oracle_multi_vector_semantics_is_min_per_key():
    """tests/test_usearch_multi.py:74-100: a key's distance is the minimum over its vectors."""
    case = _load("kat_hamming.json")["multi_cases"][0]
    # give every vector its own row key, then reduce per reference key
    rows = [[i, r[1]] for i, r in enumerate(case["rows"])]
    keys, ham, _, cnt = _run_c(rows, case["query"], len(rows), 0, 1, fixed_nbytes=4)
    best = {}
    for rk, h in zip(keys[0, : cnt[0]].tolist(), ham[0, : cnt[0]].tolist()):
        ref_key = case["rows"][rk][0]
        best[ref_key] = min(best.get(ref_key, 10**9), h)
    ranked = sorted(best.items(), key=lambda kv: (kv[1], kv[0]))[: case["count"]]
    assert [k for k, _ in ranked] == case["expected_keys"]
    assert [d for _, d in ranked] == case["expected_distances"]


def test_oracle_count_zero_rejected():
    with pytest.raises(ValueError):
        _run_c([[1, [1, 2, 3, 4]]], [1, 2, 3, 4], 0, 0, 1, fixed_nbytes=4)


@pytest.mark.parametrize("group", ["reference", "self"])
def test_oracle_matches_nphd_kats(group):
    for case in _load("kat_nphd.json")[group]:
        keys, ham, pbits, cnt = _run_c(case["rows"], case["query"], case["count"], 1, 4)
        got = [[int(k), int(h), int(p)] for k, h, p in zip(keys[0, : cnt[0]], ham[0, : cnt[0]], pbits[0, : cnt[0]])]
        assert got == case["expected"], case.get("source", case.get("note"))
        if "expected_scores" in case:
            # score = max(0, 1 - distance), iscc_search/indexes/usearch/index.py:2041-2043
            scores = [max(0.0, 1.0 - float(np.float32(h) / np.float32(p))) for _, h, p in got]
            assert scores == case["expected_scores"]


@pytest.mark.parametrize("seed", range(6))
def test_c_oracle_equals_numpy_restatement_nphd(seed):
    rng = np.random.default_rng(seed)
    n = 300
    lens = rng.choice([8, 16, 24, 32, 4, 1, 13], size=n)
    base = rng.integers(0, 256, size=32, dtype=np.uint8)
    codes = []
    for ln in lens:
        c = base[:ln].copy()
        flips = rng.integers(0, ln * 8, size=rng.integers(0, 6))
        for f in flips:
            c[f // 8] ^= 1 << (7 - f % 8)
        codes.append(c.tobytes())
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(1000)
    words, nb = pack_codes(codes, 4)
    for qlen in (8, 16, 32, 3, 21):
        query = base[:qlen].tobytes()
        q, qnb = pack_codes([query], 4)
        k = 25
        ck, ch, cp, cc = oracle_topk(1, keys, words, nb, q, qnb, k)
        exp = ref_topk(codes, [int(x) for x in keys], query, k, nphd=True)
        got = [(int(a), int(b), int(c)) for a, b, c in zip(ck[0, : cc[0]], ch[0, : cc[0]], cp[0, : cc[0]])]
        assert got == exp


@pytest.mark.parametrize("nbytes", [1, 4, 8, 16, 32])
def test_c_oracle_equals_numpy_restatement_hamming_128bit_keys(nbytes):
    rng = np.random.default_rng(nbytes)
    n = 200
    codes = [rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes() for _ in range(n)]
    keys_int = [int(rng.integers(0, 4)) << 64 | int(rng.integers(0, 2**63)) for _ in range(n)]
    keys = np.array([[k >> 64, k & (2**64 - 1)] for k in keys_int], dtype=np.uint64)
    mw = (nbytes + 7) // 8
    words, _ = pack_codes(codes, mw)
    query = codes[3]
    q, _ = pack_codes([query], mw)
    ck, ch, cp, cc = oracle_topk(0, keys, words, None, q, None, 30, fixed_nbytes=nbytes)
    exp = ref_topk(codes, keys_int, query, 30, nphd=False)
    got = [((int(a[0]) << 64) | int(a[1]), int(b), int(c)) for a, b, c in zip(ck[0, : cc[0]], ch[0, : cc[0]], cp[0, : cc[0]])]
    assert got == [(k, h, nbytes * 8) for k, h, _ in exp]


def test_oracle_thread_split_paths_agree():
    """Few queries over many rows splits rows across threads; many queries splits queries."""
    rng = np.random.default_rng(1)
    n = 1 << 17
    words = rng.integers(0, 2**64, size=(n, 1), dtype=np.uint64)
    keys = rng.permutation(n).astype(np.uint64)
    q = rng.integers(0, 2**64, size=(16, 1), dtype=np.uint64)
    many = oracle_topk(0, keys, words, None, q, None, 10)
    for i in range(0, 16, 5):
        one = oracle_topk(0, keys, words, None, q[i : i + 1], None, 10)
        for a, b in zip(many, one):
            np.testing.assert_array_equal(a[i], b[0])


def test_splitmix64_known_answers():
    """splitmix64 reference outputs for seed 0 / 1234567 (published test vectors of the generator)."""
    lib = load_oracle()
    # x -> splitmix64 step applied to state x (state is advanced by the golden gamma inside)
    assert lib.oracle_splitmix64(0) == 0xE220A8397B1DCDAF
    assert lib.oracle_splitmix64(0x9E3779B97F4A7C15) == 0x6E789E6AA1B965F4
    out = oracle_splitmix64_fill(4, seed=0x1511CC00, stride=4)
    assert out.dtype == np.uint64 and len(set(out.tolist())) == 4
    assert int(out[1]) == lib.oracle_splitmix64(0x1511CC00 + 4)
