"""
The C-ABI from plain C: ``tests/abi_client.c`` is compiled with gcc against ``include/isccsearch.h`` and
``libisccsearch_hip.so`` -- no ctypes, numpy or torch in that process -- run on the GPU, and its printed answers are compared
with the oracle's for the same rows and queries.  This is what a cgo / JNI / N-API binding of the reference's host language
would exercise.
"""

import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import np_within, oracle_topk

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "iscc_search_amd", "csrc")
MASK = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & MASK
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & MASK
    return x ^ (x >> 31)


def _parse(lines, tag):
    out = {}
    for line in lines:
        if not line.startswith(tag + " "):
            continue
        head, _, tail = line.partition(":")
        _, j, count = head.split()
        pairs = [tuple(int(v) for v in item.split(":")) for item in tail.split()]
        assert len(pairs) == int(count)
        out[int(j)] = pairs
    return out


@pytest.mark.parametrize("n,nq,k", [(70_000, 9, 5), (3_000, 40, 12)])
def test_plain_c_client_gets_the_oracles_answers(tmp_path, n, nq, k):
    exe = tmp_path / "abi_client"
    subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi_client.c"), "-o", str(exe), "-L", CSRC, "-lisccsearch_hip",
                    "-Wl,-rpath," + CSRC], check=True)
    p = subprocess.run([str(exe), str(n), str(nq), str(k)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()

    seed_a, seed_b = 0x1511CC00, 0x0BADC0DE
    words = np.array([_splitmix64(seed_a + 4 * i) for i in range(n)] + [_splitmix64(seed_b + 4 * i) for i in range(n)], dtype=np.uint64).reshape(-1, 1)
    keys = np.arange(1000, 1000 + 2 * n, dtype=np.uint64)
    q = np.array([int(words[(7919 * j) % n, 0]) ^ ((1 << (j % 5)) - 1) for j in range(nq)], dtype=np.uint64).reshape(-1, 1)

    exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=8)
    got = _parse(lines, "q")
    for j in range(nq):
        c = int(exp[3][j])
        assert got[j] == [(int(exp[0][j, i]), int(exp[1][j, i])) for i in range(c)], f"query {j}"

    within = _parse(lines, "w")
    for j in range(nq):
        ek, eh, _ = np_within(words, 8, keys, q[j], 8, k, 2)
        assert within[j] == [(int(a), int(b)) for a, b in zip(ek, eh)], f"range-limited query {j}"

    assert f"removed 1 found 0 size {2 * n - 1}" in lines
    stats = [line.split() for line in lines if line.startswith("searches ")][0]
    assert int(stats[1]) == 2 and int(stats[3]) == 2 * nq, stats


def test_plain_c_client_scores_simprints_like_the_reference_loop(tmp_path):
    """``isccsearch_simprint_score`` called from C: scores printed with 17 digits must be the checker's float64s."""
    import struct

    from iscc_search_amd.simprint import DOC_FREQ_DUP_LIMIT, HipSimprintIndex, pack_chunk_pointer
    from oracle_engine import OracleEngine
    from simprint_checker import score_lists

    assets, chunks, nq, limit = 300, 6, 23, 12
    exe = tmp_path / "abi_client"
    subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi_client.c"), "-o", str(exe), "-L", CSRC, "-lisccsearch_hip",
                    "-Wl,-rpath," + CSRC], check=True)
    p = subprocess.run([str(exe), "simprint", str(assets), str(chunks), str(nq), str(limit)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()

    as_bytes = lambda word: struct.pack(">Q", word)
    pool = [_splitmix64(0x51 + i) for i in range(50)]
    keys, vecs = [], []
    for a in range(1, assets + 1):
        for c in range(chunks):
            hh = _splitmix64(77 + 131 * a + c)
            word = pool[hh % 50] ^ ((((1 << ((hh >> 32) % 4)) - 1) << ((hh >> 40) % 60)) & MASK)
            keys.append(pack_chunk_pointer(a.to_bytes(8, "big"), 10 * c, 10 + c))
            vecs.append(np.frombuffer(as_bytes(word), dtype=np.uint8))
    oracle = HipSimprintIndex(OracleEngine(), ndim=64)
    oracle.add_raw(keys, vecs)
    simprints = [as_bytes(pool[j % 50] ^ ((1 << (j % 3)) - 1)) for j in range(nq)]
    batch = oracle._index.search(np.stack([np.frombuffer(s, dtype=np.uint8) for s in simprints]), count=4 * limit)
    lists = [[(bytes(k), int(h)) for k, h in zip(batch[q].keys, batch[q].hamming)] for q in range(nq)]
    freq = lambda s: int(oracle._index.doc_freq(np.frombuffer(s, dtype=np.uint8).reshape(1, -1), DOC_FREQ_DUP_LIMIT)[0])
    stored = lambda key: (lambda v: None if v is None else v.tobytes())(oracle._index.get(key))
    want = score_lists(simprints, lists, 64, limit, 0.8, stored, freq, assets)
    assert len(want) == limit

    got = [line for line in lines if line.startswith("r ")]
    assert len(got) == len(want) and lines[0].split()[1] == str(len(want))
    for line, (asset, score, matches, chunks_) in zip(got, want):
        head, _, tail = line.partition(":")
        _, a, s, m = head.split()
        assert int(a) == int.from_bytes(asset, "big") and float(s) == score and int(m) == matches, (line, score)
        seen = [tuple(int(v) for v in item.split(":")) for item in tail.split()]
        exp = sorted((qi, (offset << 32) | size, round((1.0 - sim) * 64), f, int.from_bytes(match, "big")) for qi, match, sim, offset, size, f in chunks_)
        assert seen == exp, line
