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
