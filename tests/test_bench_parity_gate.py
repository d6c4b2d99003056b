"""
``bench.py``'s ``cpu_baseline`` leg is also its parity gate: the oracle's answers for the timed workload are compared
bit for bit with what the GPU returned, and a difference ends the run non-zero.  (CPU tier: the "GPU" answer is handed in.)
"""

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_fails_loudly_when_results_differ_from_the_oracle():
    """`cpu_baseline` is also the parity gate of the bench: a wrong GPU answer must end the run non-zero."""
    import numpy as np

    sys.path.insert(0, ROOT)
    import bench

    class Args:
        rows, nbytes, metric, k, cpu_queries, queries = 200_000, 8, "hamming", 10, 16, 16

    q, _ = bench.make_queries(16, Args.rows, 1)
    from oracle import oracle_splitmix64_fill, oracle_topk

    words = oracle_splitmix64_fill(Args.rows, bench.SEED_CODES, stride=4).reshape(-1, 1)
    good = oracle_topk(0, np.arange(Args.rows, dtype=np.uint64), words, None, q, None, 10, fixed_nbytes=8)
    assert bench.cpu_baseline(Args, q, 1, good)["parity_checked_queries"] == 16
    bad = tuple(a.copy() for a in good)
    bad[0][5, 3] ^= np.uint64(1)
    with pytest.raises(SystemExit, match="PARITY FAILURE"):
        bench.cpu_baseline(Args, q, 1, bad)
