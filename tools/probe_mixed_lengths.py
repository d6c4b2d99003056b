#!/usr/bin/env python3
"""
An ISCC-UNIT index as the reference fills it: ONE NPHD table holding codes of 64 / 128 / 192 / 256 bits (25 M rows each here).
A query is compared with every stored length over the common prefix; the per-length lists are merged by NPHD rank.
Time per search for batches of 1 / 32 / 1 024 queries of 256-bit and of 64-bit codes.

usage (GPU box): python tools/probe_mixed_lengths.py [rows per length, default 25000000]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
eng = HipEngine(0)
t = eng.open_table(_lib.METRIC_NPHD, 1, 32)
for i, nb in enumerate((8, 16, 24, 32)):
    t.add_synthetic(nb, rows, seed=7 + i, first_row=0, key_base=i * rows)
rng = np.random.default_rng(1)
for nb in (32, 8):
    for nq in (1, 32, 1024):
        q = rng.integers(0, 2**64, size=(nq, 4), dtype=np.uint64)
        q[:, (nb + 7) // 8:] = 0
        ql = np.full(nq, nb, dtype=np.uint8)
        for _ in range(10):
            t.search(q, ql, 10)
        s0 = eng.stats()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            out = t.search(q, ql, 10)
        dt = (time.perf_counter() - t0) / reps
        s1 = eng.stats()
        print("%4d queries of %3d bits over 4 x %d rows: %.3f ms per search (%d scan launches, %d level launches per search; fallbacks %d)" % (
            nq, nb * 8, rows, dt * 1e3, (s1["scan_launches"] - s0["scan_launches"]) // reps, (s1["level_launches"] - s0["level_launches"]) // reps,
            s1["fallback_queries"] - s0["fallback_queries"]))
