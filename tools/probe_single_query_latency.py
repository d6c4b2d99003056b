#!/usr/bin/env python3
"""One query per call, a different query every call (the reference's per-unit call shape), over tables of the deployment guide's sizes:
time per call and how the speculative pass fared.  usage (GPU box): python tools/probe_single_query_latency.py [rows ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(0)
for rows in [int(a) for a in sys.argv[1:]] or [2_500, 16_384, 100_000, 1_000_000, 10_000_000]:
    t = eng.open_table(_lib.METRIC_HAMMING, 1, 8)
    t.add_synthetic(8, rows, 1)
    rng = np.random.default_rng(rows)
    qs = rng.integers(0, 2**64, size=(3000, 1, 1), dtype=np.uint64)
    for q in qs[:500]:
        t.search(q, None, 10)
    before = eng.stats()
    t0 = time.perf_counter()
    for q in qs[500:]:
        t.search(q, None, 10)
    dt = (time.perf_counter() - t0) / 2500
    after = eng.stats()
    print(f"{rows:>9} rows: {dt * 1e6:6.1f} us per call; speculative passes {after['spec_hits'] - before['spec_hits']} held / {after['spec_misses'] - before['spec_misses']} did not; "
          f"scan launches per call {(after['scan_launches'] - before['scan_launches']) / 2500:.2f}")
    t.drop()
eng.close()
