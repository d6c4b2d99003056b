#!/usr/bin/env python3
"""search_exact alone (10 M x 128-bit chunks, 512 stored simprints as queries) for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 tools/probe_exact.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_simprint import build  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(0)
idx, first, _ = build(eng, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, np.random.default_rng(0))
q = [bytes(r) for r in first[:512]]
for _ in range(3):
    idx.search_exact(q, limit=20, threshold=0.0, detailed=True)
t0 = time.perf_counter()
for _ in range(10):
    res = idx.search_exact(q, limit=20, threshold=0.0, detailed=True)
print(f"search_exact nq=512: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms, {len(res)} assets")
idx.close()
eng.close()
