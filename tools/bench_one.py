#!/usr/bin/env python3
"""One configurable search loop for profiling: python tools/bench_one.py ROWS NBYTES NQ K [REPS] [KEYWORDS] [name=value ...]
(trailing name=value pairs are engine options, e.g. blocks_per_cu=7 queries_per_pass=16)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402

opts = [a for a in sys.argv[1:] if "=" in a]
args = [a for a in sys.argv[1:] if "=" not in a]
rows, nbytes, nq, k = (int(x) for x in args[0:4])
reps = int(args[4]) if len(args) > 4 else 5
kw = int(args[5]) if len(args) > 5 else 1
eng = HipEngine(0)
for o in opts:
    name, value = o.split("=")
    eng.set_option(name, int(value))
t = eng.open_table(_lib.METRIC_HAMMING, kw, nbytes)
t.add_synthetic(nbytes, rows, 7)
q = np.random.default_rng(0).integers(0, 2**64, size=(nq, t.max_words), dtype=np.uint64)
t.search(q, None, k)
t0 = time.perf_counter()
for _ in range(reps):
    t.search(q, None, k)
dt = (time.perf_counter() - t0) / reps
print(f"{rows} x {nbytes*8}-bit, nq={nq}, k={k} {' '.join(opts)}: {dt*1e3:.3f} ms/call, {nq/dt:.0f} qps")
eng.close()
