#!/usr/bin/env python3
"""Where the time of one 1 024-query step goes on the host: Python wrapper, C call, device (HIP events)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes  # noqa: E402

from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine, _alloc_out  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = 10
eng = HipEngine(0)
t = eng.open_table(_lib.METRIC_HAMMING, 1, 8)
t.add_synthetic(8, rows, 1)
q = np.random.default_rng(1).integers(0, 2**64, size=(nq, 1), dtype=np.uint64)
for _ in range(30):
    t.search(q, None, k)
N = 50
t0 = time.perf_counter()
for _ in range(N):
    t.search(q, None, k)
whole = (time.perf_counter() - t0) / N
out, addr = _alloc_out(nq, k, 1)
qn = t._nbytes(None, nq)
qw = t._words(q)
t0 = time.perf_counter()
for _ in range(N):
    eng._lib.isccsearch_search(eng.handle, t.id, nq, _lib.ptr(qw), _lib.ptr(qn), k, *addr)
ccall = (time.perf_counter() - t0) / N
eng.stats(reset=True)
eng.set_option("profile", 1)
for _ in range(N):
    t.search(q, None, k)
eng.set_option("profile", 0)
st = eng.stats(reset=True)
scan = (st["scan_ms"] + st["level_ms"]) / N
print("rows %d, %d queries: python call %.1f us, C call alone %.1f us (wrapper %.1f us), scan launches %.1f us -> everything else inside the C call %.1f us" % (
    rows, nq, whole * 1e6, ccall * 1e6, (whole - ccall) * 1e6, scan * 1e3, ccall * 1e6 - scan * 1e3))
