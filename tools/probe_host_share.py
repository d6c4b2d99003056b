#!/usr/bin/env python3
"""Host share of a simprint-sized search (10 M x 128-bit rows, 512 queries, k = 400): the Python wrapper against the bare C call."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine, _alloc_out  # noqa: E402

eng = HipEngine(0)
t = eng.open_table(_lib.METRIC_HAMMING, 1, 16)
t.add_synthetic(16, 10_000_000, seed=1)
rng = np.random.default_rng(3)
nq, k = 512, 400
q = rng.integers(0, 2**63, size=(nq, 2), dtype=np.uint64)
for _ in range(30):
    t.search(q, None, k)
eng.set_option("profile", 1)
s0 = eng.stats(reset=True)
t0 = time.perf_counter()
for _ in range(50):
    t.search(q, None, k)
wrap = (time.perf_counter() - t0) / 50
s1 = eng.stats()
out, addr = _alloc_out(nq, k, 1)
lib = eng._lib
t0 = time.perf_counter()
for _ in range(50):
    _lib.check(lib.isccsearch_search(eng.handle, t.id, nq, _lib.ptr(q), None, k, *addr))
bare = (time.perf_counter() - t0) / 50
t0 = time.perf_counter()
for _ in range(50):
    _alloc_out(nq, k, 1)
alloc = (time.perf_counter() - t0) / 50
print("HipTable.search %.3f ms per call; bare isccsearch_search into the same arrays %.3f ms; _alloc_out %.3f ms; scan launches %.3f ms per call" % (
    wrap * 1e3, bare * 1e3, alloc * 1e3, s1["scan_ms"] / 50))
