#!/usr/bin/env python3
"""Where does a range-limited search spend its time?  Wall clock against the scan launches' HIP events, per radius and batch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(0)
for item in filter(None, os.environ.get("ISCC_HIP_OPTS", "").split(",")):
    eng.set_option(item.split("=")[0].strip(), int(item.split("=")[1]))
t = eng.open_table(_lib.METRIC_HAMMING, 1, 8)
t.add_synthetic(8, int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 1)
_, cols = t.export_rows(8, 12345, 1024)
stored = cols.T.copy()
for nq in (1, 8, 16, 64):
    for r, k in ((0, 1000), (4, 1000), (0, 10), (4, 10)):
        q = stored[:nq]
        t.search_within(q, None, k, r)
        eng.stats(reset=True)
        eng.set_option("profile", 1)
        t0 = time.perf_counter()
        for _ in range(5):
            out = t.search_within(q, None, k, r)
        dt = (time.perf_counter() - t0) / 5
        eng.set_option("profile", 0)
        st = eng.stats(reset=True)
        print("nq=%3d r=%d k=%4d: %7.3f ms/call; scan %7.3f ms in %4.1f launches per call (mfma %d); fallbacks %d; results/query %.1f" % (
            nq, r, k, dt * 1e3, st["scan_ms"] / 5, st["scan_launches"] / 5, st["scan_mfma_launches"], st["fallback_queries"], out[3].mean()))
t.drop()
eng.close()
