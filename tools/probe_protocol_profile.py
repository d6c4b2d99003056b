#!/usr/bin/env python3
"""cProfile of HipIndexManager.search_assets (4 units, limit 10) over 10 000 units: where the host's share of a request goes."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_protocol import make_asset  # noqa: E402
from iscc_search_amd.index import HipIndexManager  # noqa: E402
from iscc_search_amd.schema import IsccIndex, IsccQuery  # noqa: E402

rng = np.random.default_rng(0)
assets = [make_asset(rng, i) for i in range(2500)]
m = HipIndexManager("hip:///")
m.create_index(IsccIndex(name="c1"))
m.add_assets("c1", assets)
qs = [IsccQuery(iscc_code=a.iscc_code) for a in assets[:300]]
for q in qs[:20]:
    m.search_assets("c1", q, limit=10)
t0 = time.perf_counter()
for q in qs:
    m.search_assets("c1", q, limit=10)
print(f"{(time.perf_counter() - t0) / len(qs) * 1e6:.1f} us per search_assets")
pr = cProfile.Profile()
pr.enable()
for q in qs:
    m.search_assets("c1", q, limit=10)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
m.close()
