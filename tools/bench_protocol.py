#!/usr/bin/env python3
"""BASELINE config 1 through the real backend: 10 000 random 64-bit units via the full IsccIndexProtocol
(HipIndexManager on the GPU): assets/s for add_assets, searches/s for search_assets (one thread and 16 threads)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_asset  # noqa: E402
from iscc_search_amd.index import HipIndexManager, normalize_query  # noqa: E402
from iscc_search_amd.schema import IsccIndex, IsccQuery  # noqa: E402

rng = np.random.default_rng(0)
n_assets = int(sys.argv[1]) if len(sys.argv) > 1 else 2500      # x 4 units = 10 000 codes
assets = [make_asset(rng, i) for i in range(n_assets)]
m = HipIndexManager("hip:///")
m.create_index(IsccIndex(name="c1"))
t0 = time.perf_counter()
for i in range(0, n_assets, 500):
    m.add_assets("c1", assets[i : i + 500])
dt = time.perf_counter() - t0
print(f"add_assets: {n_assets} assets ({n_assets * 4} units) in {dt:.2f} s = {n_assets / dt:.0f} assets/s")
queries = [IsccQuery(iscc_code=a.iscc_code) for a in assets[:200]]
m.search_assets("c1", queries[0], limit=10)
t0 = time.perf_counter()
for q in queries:
    r = m.search_assets("c1", q, limit=10)
    assert r.global_matches[0].score == 1.0
dt = time.perf_counter() - t0
print(f"search_assets (4 units per query, limit 10), 1 thread: {len(queries) / dt:.0f} searches/s ({dt / len(queries) * 1e3:.2f} ms each)")


def worker(chunk):
    for q in chunk:
        m.search_assets("c1", q, limit=10)


threads = [threading.Thread(target=worker, args=(queries[i::16],)) for i in range(16)]
t0 = time.perf_counter()
for th in threads:
    th.start()
for th in threads:
    th.join()
dt = time.perf_counter() - t0
print(f"search_assets, 16 threads: {len(queries) / dt:.0f} searches/s")
m.close()

# py-memory-style: what the reference's memory:// backend does per search -- normalise the query,
# then compare iscc_code strings against EVERY stored asset, score 1.0 (iscc_search/indexes/memory/index.py:204-232).
# One core, GIL-bound by construction; no distance is computed, so this is a plumbing baseline only.
store = {a.iscc_id: a for a in assets}
t0 = time.perf_counter()
for q in queries:
    nq = normalize_query(q)
    types = {u: 1.0 for u in nq.units or []}
    hits = [(a.iscc_id, 1.0, types, a.metadata) for a in store.values() if nq.iscc_code and a.iscc_code and a.iscc_code == nq.iscc_code][:10]
    assert len(hits) == 1
dt = time.perf_counter() - t0
print(f"py-memory-style loop over {n_assets} assets, 1 core: {len(queries) / dt:.0f} searches/s ({dt / len(queries) * 1e3:.2f} ms each)")
