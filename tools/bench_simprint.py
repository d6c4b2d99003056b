#!/usr/bin/env python3
"""End-to-end timing of HipSimprintIndex (config 5 shape): approximate search_raw (GPU search + host IDF scoring,
document frequencies from the device), hard-boundary search_exact, and the frequency-column build."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iscc_search_amd.engine import HipEngine  # noqa: E402
from iscc_search_amd.simprint import HipSimprintIndex, pack_chunk_pointer  # noqa: E402

eng = HipEngine(0)
for _item in filter(None, os.environ.get("ISCC_HIP_OPTS", "").split(",")):      # e.g. ISCC_HIP_OPTS=mfma=0
    eng.set_option(_item.split("=")[0].strip(), int(_item.split("=")[1]))
rng = np.random.default_rng(0)
n_assets, chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, 40      # 4 M chunks
ndim = 128
idx = HipSimprintIndex(eng, ndim=ndim)
t0 = time.perf_counter()
bodies = (np.arange(n_assets, dtype=np.uint64) + np.uint64(1)).astype(">u8").view("V8")
for a0 in range(0, n_assets, 10_000):
    a1 = min(n_assets, a0 + 10_000)
    vecs = rng.integers(0, 256, size=((a1 - a0) * chunks, ndim // 8), dtype=np.uint8)
    keys = [pack_chunk_pointer(bytes(bodies[a]), c * 100, 100) for a in range(a0, a1) for c in range(chunks)]
    idx.add_raw(keys, list(vecs))
    if a0 == 0:
        first = vecs[: chunks * 5].copy()
print(f"ingest {idx.size} chunks: {time.perf_counter() - t0:.1f} s")
for nq in (16, 64, 256):
    # query = chunks of the first assets with a few bits flipped
    q = first[:nq].copy()
    q[:, 0] ^= 3
    simprints = [bytes(r) for r in q]
    for limit in (10,):
        for mode, kw in (("callback freq=1", dict(doc_freq_fn=lambda s: 1)), ("device doc freq", dict(device_doc_freq=True)), ("device doc freq", dict(device_doc_freq=True))):
            b0 = eng.stats()["freq_builds"]
            t0 = time.perf_counter()
            res = idx.search_raw(simprints, limit=limit * 2, threshold=0.75, detailed=True, total_assets=n_assets, **kw)
            dt = time.perf_counter() - t0
            built = eng.stats()["freq_builds"] - b0
            print(f"search_raw nq={nq} limit={limit*2} (count={limit*2*20}) {mode}{' [column built]' if built else ''}: "
                  f"{dt*1e3:8.2f} ms, {len(res)} assets, top score {res[0].score:.4f}")
        exact_q = [bytes(r) for r in first[:nq]]
        idx.search_exact(exact_q, limit=limit * 2, threshold=0.0, detailed=True)     # first call grows the pinned result buffers
        t0 = time.perf_counter()
        res = idx.search_exact(exact_q, limit=limit * 2, threshold=0.0, detailed=True)
        print(f"search_exact nq={nq}: {(time.perf_counter() - t0)*1e3:8.2f} ms, {len(res)} assets")
eng.close()
