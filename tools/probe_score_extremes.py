#!/usr/bin/env python3
"""The largest simprint request the library takes (8 192 query simprints, limit 204 -> 4 080 neighbours each) scored on the device
against the host scoring of the same lists; plus limit beyond MAX_K / oversampling (radius lists).  usage (GPU box): python tools/probe_score_extremes.py [rows]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import flip_bits  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402
from iscc_search_amd.simprint import HipSimprintIndex, pack_chunk_pointer  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(5)
eng = HipEngine(0)
idx = HipSimprintIndex(eng, ndim=128)
pool = [rng.integers(0, 256, size=16, dtype=np.uint8).tobytes() for _ in range(4000)]
keys, vecs = [], []
for i in range(rows):
    a, c = divmod(i, 20)
    keys.append(pack_chunk_pointer((a + 1).to_bytes(8, "big"), c * 7, 7))
    base = pool[int(rng.integers(0, len(pool)))] if i % 3 == 0 else rng.integers(0, 256, size=16, dtype=np.uint8).tobytes()
    vecs.append(np.frombuffer(flip_bits(base, int(rng.integers(0, 6))) if i % 3 == 0 else base, dtype=np.uint8))
t0 = time.perf_counter()
idx.add_raw(keys, vecs)
print(f"{rows} chunks added in {time.perf_counter() - t0:.1f} s", flush=True)
for nq, limit in ((8192, 204), (8192, 20), (100, 300), (8192, 1)):
    simprints = [flip_bits(pool[i % len(pool)], i % 5) for i in range(nq)]
    for dev_freq in (True, False):
        t0 = time.perf_counter()
        a = idx.search_raw(simprints, limit=limit, threshold=0.9, detailed=True, total_assets=rows // 20, device_doc_freq=dev_freq)
        t1 = time.perf_counter()
        b = idx._search_raw_host(simprints, limit, 0.9, True, None, rows // 20, dev_freq)
        t2 = time.perf_counter()
        same = len(a) == len(b) and all(x.iscc_id_body == y.iscc_id_body and x.score == y.score and x.matches == y.matches and
                                        [(c.query, c.match, c.score, c.offset, c.size, c.freq) for c in x.chunks] == [(c.query, c.match, c.score, c.offset, c.size, c.freq) for c in y.chunks]
                                        for x, y in zip(a, b))
        print(f"nq={nq} limit={limit} device_doc_freq={dev_freq}: device {1e3 * (t1 - t0):.1f} ms, host {1e3 * (t2 - t1):.1f} ms, {len(a)} assets, same={same}", flush=True)
        assert same
idx.close()
eng.close()
