#!/usr/bin/env python3
"""Ingest rate of the C-ABI: isccsearch_add from host arrays (the reference pays an HNSW insert per vector here,
`usearch/index.py:440`), remove, and the lazy host key index."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(0)
rng = np.random.default_rng(0)
for nbytes, kw in ((8, 1), (32, 1), (16, 2)):
    n, batch = 20_000_000, 2_000_000
    t = eng.open_table(_lib.METRIC_HAMMING, kw, nbytes)
    mw = t.max_words
    words = rng.integers(0, 2**64, size=(batch, mw), dtype=np.uint64)
    t0 = time.perf_counter()
    for b in range(n // batch):
        if kw == 2:
            keys = np.stack([np.full(batch, b, dtype=np.uint64), np.arange(batch, dtype=np.uint64)], axis=1)
        else:
            keys = np.arange(b * batch, (b + 1) * batch, dtype=np.uint64)
        t.add(keys, words, None, trusted_unique=True)
    dt = time.perf_counter() - t0
    print(f"add: {n} x {nbytes*8}-bit rows ({kw*64}-bit keys) in batches of {batch}: {dt:.2f} s = {n/dt/1e6:.1f} M rows/s ({n*(nbytes+8*kw)/dt/1e9:.2f} GB/s host->device incl. column split)")
    t0 = time.perf_counter()
    probe = keys[:1000]
    found = t.contains(probe)
    dt = time.perf_counter() - t0
    print(f"  first contains() (builds the host key index over {n} rows): {dt:.2f} s, all found: {bool(found.all())}")
    t0 = time.perf_counter()
    removed = t.remove(keys[: batch // 2])
    dt = time.perf_counter() - t0
    print(f"  remove {removed} rows: {dt:.2f} s = {removed/dt/1e6:.2f} M rows/s")
    t.drop()
eng.close()
