#!/usr/bin/env python3
"""
BASELINE config 5 end to end at its stated size: 10 M chunk fingerprints per table (``usearch/index.py:1746-1763`` keeps one
fixed-``ndim`` table per simprint type), 128-bit chunk-pointer keys, for ndim = 64 / 128 / 256:

  * ``search_raw``  -- the approximate path of ``iscc_search/indexes/simprint/usearch_core.py:137-269`` on
    ``HipSimprintIndex``: ONE library call (``isccsearch_simprint_score``: batched exact Hamming search with count = limit x 20,
    threshold, best chunk per asset and query, IDF-weighted scoring, sort and cut on the device), and beside it round 3's shape
    of the same request (``_search_raw_host``: the neighbour lists cross PCIe and are scored by Python), which must return
    the same assets with the same float64 scores;
  * ``search_exact`` -- the hard-boundary collision search (``lmdb_ops.py:169-301``): range-limited lookups at distance 0.

Every time is split into the DEVICE share (wall time inside the C-ABI calls: search, vector gather, frequency column) and the
HOST share (everything else: query packing, match filtering, per-asset aggregation, scoring, object construction).

usage (GPU box): python tools/bench_simprint.py [--raw-only] [chunks per table, default 10000000] [ndim ...]
   run under `rocprofv3 --kernel-trace --stats` with --raw-only (search_raw through the library alone) for the per-kernel
   table committed as profiles/r04_kernel_stats_simprint.csv
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iscc_search_amd.engine import HipEngine  # noqa: E402
from iscc_search_amd.simprint import HipSimprintIndex  # noqa: E402

CHUNKS_PER_ASSET = 40


class DeviceClock:
    """Wall time spent inside the index's device entry points (each ends with the results on the host)."""

    NAMES = ("search_arrays", "search_within", "get_many", "get_freq", "doc_freq", "score_assets", "exact_assets")

    def __init__(self, index):
        self.seconds = 0.0
        for name in self.NAMES:
            inner = getattr(index, name)

            def timed(*a, _inner=inner, **kw):
                t0 = time.perf_counter()
                try:
                    return _inner(*a, **kw)
                finally:
                    self.seconds += time.perf_counter() - t0

            setattr(index, name, timed)

    def take(self):
        s, self.seconds = self.seconds, 0.0
        return s


def build(eng, ndim, n_chunks, rng):
    idx = HipSimprintIndex(eng, ndim=ndim)
    nb = ndim // 8
    first = None
    t0 = time.perf_counter()
    for lo in range(0, n_chunks, 1 << 20):
        n = min(1 << 20, n_chunks - lo)
        rows = np.arange(lo, lo + n, dtype=np.uint64)
        keys = np.stack([rows // np.uint64(CHUNKS_PER_ASSET) + np.uint64(1),                          # asset body (big-endian value)
                         ((rows % np.uint64(CHUNKS_PER_ASSET)) * np.uint64(100) << np.uint64(32)) | np.uint64(100)], axis=1)   # offset | size
        vecs = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
        idx._index.add(keys, vecs, trusted_unique=True)
        if first is None:
            first = vecs[: CHUNKS_PER_ASSET * 16].copy()
    return idx, first, time.perf_counter() - t0


def main():
    raw_only = "--raw-only" in sys.argv
    args = [a for a in sys.argv[1:] if a != "--raw-only"]
    n_chunks = int(args[0]) if args else 10_000_000
    ndims = [int(a) for a in args[1:]] or [64, 128, 256]
    eng = HipEngine(0)
    for item in filter(None, os.environ.get("ISCC_HIP_OPTS", "").split(",")):      # e.g. ISCC_HIP_OPTS=mfma=0
        eng.set_option(item.split("=")[0].strip(), int(item.split("=")[1]))
    rng = np.random.default_rng(0)
    n_assets = n_chunks // CHUNKS_PER_ASSET
    for ndim in ndims:
        idx, first, secs = build(eng, ndim, n_chunks, rng)
        print(f"== ndim {ndim}: {idx.size} chunks of {n_assets} assets ingested in {secs:.1f} s")
        clock = DeviceClock(idx._index)
        for nq in (16, 64, 256, 512):
            q = first[:nq].copy()
            q[:, 0] ^= 3                                     # two flipped bits: an approximate, not an exact, match
            simprints = [bytes(r) for r in q]
            kw = dict(limit=20, threshold=0.75, detailed=True, total_assets=n_assets, device_doc_freq=True)
            idx.search_raw(simprints, **kw)                  # first call: frequency column, pinned buffers
            clock.take()
            reps = 9

            def timed(fn):
                """median over `reps` calls of (total, device share) -- and the slowest call: buffers still growing, allocator, GC"""
                rows = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    res = fn()
                    total = time.perf_counter() - t0
                    rows.append((total, clock.take()))
                rows.sort()
                return rows[reps // 2][0], rows[reps // 2][1], rows[-1][0], res

            total, dev, worst, res = timed(lambda: idx.search_raw(simprints, **kw))
            print(f"search_raw   nq={nq:3d} count=400: {total * 1e3:7.2f} ms = device {dev * 1e3:6.2f} + host {(total - dev) * 1e3:6.2f} (slowest of {reps}: {worst * 1e3:.2f}); "
                  f"{len(res)} assets, top score {res[0].score:.4f}")
            if raw_only:
                continue
            if nq <= 512 and (ndim != 64 or nq <= 64):       # (64-bit random fingerprints: 200 000 matched assets, seconds of Python)
                host_kw = dict(limit=20, threshold=0.75, detailed=True, doc_freq_fn=None, total_assets=n_assets, device_doc_freq=True)
                total_h, dev_h, worst_h, res_h = timed(lambda: idx._search_raw_host(simprints, **host_kw))
                same = [(r.iscc_id_body, r.score, r.matches) for r in res] == [(r.iscc_id_body, r.score, r.matches) for r in res_h]
                print(f"  scored on the host   : {total_h * 1e3:7.2f} ms = device {dev_h * 1e3:6.2f} + host {(total_h - dev_h) * 1e3:6.2f}; same assets and scores: {same}")
            exact_q = [bytes(r) for r in first[:nq]]
            idx.search_exact(exact_q, limit=20, threshold=0.0, detailed=True)
            clock.take()
            total, dev, worst, res = timed(lambda: idx.search_exact(exact_q, limit=20, threshold=0.0, detailed=True))
            print(f"search_exact nq={nq:3d}          : {total * 1e3:7.2f} ms = device {dev * 1e3:6.2f} + host {(total - dev) * 1e3:6.2f} (slowest of {reps}: {worst * 1e3:.2f}); {len(res)} assets")
        idx.close()
    eng.close()


if __name__ == "__main__":
    main()
