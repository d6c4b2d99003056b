#!/usr/bin/env python3
"""
Accounting of the candidate path of the matrix-core scan (VERDICT r3 item 4 -> profiles/r04_candidate_path.txt).

Per case (rows, code bits, queries, k): the scan time of a top-k step (HIP events around the scan launches, inside the engine),
the candidates the scan appended per query (engine statistic `candidates`, read back while profiling), the same table scanned
under k = 1 (the floor: the same rows and queries with next to no candidates), hence the time per candidate; and a sweep of
range-limited searches (ONE scan launch under a given radius: the number of candidates is a free parameter there and the
thresholds never move -- no distance counts, one returned atomic per candidate).

usage (GPU box): python tools/probe_candidate_path.py [case ...]     case = rows:bits:queries:k
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402

CASES = ["10000000:64:512:400", "10000000:128:512:400", "100000000:64:1024:100", "100000000:64:1024:1000"]


def main():
    cases = sys.argv[1:] or CASES
    engine = HipEngine(0)
    for item in filter(None, os.environ.get("ISCC_HIP_OPTS", "").split(",")):
        engine.set_option(item.split("=")[0].strip(), int(item.split("=")[1]))
    rng = np.random.default_rng(7)
    tables = {}
    for case in cases:
        rows, bits, nq, k = (int(x) for x in case.split(":"))
        key = (rows, bits)
        if key not in tables:
            for t in tables.values():
                t.drop()
            tables.clear()
            t = tables[key] = engine.open_table(_lib.METRIC_HAMMING, 2 if bits != 64 or rows <= 10_000_000 else 1, bits // 8)
            t.add_synthetic(bits // 8, rows, 12345)
        t = tables[key]
        q = rng.integers(0, 2**64, size=(nq, t.max_words), dtype=np.uint64)

        def measure(fn, reps=8):
            for _ in range(4):
                fn()
            engine.stats(reset=True)
            engine.set_option("profile", 1)
            engine.set_option("count_candidates", 1)
            t0 = time.perf_counter()
            for _ in range(reps):
                out = fn()
            wall = (time.perf_counter() - t0) / reps
            engine.set_option("profile", 0)
            engine.set_option("count_candidates", 0)
            st = engine.stats(reset=True)
            launches = st["scan_launches"] + st["level_launches"]
            return {"scan_ms": (st["scan_ms"] + st["level_ms"]) / reps, "launches": launches / reps, "wall_ms": wall * 1e3,
                    "cand_per_query": st["candidates"] / max(1, st["candidate_batches"]) / nq, "hits": st["spec_hits"], "misses": st["spec_misses"],
                    "out": out}

        print(f"== {rows} x {bits}-bit, {nq} queries, k = {k}")
        floor = measure(lambda: t.search(q, None, 1))
        top = measure(lambda: t.search(q, None, k))
        kth = int(top["out"][1][:, k - 1].max())
        extra_us = (top["scan_ms"] - floor["scan_ms"]) * 1e3
        total_cand = top["cand_per_query"] * nq
        print(f"   top-k step : scan {top['scan_ms']:.3f} ms in {top['launches']:.1f} launches (wall {top['wall_ms']:.3f} ms incl. the read-back of this probe), "
              f"{top['cand_per_query']:.0f} candidates per query = {top['cand_per_query'] / k:.2f} k; worst k-th distance {kth}; hinted steps {top['hits']}, misses {top['misses']}")
        print(f"   k = 1 floor: scan {floor['scan_ms']:.3f} ms, {floor['cand_per_query']:.0f} candidates per query")
        if total_cand > 0:
            print(f"   => {extra_us:.0f} us beyond the floor for {total_cand:.0f} candidates = {extra_us * 1e3 / total_cand:.2f} ns of launch time per candidate"
                  f" ({total_cand / max(extra_us, 1e-9):.0f} candidates per us)")
        radii = [int(x) for x in os.environ["PROBE_RADII"].split(",")] if os.environ.get("PROBE_RADII") else sorted({max(0, kth - 2), kth, kth + 2})
        for radius in radii:
            r = measure(lambda: t.search_within(q, None, min(_lib.MAX_K, 4 * k), radius), reps=5)
            tc = r["cand_per_query"] * nq
            ex = (r["scan_ms"] - floor["scan_ms"]) * 1e3
            print(f"   radius {radius:3d} (fixed threshold, no counts): scan {r['scan_ms']:.3f} ms, {r['cand_per_query']:.0f} candidates per query"
                  + (f", {ex * 1e3 / tc:.2f} ns per candidate beyond the floor" if tc > 0 else ""))
    engine.close()


if __name__ == "__main__":
    main()
