#!/usr/bin/env python3
"""
What does a candidate cost the matrix-core scan, and does the time of day inside a step matter?

Range-limited searches run ONE scan launch over the whole table under a threshold given by the caller, so the number of
candidates per launch is a free parameter: radius r meets ~ rows * P(binomial(64, 1/2) <= r) candidates per random query.
Prints the scan time per launch (HIP events inside the engine) for a sweep of radii, then the per-launch times of a top-k
search with different level growths.

usage (GPU box): python tools/probe_candidates.py [rows] [queries]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    engine = HipEngine(0)
    table = engine.open_table(_lib.METRIC_HAMMING, 1, 8)
    table.add_synthetic(8, rows, 12345)
    rng = np.random.default_rng(7)
    q = rng.integers(0, 1 << 63, size=(nq, 1), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(nq, 1), dtype=np.uint64)

    def timed(fn, reps):
        for _ in range(3):
            fn()
        engine.stats(reset=True)
        engine.set_option("profile", 1)
        for _ in range(reps):
            out = fn()
        engine.set_option("profile", 0)
        return engine.stats(reset=True), out

    print("rows %d, queries %d" % (rows, nq))
    for radius in (0, 6, 10, 11, 12, 13, 14, 15):
        st, out = timed(lambda: table.search_within(q, None, 10, radius), 10)
        found = int(out[3].sum())
        print("radius %2d: %8.1f us per scan launch (%d launches, mfma %d); results/query %.2f" % (
            radius, 1e3 * st["scan_ms"] / st["scan_launches"], st["scan_launches"], st["scan_mfma_launches"], found / nq))
    for growth in (2, 4, 16, 64, 256):
        engine.set_option("mfma_level_growth", growth)
        st, out = timed(lambda: table.search(q, None, 10), 10)
        print("growth %3d: levels %6.1f us in %4.1f launches, collect %6.1f us in %3.1f launches, per step; level rows/us %.1f collect rows/us %.1f" % (
            growth, 1e3 * st["level_ms"] / 10, st["level_launches"] / 10, 1e3 * st["scan_ms"] / 10, st["scan_launches"] / 10,
            st["level_pair_words"] / nq / max(1e-9, 1e3 * st["level_ms"]), st["scan_pair_words"] / nq / max(1e-9, 1e3 * st["scan_ms"])))
    engine.close()


if __name__ == "__main__":
    main()
