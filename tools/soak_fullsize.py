#!/usr/bin/env python3
"""
Differential soak at full size: the single self-tightening pass on the matrix cores against the level design on the matrix cores
and against the XOR + popcount kernels (three independent ways to the same exact top-k), over many random batches of a
100 M-row table -- batch size, k and the share of planted near-duplicates vary per round.  (The oracle comparison at this size is
bench.py's parity gate and tests/test_gpu_fullsize.py; this tool looks for RARE disagreements -- races in the threshold
updates, list overflows -- across many rounds.)

usage (GPU box): python tools/soak_fullsize.py [rounds] [rows] [nbytes]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402

SEED = 0x1511CC00


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    nbytes = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    words = (nbytes + 7) // 8
    engine = HipEngine(0)
    table = engine.open_table(_lib.METRIC_HAMMING, 1, nbytes)
    table.add_synthetic(nbytes, rows, SEED)
    rng = np.random.default_rng(20260)
    t0 = time.perf_counter()
    mismatches = 0
    for rnd in range(rounds):
        nq = int(rng.choice([17, 40, 64, 70, 96, 128, 160, 200, 512, 1000, 1024]))
        k = int(rng.choice([1, 10, 10, 10, 37, 100, 256, 512, 1000, 2048]))
        q = rng.integers(0, 2**64, size=(nq, words), dtype=np.uint64)
        planted = rng.random(nq) < rng.choice([0.0, 0.25, 0.9])
        _, stored = table.export_rows(nbytes, int(rng.integers(0, rows - nq)), nq)
        near = stored.T.copy()
        near[:, words - 1] ^= (np.uint64(1) << rng.integers(0, 4, size=nq).astype(np.uint64)) - np.uint64(1)
        q[planted] = near[planted]
        # extremes of a dot product (round 3: the packed kernel holds two of them per accumulator): all ones, one bit, and --
        # every eighth round -- an all-zero query, which sends a 64-bit batch to the unpacked kernel
        q[0] = ~np.uint64(0)
        q[1] = 0
        q[1, 0] = np.uint64(1) << np.uint64(63)
        if rnd % 8 == 3:
            q[2] = 0
        if nbytes % 8:
            q[:, words - 1] &= ~np.uint64(0) << np.uint64(8 * (8 - nbytes % 8))
        engine.set_option("mfma", 1)
        engine.set_option("self_tighten", 1)
        single = table.search(q, None, k)
        engine.set_option("self_tighten", 0)
        levels = table.search(q, None, k)
        engine.set_option("self_tighten", 1)
        results = {"levels": levels}
        # a repeated batch runs under the hint its first search left: <= 128 queries the speculative range-limited pass, larger
        # ones the single pass started under the hint (and, every other round, queries of which some have near-duplicates: the
        # hint seeded by one kind of batch meets the other)
        results["hinted"] = table.search(q, None, k)
        if words == 1 and rnd % 2 == 1:           # one-word codes: the unpacked matrix-core kernel as a fourth way
            engine.set_option("mfma_pack", 0)
            results["unpacked"] = table.search(q, None, k)
            engine.set_option("mfma_pack", 1)
        if rnd % 4 == 0:                          # the XOR + popcount kernels take ~13 ms per 1 024 queries: every fourth round
            engine.set_option("mfma", 0)
            results["xor"] = table.search(q, None, k)
            engine.set_option("mfma", 1)
        for name, other in results.items():
            for a, b, what in zip(single, other, ("keys", "hamming", "prefix_bits", "count")):
                if not np.array_equal(a, b):
                    mismatches += 1
                    print("MISMATCH round %d (nq %d, k %d): single pass vs %s: %s" % (rnd, nq, k, name, what), flush=True)
        if rnd % 10 == 9:
            print("round %d: %d mismatches so far, %.1f s" % (rnd + 1, mismatches, time.perf_counter() - t0), flush=True)
    st = engine.stats()
    print("soak: %d rounds over %d x %d-bit rows, %d mismatches; fallbacks %d, single-pass retries %d, speculative passes %d hit / %d missed" % (
        rounds, rows, nbytes * 8, mismatches, st["fallback_queries"], st["self_retries"], st["spec_hits"], st["spec_misses"]))
    table.drop()
    engine.close()
    sys.exit(1 if mismatches else 0)


if __name__ == "__main__":
    main()
