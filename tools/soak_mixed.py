#!/usr/bin/env python3
"""
Differential soak of the per-table hint on an index of SEVERAL code lengths (4 x `rows` codes of 64 / 128 / 192 / 256 bits in one NPHD
table): two batches of one shape are searched with the hints on -- the first under whatever an earlier round of that shape left, the
second under the first one's, and once more under its own -- and then with the hints off (`speculate = 0`: the ordinary path);
the answers must be identical.  Batch size, query length, k and the share of near-duplicate
queries vary per round.

usage (GPU box): python tools/soak_mixed.py [rounds, default 300] [rows per length, default 25000000]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 25_000_000
    eng = HipEngine(0)
    t = eng.open_table(_lib.METRIC_NPHD, 1, 32)
    for i, nb in enumerate((8, 16, 24, 32)):
        t.add_synthetic(nb, rows, seed=7 + i, first_row=0, key_base=i * rows)
    rng = np.random.default_rng(99)
    mismatches, t0 = 0, time.perf_counter()
    for rnd in range(rounds):
        nq = int(rng.choice([1, 1, 3, 8, 40, 128, 200, 700]))
        nb = int(rng.choice([8, 16, 24, 32]))
        k = int(rng.choice([1, 10, 10, 50, 300]))
        w = nb // 8

        def batch():
            q = rng.integers(0, 2**64, size=(nq, 4), dtype=np.uint64)
            if rng.random() < 0.5:                   # near-duplicates of stored codes of a random length
                src = int(rng.choice([8, 16, 24, 32]))
                _, stored = t.export_rows(src, int(rng.integers(0, rows - nq)), nq)
                near = np.zeros((nq, 4), dtype=np.uint64)
                near[:, : src // 8] = stored.T
                near[:, 0] ^= np.uint64(1) << rng.integers(0, 64, size=nq).astype(np.uint64)
                pick = rng.random(nq) < 0.5
                q[pick] = near[pick]
            q[:, w:] = 0
            return q

        ql = np.full(nq, nb, dtype=np.uint8)
        q1, q2 = batch(), batch()
        eng.set_option("speculate", 1)
        hinted = [t.search(q1, ql, k), t.search(q2, ql, k), t.search(q2, ql, k)]     # under the hint of: an earlier round / q1's batch / its own
        eng.set_option("speculate", 0)
        plain = [t.search(q1, ql, k), t.search(q2, ql, k)]
        for name, got, exp in (("first", hinted[0], plain[0]), ("second", hinted[1], plain[1]), ("repeated", hinted[2], plain[1])):
            for a_, b_, what in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
                if not np.array_equal(a_, b_):
                    mismatches += 1
                    print("MISMATCH round %d (nq %d, %d bits, k %d): %s hinted search vs the ordinary path: %s" % (rnd, nq, nb * 8, k, name, what), flush=True)
        if rnd % 50 == 49:
            print("round %d: %d mismatches so far, %.1f s" % (rnd + 1, mismatches, time.perf_counter() - t0), flush=True)
    st = eng.stats()
    print("soak (mixed lengths): %d rounds over 4 x %d rows, %d mismatches; fallbacks %d, single-pass retries %d, hints %d held / %d did not" % (
        rounds, rows, mismatches, st["fallback_queries"], st["self_retries"], st["spec_hits"], st["spec_misses"]))
    sys.exit(1 if mismatches else 0)


if __name__ == "__main__":
    main()
