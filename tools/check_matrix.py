"""Parity matrix on the GPU: every code length class x queries-per-pass x batch size against the oracle-backed model
(kept from the hunt for the stale-SGPR-base bug; prints only mismatches, then `done`)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from iscc_search_amd.engine import HipEngine
from oracle_engine import OracleTable

eng = HipEngine(0)
rng = np.random.default_rng(5)
n = 40000
for nbytes in (8, 16, 24, 32, 20):
    mw = (nbytes + 7) // 8
    words = rng.integers(0, 2**64, size=(n, mw), dtype=np.uint64)
    if nbytes % 8:
        words[:, -1] &= np.uint64((~0 << (8 * (8 - nbytes % 8))) & (2**64 - 1))
    keys = np.arange(1, n + 1, dtype=np.uint64)
    model = OracleTable(0, 1, nbytes); model.add(keys, words)
    for tq in (8, 10, 12, 16):
        for nq in (9, 160, 1100):
            eng.set_option("queries_per_pass", tq)
            t = eng.open_table(0, 1, nbytes)
            t.add(keys, words)
            q = rng.integers(0, 2**64, size=(nq, mw), dtype=np.uint64)
            if nbytes % 8:
                q[:, -1] &= np.uint64((~0 << (8 * (8 - nbytes % 8))) & (2**64 - 1))
            print(f"about to run nbytes={nbytes} tq={tq} nq={nq}", flush=True)
            got = t.search(q, None, 10); exp = model.search(q, None, 10)
            bad = np.nonzero((got[3] != exp[3]) | (got[0] != exp[0]).any(axis=1))[0]
            if len(bad):
                print(f"nbytes={nbytes} tq={tq} nq={nq}: {len(bad)} bad queries, first {bad[:6]}, counts {got[3][bad[:6]]}")
            t.drop()
print("done")
eng.close()
