import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from iscc_search_amd.engine import HipEngine
from oracle_engine import OracleTable

eng = HipEngine(0)
rng = np.random.default_rng(5)
for n in (7, 300, 5000, 40000):
    for tq in (8, 16):
        for nq in (76, 1024, 1100, 2100):
            eng.set_option("queries_per_pass", tq)
            t = eng.open_table(0, 1, 8)
            model = OracleTable(0, 1, 8)
            words = rng.integers(0, 2**64, size=(n, 1), dtype=np.uint64)
            keys = np.arange(1, n + 1, dtype=np.uint64)
            t.add(keys, words); model.add(keys, words)
            q = rng.integers(0, 2**64, size=(nq, 1), dtype=np.uint64)
            got = t.search(q, None, 10); exp = model.search(q, None, 10)
            bad = np.nonzero(got[3] != exp[3])[0]
            badk = np.nonzero((got[0] != exp[0]).any(axis=1))[0]
            print(f"n={n} tq={tq} nq={nq}: count mismatches {len(bad)} {bad[:5]}..{bad[-3:] if len(bad) else ''} key mismatches {len(badk)} {badk[:5]}")
            t.drop()
eng.close()
