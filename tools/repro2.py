import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from iscc_search_amd.engine import HipEngine
from oracle_engine import OracleTable
from test_gpu_fuzz import _mask

eng = HipEngine(0)
seed = 0
rng = np.random.default_rng(1000 + seed)
metric = int(rng.integers(0, 2)); key_words = int(rng.integers(1, 3)); max_bytes = int(rng.choice([1, 3, 8, 12, 16, 24, 32]))
mw = (max_bytes + 7) // 8
tq = int(rng.choice([8, 10, 12, 16, 32]))
print("metric", metric, "key_words", key_words, "max_bytes", max_bytes, "tq", tq)
eng.set_option("queries_per_pass", tq)
t = eng.open_table(metric, key_words, max_bytes); model = OracleTable(metric, key_words, max_bytes)
lengths = [max_bytes] if metric == 0 else sorted({max_bytes, max(1, max_bytes // 2), max(1, max_bytes - 3), 1})
bases = rng.integers(0, 2**64, size=(6, mw), dtype=np.uint64)
next_key = 1
for step in range(2):
    op = rng.choice(["add", "add", "remove", "search", "search"])
    print("step", step, op)
    if op == "add" or step == 0:
        n = int(rng.choice([1, 7, 300, 5000, 40000]))
        lens = rng.choice(lengths, size=n).astype(np.uint8)
        words = bases[rng.integers(0, len(bases), size=n)].copy()
        flips = rng.integers(0, 4, size=n)
        for f in range(1, 4):
            sel = flips >= f
            words[sel, 0] ^= np.uint64(1) << rng.integers(0, 64, size=int(sel.sum())).astype(np.uint64)
        if rng.random() < 0.5:
            words = rng.integers(0, 2**64, size=(n, mw), dtype=np.uint64)
        words = _mask(words, lens)
        if key_words == 2:
            keys = np.stack([rng.integers(0, 3, size=n).astype(np.uint64), np.arange(next_key, next_key + n, dtype=np.uint64)], axis=1)
        else:
            keys = np.arange(next_key, next_key + n, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(2**61 - 1)
        next_key += n
        nb = lens if metric == 1 else None
        t.add(keys, words, nb); model.add(keys, words, nb)
        print("added", n, "lens", np.unique(lens, return_counts=True))
    else:
        nq = int(rng.choice([1, 5, 37, 1100])); k = int(rng.choice([1, 10, 100, 1000, 4096]))
        if nq * k > 400_000: k = 10
        qlens = rng.choice(lengths, size=nq).astype(np.uint8) if metric == 1 else None
        q = bases[rng.integers(0, len(bases), size=nq)].copy()
        q[:, 0] ^= rng.integers(0, 16, size=nq).astype(np.uint64)
        q = _mask(q, qlens if metric == 1 else max_bytes)
        print("search nq", nq, "k", k, "qlens", None if qlens is None else np.unique(qlens, return_counts=True))
        got = t.search(q, qlens, k); exp = model.search(q, qlens, k)
        bad = np.nonzero(got[3] != exp[3])[0]
        print("count mismatches", len(bad), bad[:10], "got", got[3][bad[:10]], "exp", exp[3][bad[:10]])
        if qlens is not None: print("qlens of bad", np.unique(qlens[bad], return_counts=True))
        print(eng.stats())
        # the same queries one length class at a time
        if qlens is not None:
            for L in np.unique(qlens):
                sel = np.nonzero(qlens == L)[0]
                g2 = t.search(q[sel], qlens[sel], k); e2 = model.search(q[sel], qlens[sel], k)
                print("class", L, "n", len(sel), "mismatch", int((g2[3] != e2[3]).sum()))
eng.close()
