#!/usr/bin/env python3
"""Where a search_assets call through the leader front spends its time (cProfile of the leader; two ranks on one GPU)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_protocol import make_asset  # noqa: E402
from iscc_search_amd.index import HipIndexManager  # noqa: E402
from iscc_search_amd.schema import IsccIndex, IsccQuery  # noqa: E402

uri = sys.argv[1] if len(sys.argv) > 1 else "hip:///?devices=2&backend=gloo&same_gpu=1"
rng = np.random.default_rng(0)
assets = [make_asset(rng, i) for i in range(2500)]
m = HipIndexManager(uri)
m.create_index(IsccIndex(name="c1"))
m.add_assets("c1", assets)
for label, pick in (("META only", lambda a: a.units[:1]), ("META+CONTENT+DATA", lambda a: a.units[:3]), ("INSTANCE only", lambda a: a.units[3:]), ("all four", lambda a: a.units)):
    qs = [IsccQuery(units=pick(a)) for a in assets[:30]]
    m.search_assets("c1", qs[0], limit=10)
    t0 = time.perf_counter()
    for q in qs:
        m.search_assets("c1", q, limit=10)
    print(f"{label}: {(time.perf_counter() - t0) / len(qs) * 1e3:.3f} ms per search_assets")
os.environ["ISCC_PROBE_PHASES"] = "1"
from iscc_search_amd import sharded  # noqa: E402

orig = sharded.ShardedTable._exchange


def timed_exchange(self, q_words, q_nbytes, k, max_hamming, how):
    t0 = time.perf_counter()
    out = orig(self, q_words, q_nbytes, k, max_hamming, how)
    print(f"[leader] exchange nq {q_words.shape[0]} k {k} radius {max_hamming}: {(time.perf_counter() - t0) * 1e3:.3f} ms")
    return out


sharded.ShardedTable._exchange = timed_exchange
for a in assets[:3]:
    m.search_assets("c1", IsccQuery(units=a.units), limit=10)
sharded.ShardedTable._exchange = orig
qs = [IsccQuery(units=a.units[:1]) for a in assets[:10]]
pr = cProfile.Profile()
pr.enable()
for q in qs:
    m.search_assets("c1", q, limit=10)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
m.close()
