#!/usr/bin/env python3
"""Where a search_assets call through the leader front spends its time on the leader (two ranks on one GPU): phase timers."""
import collections
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_protocol import make_asset  # noqa: E402
from iscc_search_amd import shard_front, sharded  # noqa: E402
from iscc_search_amd.index import HipIndexManager  # noqa: E402
from iscc_search_amd.schema import IsccIndex, IsccQuery  # noqa: E402

uri = sys.argv[1] if len(sys.argv) > 1 else "hip:///?devices=2&backend=gloo&same_gpu=1"
rng = np.random.default_rng(0)
assets = [make_asset(rng, i) for i in range(2500)]
m = HipIndexManager(uri)
m.create_index(IsccIndex(name="c1"))
m.add_assets("c1", assets)
acc = collections.defaultdict(float)


def wrap(obj, name, label):
    inner = getattr(obj, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return inner(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t0

    setattr(obj, name, timed)


import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

wrap(sharded.HipShardOps, "local_search", "local_search (async enqueue)")
wrap(sharded.HipShardOps, "merge_strided", "merge_strided (launch + sync + unpack)")
wrap(sharded.HipShardOps, "merge", "merge (launch + sync + unpack)")
wrap(dist, "all_gather_into_tensor", "all_gather_into_tensor")
wrap(torch, "cat", "torch.cat")
wrap(shard_front.Channel, "send", "request pipe write")
wrap(shard_front.LeaderEngine, "run", "LeaderEngine.run (whole operation)")
for label, pick in (("1 unit", lambda a: a.units[:1]), ("4 units", lambda a: a.units)):
    qs = [IsccQuery(units=pick(a)) for a in assets[:100]]
    m.search_assets("c1", qs[0], limit=10)
    acc.clear()
    t0 = time.perf_counter()
    for q in qs:
        m.search_assets("c1", q, limit=10)
    total = (time.perf_counter() - t0) / len(qs)
    print(f"{label}: {total * 1e3:.3f} ms per search_assets")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"    {k:45s} {v / len(qs) * 1e6:8.1f} us")
m.close()
