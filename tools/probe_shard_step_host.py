#!/usr/bin/env python3
"""
Host time of one sharded step (ShardedTable.search with the collective enabled on one rank): a table small enough that the GPU work is
tens of microseconds, so the step time IS the fixed cost; cProfile of the same loop beside it.
usage (GPU box): python tools/probe_shard_step_host.py [rows] [queries]
"""
import cProfile
import os
import pstats
import socket
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402
from iscc_search_amd.sharded import HipShardOps, ShardedTable  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
eng = HipEngine(0)
t = eng.open_table(_lib.METRIC_HAMMING, 1, 8)
t.add_synthetic(8, rows, 1)
sh = ShardedTable(HipShardOps(t, torch.device("cuda", 0)), always_gather=True)
rng = np.random.default_rng(0)
batches = [rng.integers(0, 2**64, size=(nq, 1), dtype=np.uint64) for _ in range(8)]
for i in range(200):
    sh.search(batches[i % 8], None, 10)
N = 2000
t0 = time.perf_counter()
for i in range(N):
    sh.search(batches[i % 8], None, 10)
dt = (time.perf_counter() - t0) / N
print(f"sharded step, {rows} rows x {nq} queries, collective on one rank: {dt * 1e6:.1f} us per step")
t0 = time.perf_counter()
for i in range(N):
    t.search(batches[i % 8], None, 10)
print(f"unsharded call (isccsearch_search): {(time.perf_counter() - t0) / N * 1e6:.1f} us per call")
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    sh.search(batches[i % 8], None, 10)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
dist.destroy_process_group()
