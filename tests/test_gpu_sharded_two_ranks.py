"""
Two PROCESSES, each with its own engine and its own row-range shard on the (one) GPU of the box, driving the product
``ShardedTable`` + ``HipShardOps``: local device search, exchange, device merge.  The exchange runs over gloo (ranks
that share a GPU cannot form an RCCL communicator), staged through the host by ``ShardedTable``; everything else is
the code path of the multi-GPU deployment.  Every rank must hold exactly the ORACLE's answer over the unsharded rows.
"""

import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROWS, SEED, NQ, K = 3_000_000, 0x1511CC00, 24, 10


def _queries():
    rng = np.random.default_rng(5)
    return rng.integers(0, 2**64, size=(NQ, 1), dtype=np.uint64)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from iscc_search_amd.engine import HipEngine
    from iscc_search_amd.sharded import HipShardOps, ShardedTable, shard_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    engine = HipEngine(0)
    try:
        lo, hi = shard_range(ROWS, rank, world)
        table = engine.open_table(0, 1, 8)
        table.add_synthetic(8, hi - lo, SEED, first_row=lo)
        sharded = ShardedTable(HipShardOps(table, "cuda:0"))
        q = _queries()
        out = {}
        for i, a in enumerate(sharded.search(q, None, K)):
            out[f"s{i}"] = a
        for i, a in enumerate(sharded.search_within(q, None, 50, 14)):
            out[f"w{i}"] = a
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), **out)
    finally:
        engine.close()
        dist.destroy_process_group()


def test_two_processes_two_shards_one_gpu(tmp_path):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from oracle import np_within, oracle_splitmix64_fill, oracle_topk

    words = oracle_splitmix64_fill(ROWS, SEED, stride=4).reshape(ROWS, 1)
    row_keys = np.arange(ROWS, dtype=np.uint64)
    q = _queries()
    want_s = oracle_topk(0, row_keys, words, None, q, None, K)
    want_w = (np.zeros((NQ, 50), np.uint64), np.zeros((NQ, 50), np.uint32), np.zeros((NQ, 50), np.uint16), np.zeros(NQ, np.uint32))
    for i in range(NQ):
        kk, h, p = np_within(words, 8, row_keys, q[i], 8, 50, 14)
        c = len(h)
        want_w[0][i, :c], want_w[1][i, :c], want_w[2][i, :c], want_w[3][i] = kk, h, p, c
    for rank in range(2):
        with np.load(os.path.join(tmp_path, f"r{rank}.npz")) as z:
            for i in range(4):
                np.testing.assert_array_equal(z[f"s{i}"], want_s[i], err_msg=f"rank {rank} search field {i}")
                np.testing.assert_array_equal(z[f"w{i}"], want_w[i], err_msg=f"rank {rank} within field {i}")
    assert int(want_w[3].sum()) > 0          # the radius admits rows, so the within lists are not trivially empty


def _manager_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"          # both ranks share the box's one GPU
    import json

    import torch.distributed as dist

    from iscc_search_amd.index import HipIndexManager
    from test_sharded_manager_gloo import scenario

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        uri = f"hip://{out_dir}/store?devices={world}"
        out = scenario(HipIndexManager(uri), reopen=lambda: HipIndexManager(uri))       # the product engine: HipEngine + ShardedEngine
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
            json.dump(out, f)
    finally:
        dist.destroy_process_group()


def test_sharded_manager_on_two_processes_equals_the_oracle_backed_unsharded_manager(tmp_path):
    """``hip:///path?devices=2`` end to end on the GPU (N1): every rank's protocol answers equal the oracle engine's."""
    import json

    import torch.multiprocessing as mp

    from iscc_search_amd.index import HipIndexManager
    from oracle_engine import OracleEngine
    from test_sharded_manager_gloo import scenario

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_manager_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    uri = f"hip://{tmp_path}/single/store"
    want = json.loads(json.dumps(scenario(HipIndexManager(uri, engine=OracleEngine()), reopen=lambda: HipIndexManager(uri, engine=OracleEngine()))))
    for rank in range(2):
        with open(tmp_path / f"rank{rank}.json") as f:
            got = json.load(f)
        assert len(got) == len(want)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g == w, f"rank {rank}, answer {i}"


def test_asynchronous_step_marks_overflow_and_the_step_is_repeated_exactly(hip_engine):
    """
    The multi-GPU step runs without host round-trips (isccsearch_search_device_async -> collective -> merge_device_after).
    A candidate list that overflows cannot take the exact fallback there: it is marked (COUNT_OVERFLOW), the marker survives
    the merge, and ShardedTable repeats the step through the synchronous path.  120 000 identical codes overflow every list.
    """
    from iscc_search_amd import _lib
    from iscc_search_amd.sharded import HipShardOps, ShardedTable, block_bytes
    from oracle import oracle_topk

    import torch

    n, k = 120_000, 10
    words = np.full((n, 1), 0x2222222222222222, dtype=np.uint64)
    words[::5] ^= np.uint64(3)
    keys = np.arange(n, dtype=np.uint64)[::-1].copy() + np.uint64(9)
    q = np.array([[0x2222222222222222], [0x2222222222222221], [0x0F0F0F0F0F0F0F0F]], dtype=np.uint64)
    t = hip_engine.open_table(0, 1, 8)
    try:
        t.add(keys, words)
        # the raw asynchronous call: overflowed queries carry the marker, the others are complete
        rec_bytes, blk = block_bytes(len(q), k)
        buf = torch.empty(blk, dtype=torch.uint8, device="cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        t.search_device(q, None, k, buf.data_ptr(), buf.data_ptr() + rec_bytes, consumer_stream=stream)
        merged = hip_engine.merge_device(1, len(q), k, 1, buf.data_ptr(), buf.data_ptr() + rec_bytes, blk, blk, after_stream=stream)
        assert (merged[3] == _lib.COUNT_OVERFLOW).any()
        # the product path repeats the step synchronously and ends up bit-identical to the oracle
        before = hip_engine.stats()["fallback_queries"]
        got = ShardedTable(HipShardOps(t, "cuda:0")).search(q, None, k)
        assert hip_engine.stats()["fallback_queries"] > before
        exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=8)
        for g, e, name in zip(got, exp, ("keys", "hamming", "prefix_bits", "count")):
            np.testing.assert_array_equal(g, e, err_msg=name)
    finally:
        t.drop()
