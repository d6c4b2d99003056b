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


def _stored_codes(n):
    """The codes of rows 0 .. n-1 of the synthetic table (SURVEY 8d generator), as queries: nearest neighbour at distance 0."""
    from oracle import oracle_splitmix64_fill

    return oracle_splitmix64_fill(n, SEED, stride=4).reshape(n, 1)


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
        # SMALL batches (the protocol's per-unit searches): from the second search of a shape on, every shard takes ONE range-limited
        # pass under the global k-th distance of the search before (isccsearch_search_device_async with a hint); a batch of stored
        # codes leaves a hint no random batch can hold: noticed on the merged lists, repeated without it
        near = _stored_codes(3)
        small = [q[:1], q[1:2], q[2:3], q[:5], q[5:10], near, q[10:13], q[13:16], q[16:17]]
        for j, qq in enumerate(small):
            for i, a in enumerate(sharded.search(qq, None, K)):
                out[f"m{j}_{i}"] = a
        # k = 1: three stored codes end at distance 0 and leave a hint of 2 bits, which three random queries cannot hold
        for j, qq in enumerate([near, q[10:13], q[13:16]]):
            for i, a in enumerate(sharded.search(qq, None, 1)):
                out[f"n{j}_{i}"] = a
        out["hints"] = np.array([sharded.hint_hits, sharded.hint_misses])
        # the per-unit searches of one request on two tables: ONE exchange for all of them (ShardedTable.search_many), twice (the
        # second time under the first one's hints)
        table2 = engine.open_table(0, 1, 8)
        table2.add_synthetic(8, hi - lo, SEED + 99, first_row=lo)
        sharded2 = ShardedTable(HipShardOps(table2, "cuda:0"))
        for rep in range(2):
            fused = ShardedTable.search_many([(sharded, q[17:18], None, K, None), (sharded2, q[18:19], None, K, None), (sharded2, q[19:22], None, 3, 16)])
            assert fused is not None
            for j, res in enumerate(fused):
                for i, a in enumerate(res):
                    out[f"f{rep}_{j}_{i}"] = a
        out["hints2"] = np.array([sharded.hint_hits + sharded2.hint_hits, sharded.hint_misses + sharded2.hint_misses])
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
    # the small batches, hinted from their second search on, and the fused per-unit searches
    words2 = oracle_splitmix64_fill(ROWS, SEED + 99, stride=4).reshape(ROWS, 1)
    small = [q[:1], q[1:2], q[2:3], q[:5], q[5:10], _stored_codes(3), q[10:13], q[13:16], q[16:17]]
    for rank in range(2):
        with np.load(os.path.join(tmp_path, f"r{rank}.npz")) as z:
            for j, qq in enumerate(small):
                want = oracle_topk(0, row_keys, words, None, qq, None, K)
                for i in range(4):
                    np.testing.assert_array_equal(z[f"m{j}_{i}"], want[i], err_msg=f"rank {rank} small batch {j} field {i}")
            for j, qq in enumerate([_stored_codes(3), q[10:13], q[13:16]]):
                want = oracle_topk(0, row_keys, words, None, qq, None, 1)
                for i in range(4):
                    np.testing.assert_array_equal(z[f"n{j}_{i}"], want[i], err_msg=f"rank {rank} k = 1 batch {j} field {i}")
            assert z["hints"][0] >= 3 and z["hints"][1] >= 1, z["hints"]          # hints held, and the one left by the stored codes did not
            fused_want = [oracle_topk(0, row_keys, words, None, q[17:18], None, K), oracle_topk(0, row_keys, words2, None, q[18:19], None, K)]
            within = (np.zeros((3, 3), np.uint64), np.zeros((3, 3), np.uint32), np.zeros((3, 3), np.uint16), np.zeros(3, np.uint32))
            for i in range(3):
                kk, h, p = np_within(words2, 8, row_keys, q[19 + i], 8, 3, 16)
                c = len(h)
                within[0][i, :c], within[1][i, :c], within[2][i, :c], within[3][i] = kk, h, p, c
            fused_want.append(within)
            for rep in range(2):
                for j, want in enumerate(fused_want):
                    for i in range(4):
                        np.testing.assert_array_equal(z[f"f{rep}_{j}_{i}"], want[i], err_msg=f"rank {rank} fused pass {rep} item {j} field {i}")
            assert z["hints2"][0] > z["hints"][0]                                   # the second fused request started under the first one's hints


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
