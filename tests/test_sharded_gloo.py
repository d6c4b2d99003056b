"""
Multi-process (world_size 2, gloo, CPU) cover of the row-sharded search path:
``ShardedTable`` = local exact top-k per shard -> ONE all-gather of {records | counts} blocks ->
k-way merge, identical on every rank and identical to the unsharded oracle answer.
The device kernels are replaced by the oracle-backed ops object; the exchange layout, the shard
arithmetic and the collective call are the product code (``iscc_search_amd/sharded.py``).
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from iscc_search_amd.sharded import RECORD_BYTES, ShardedTable, block_bytes, shard_of_key, shard_range
from oracle import oracle_topk
from oracle_engine import RECORD_DTYPE, OracleTable


class OracleShardOps:
    """CPU stand-in for HipShardOps with the same block layout."""

    def __init__(self, table):
        self.table = table
        self.key_words = table.key_words

    supports_hint = False      # HintedOracleShardOps below: the single pass started under a hint lists the rows WITHIN it

    def local_search(self, q_words, q_nbytes, k, max_hamming=None, synchronous=False, hint=None):
        if hint is not None and max_hamming is None:
            type(self).hinted_calls = getattr(type(self), "hinted_calls", 0) + 1
            max_hamming = hint
        rec, cnt = self.table.search_records(q_words, q_nbytes, k, max_hamming)
        nq = q_words.shape[0]
        rec_bytes, blk = block_bytes(nq, k)
        buf = np.zeros(blk, dtype=np.uint8)
        buf[:rec_bytes] = rec.reshape(-1).view(np.uint8)
        buf[rec_bytes : rec_bytes + nq * 4] = cnt.astype("<i4").view(np.uint8)
        return torch.from_numpy(buf)

    def local_doc_freq(self, q_words, q_nbytes, dup_limit):
        return self.table.doc_freq_counted(q_words, q_nbytes, dup_limit)

    def merge(self, gathered, n_lists, nq, k):
        rec_bytes, blk = block_bytes(nq, k)
        raw = gathered.numpy()
        entries = [[] for _ in range(nq)]
        for l in range(n_lists):
            block = raw[l * blk : (l + 1) * blk]
            rec = block[:rec_bytes].view(RECORD_DTYPE).reshape(nq, k)
            cnt = block[rec_bytes : rec_bytes + nq * 4].view("<i4")
            for q in range(nq):
                for i in range(int(cnt[q])):
                    r = rec[q, i]
                    entries[q].append((int(r["dist_rank"]), int(r["key_hi"]), int(r["key_lo"]), int(r["hamming"]), int(r["prefix_bits"])))
        shape = (nq, k, 2) if self.key_words == 2 else (nq, k)
        keys = np.zeros(shape, np.uint64)
        ham = np.zeros((nq, k), np.uint32)
        pb = np.zeros((nq, k), np.uint16)
        out_cnt = np.zeros(nq, np.uint32)
        for q in range(nq):
            top = sorted(entries[q])[:k]
            out_cnt[q] = len(top)
            for i, (_, hi, lo, h, p) in enumerate(top):
                if self.key_words == 2:
                    keys[q, i] = (hi, lo)
                else:
                    keys[q, i] = lo
                ham[q, i], pb[q, i] = h, p
        return keys, ham, pb, out_cnt


def _dataset(metric):
    rng = np.random.default_rng(123)
    n = 4000
    if metric == 1:
        lens = rng.choice([8, 16, 32], size=n).astype(np.uint8)
        words = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
        for j in range(4):
            words[lens.astype(np.int64) <= 8 * j, j] = 0
        q = words[[3, 99, 2500]].copy()
        q[:, 0] ^= np.uint64(5)
        qlens = lens[[3, 99, 2500]].copy()
    else:
        lens = None
        words = rng.integers(0, 8, size=(n, 1), dtype=np.uint64)        # tiny code space: massive ties across shards
        q = np.array([[1], [6], [3]], dtype=np.uint64)
        qlens = None
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(17)
    return keys, words, lens, q, qlens


def _worker(rank, world, port, metric, routing, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        keys, words, lens, q, qlens = _dataset(metric)
        n = len(keys)
        if routing == "range":
            lo, hi = shard_range(n, rank, world)
            mine = np.arange(lo, hi)
        else:
            mine = np.array([i for i in range(n) if shard_of_key(int(keys[i]), world) == rank])
        table = OracleTable(metric, 1, 32 if metric == 1 else 8)
        table.add(keys[mine], words[mine], None if lens is None else lens[mine])
        sharded = ShardedTable(OracleShardOps(table))
        assert (sharded.rank, sharded.world_size) == (rank, world)
        got = sharded.search(q, qlens, 12)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), *got)
    finally:
        dist.destroy_process_group()


def _within_dataset():
    """Simprint-style table: 128-bit keys (asset, chunk), few distinct 64-bit codes, assets spread over both shards."""
    rng = np.random.default_rng(77)
    n = 3000
    words = rng.integers(0, 6, size=(n, 1), dtype=np.uint64) * np.uint64(0x0101010101010101)
    keys = np.stack([rng.integers(1, 25, size=n).astype(np.uint64), rng.permutation(n).astype(np.uint64)], axis=1)
    q = np.array([[0], [0x0303030303030303], [0x0303030303030302], [0x7777777777777777]], dtype=np.uint64)
    return keys, words, q


def _within_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        keys, words, q = _within_dataset()
        lo, hi = shard_range(len(keys), rank, world)
        table = OracleTable(0, 2, 8)
        table.add(keys[lo:hi], words[lo:hi])
        sharded = ShardedTable(OracleShardOps(table))
        out = {}
        for r, k in ((0, 1000), (0, 7), (1, 40), (9, 25)):
            for i, a in enumerate(sharded.search_within(q, None, k, r)):
                out[f"w_{r}_{k}_{i}"] = a
        out["freq_1000"] = sharded.doc_freq(q, None, 1000)
        out["freq_5"] = sharded.doc_freq(q, None, 5)
        np.savez(os.path.join(out_dir, f"w{rank}.npz"), **out)
    finally:
        dist.destroy_process_group()


def test_two_rank_range_limited_search_and_doc_freq_equal_the_unsharded_table(tmp_path):
    world = 2
    mp.spawn(_within_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    keys, words, q = _within_dataset()
    whole = OracleTable(0, 2, 8)
    whole.add(keys, words)
    for rank in range(world):
        with np.load(os.path.join(tmp_path, f"w{rank}.npz")) as z:
            for r, k in ((0, 1000), (0, 7), (1, 40), (9, 25)):
                for i, e in enumerate(whole.search_within(q, None, k, r)):
                    np.testing.assert_array_equal(z[f"w_{r}_{k}_{i}"], e, err_msg=f"rank {rank} r={r} k={k} field {i}")
            np.testing.assert_array_equal(z["freq_1000"], whole.doc_freq(q, None, 1000))
            np.testing.assert_array_equal(z["freq_5"], whole.doc_freq(q, None, 5))
            assert z["freq_1000"][0] > 1 and z["freq_1000"][3] == 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("metric,routing", [(0, "range"), (1, "range"), (1, "hash")])
def test_two_rank_sharded_search_equals_unsharded_oracle(tmp_path, metric, routing):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), metric, routing, str(tmp_path)), nprocs=world, join=True)
    keys, words, lens, q, qlens = _dataset(metric)
    exp = oracle_topk(metric, keys, words, lens, q, qlens, 12, fixed_nbytes=8 if metric == 0 else 0)
    results = []
    for r in range(world):
        with np.load(os.path.join(tmp_path, f"r{r}.npz")) as z:
            results.append([z[f"arr_{i}"] for i in range(4)])
    for got in results:                      # every rank holds the same, exact, global answer
        for g, e in zip(got, exp):
            np.testing.assert_array_equal(g, e)


def test_shard_arithmetic():
    for n in (0, 1, 7, 100, 10**8 + 3):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert block_bytes(3, 5) == (3 * 5 * RECORD_BYTES, 3 * 5 * RECORD_BYTES + 16)
    assert {shard_of_key(k, 8) for k in range(1000)} == set(range(8))


def test_one_shard_takes_the_direct_entry_point_when_the_ops_offer_one():
    """
    With a single rank and no forced collective ``ShardedTable`` hands the whole search to ``ops.search_single`` (the product's
    host entry point: no result block, no merge launch); ops without it -- the oracle stand-in above -- keep the block + merge
    path.  Both must give the same answer.
    """
    keys, words, lens, q, qlens = _dataset(0)
    table = OracleTable(0, 1, 8)
    table.add(keys, words, None)

    class DirectOps(OracleShardOps):
        calls = 0

        def search_single(self, q_words, q_nbytes, k, max_hamming=None):
            DirectOps.calls += 1
            block = OracleShardOps.local_search(self, q_words, q_nbytes, k, max_hamming)
            return OracleShardOps.merge(self, block, 1, q_words.shape[0], k)

        def local_search(self, *a, **kw):       # the shortcut must not build a block
            raise AssertionError("local_search called although search_single is offered")

    direct = ShardedTable(DirectOps(table))
    assert direct.world_size == 1
    got = direct.search(q, qlens, 12)
    got_within = direct.search_within(q, qlens, 12, 1)
    assert DirectOps.calls == 2
    plain = ShardedTable(OracleShardOps(table))
    for a, b in zip(got, plain.search(q, qlens, 12)):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(got_within, plain.search_within(q, qlens, 12, 1)):
        np.testing.assert_array_equal(a, b)


class HintedOracleShardOps(OracleShardOps):
    supports_hint = True


def _hint_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(4711)
        n, nq, k = 6000, 5, 8
        words = rng.integers(0, 2**64, size=(n, 1), dtype=np.uint64)
        dup = np.uint64(0x0F0F0F0F0F0F0F0F)
        words[:40, 0] = dup ^ (np.uint64(1) << rng.integers(0, 64, size=40).astype(np.uint64))     # 40 rows one bit off `dup`
        keys = rng.permutation(n).astype(np.uint64) + np.uint64(3)
        lo, hi = shard_range(n, rank, world)
        table = OracleTable(0, 1, 8)
        table.add(keys[lo:hi], words[lo:hi])
        HintedOracleShardOps.hinted_calls = 0
        sharded = ShardedTable(HintedOracleShardOps(table))
        assert sharded.use_hints
        random_q = rng.integers(0, 2**64, size=(4 * nq, 1), dtype=np.uint64)
        near_q = np.full((nq, 1), dup, dtype=np.uint64)
        log = []

        def ask(q):
            before = HintedOracleShardOps.hinted_calls
            got = sharded.search(q, None, k)
            exp = oracle_topk(0, keys, words, None, q, None, k, fixed_nbytes=8)
            for g, e in zip(got, exp):
                np.testing.assert_array_equal(g, e)
            log.append(HintedOracleShardOps.hinted_calls - before)

        ask(random_q[:nq])            # 0: nothing to go by; seeds the GLOBAL k-th distance + 2
        ask(random_q[nq : 2 * nq])    # 1: every shard starts under it; the merged lists hold k rows: stands
        ask(near_q)                   # 1: far inside the hint (which decays by one bit)
        ask(random_q[2 * nq : 3 * nq])   # 1
        sharded._hints[(int(nq).bit_length(), k, None)][0] = 3    # a hint that is too tight for random queries ...
        ask(random_q[3 * nq :])       # 1: ... the merged lists come up short: the step is repeated without it (and re-seeds)
        ask(random_q[:nq])            # 1: one miss does not back off
        assert log == [0, 1, 1, 1, 1, 1], log
        ask(near_q[:2])               # another batch-size class: its own hint
        assert log[-1] == 0
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array(log))
    finally:
        dist.destroy_process_group()


def test_shards_start_under_the_global_kth_distance_of_the_previous_step(tmp_path):
    """Hit, decay, miss and re-seed of the sharded hint, two ranks over gloo, every answer against the unsharded oracle."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mp.spawn(_hint_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0.npy", "ok1.npy"]
