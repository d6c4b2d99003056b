"""
N1: ``hip:///?devices=2`` -- ``HipIndexManager`` over a ``ShardedEngine``, two processes (gloo, CPU).

Every rank runs the same protocol calls; rows are routed by key hash, searches are local top-k + ONE all-gather + merge
(``sharded.ShardedTable``), owner lookups (contains / get / size / remove count) one small all-reduce.  Each rank's
answers must equal those of an unsharded manager over the same oracle engine, call for call: add (created / updated),
update that replaces units and chunks, update that drops the INSTANCE unit, search by code / units / iscc_id /
simprints (approximate mode with device document frequencies, and exact mode), snapshot + reload.
"""

import json
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import flip_bits, make_asset, sp
from iscc_search_amd import codec
from iscc_search_amd.index import HipIndexManager
from iscc_search_amd.schema import IsccIndex, IsccQuery
from oracle_engine import OracleEngine


def scenario(manager, reopen=None):
    """The same sequence of protocol calls for the sharded and the unsharded manager; returns everything they answered."""
    rng = np.random.default_rng(11)
    out = []
    dump = lambda r: json.dumps(r.model_dump(mode="json"), sort_keys=True)       # noqa: E731
    s = [rng.integers(0, 256, size=16, dtype=np.uint8).tobytes() for _ in range(4)]
    assets = [make_asset(rng, i, metadata={"source": f"https://example.com/{i}"}) for i in range(60)]
    for i in range(0, 40, 3):       # chunk fingerprints on some assets: shared simprints give document frequencies > 1
        chunks = [sp(s[i % 4], 0, 10), sp(flip_bits(s[(i + 1) % 4], i % 5), 10, 20), sp(s[0], 30, 5)]
        assets[i] = assets[i].model_copy(update={"simprints": {"CONTENT_TEXT_V0": chunks}})
    manager.create_index(IsccIndex(name="main"))
    out.append([r.status.value for r in manager.add_assets("main", assets[:35])])
    out.append([r.status.value for r in manager.add_assets("main", assets[30:])])          # 30..34 again: idempotent re-add
    out.append(manager.get_index("main").assets)
    # updates: new units + new chunks for asset 3; asset 7 loses its INSTANCE unit
    newer = make_asset(rng, 3, simprints={"CONTENT_TEXT_V0": [sp(s[2], 0, 7), sp(s[3], 7, 7)]})
    inst = [u for u in assets[7].units if codec.Iscc(u).unit_type.startswith("INSTANCE_")][0]
    no_inst = assets[7].model_copy(update={"units": [u for u in assets[7].units if u != inst], "iscc_code": None})
    out.append([r.status.value for r in manager.add_assets("main", [newer, no_inst])])
    queries = [
        IsccQuery(iscc_code=assets[10].iscc_code),
        IsccQuery(iscc_code=assets[3].iscc_code),                      # the replaced version: no longer a 1.0 match
        IsccQuery(iscc_code=newer.iscc_code),
        IsccQuery(units=[inst]),                                       # dropped INSTANCE row
        IsccQuery(units=assets[20].units[:2]),
        IsccQuery(iscc_id=assets[12].iscc_id),
        IsccQuery(simprints={"CONTENT_TEXT_V0": [codec.encode_base64(s[0]), codec.encode_base64(s[1])]}),
        IsccQuery(simprints={"CONTENT_TEXT_V0": [codec.encode_base64(flip_bits(s[2], 2))]}),
    ]
    for q in queries:
        out.append(dump(manager.search_assets("main", q, limit=10)))
    idx = manager._indexes["main"]
    out.append(dump(idx.search_assets(queries[6], limit=10, exact=True)))                 # hard-boundary collision search
    table = idx._sp_tables["CONTENT_TEXT_V0"]
    out.append([int(f) for f in table.doc_freq([s[0], s[1], s[2], bytes(16)])])
    # both branches of the sharded count: above, every collision fits the limit and the shards' counts add;
    # with a limit of 3 the list is cut ACROSS the shards in key order and only the merged list knows which assets survive
    out.append([int(f) for f in table.doc_freq([s[0], s[1], s[2], bytes(16)], dup_limit=3)])
    out.append([int(f) for f in table.doc_freq([s[0], s[1]], dup_limit=1)])
    out.append(table.size)
    out.append(manager.get_asset("main", assets[3].iscc_id).iscc_code)
    if reopen is not None:
        manager.flush()
        manager.close()
        manager = reopen()
        out.append(sorted((i.name, i.assets) for i in manager.list_indexes()))
        for q in queries:
            out.append(dump(manager.search_assets("main", q, limit=10)))
        out.append([r.status.value for r in manager.add_assets("main", [make_asset(rng, 3)])])      # keep writing after a restore
        out.append(dump(manager.search_assets("main", IsccQuery(iscc_code=newer.iscc_code), limit=5)))
    manager.close()
    return out


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from iscc_search_amd.sharded_engine import ShardedEngine
        from test_sharded_gloo import OracleShardOps

        uri = f"hip://{out_dir}/store?devices={world}"
        make = lambda: HipIndexManager(uri, engine=ShardedEngine(OracleEngine(), ops_factory=OracleShardOps))      # noqa: E731
        m = make()
        assert m.devices == world
        out = scenario(m, reopen=make)
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
            json.dump(out, f)
        # the rows really are spread over the ranks
        m2 = make()
        local = m2._index("main")._unit_tables["DATA_NONE_V0"]._table.local.size
        with open(os.path.join(out_dir, f"local{rank}.json"), "w") as f:
            json.dump(local, f)
        m2.close()
    finally:
        dist.destroy_process_group()


def test_sharded_manager_answers_like_the_unsharded_one(tmp_path):
    world = 2
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    single_dir = tmp_path / "single"
    uri = f"hip://{single_dir}/store"
    want = scenario(HipIndexManager(uri, engine=OracleEngine()), reopen=lambda: HipIndexManager(uri, engine=OracleEngine()))
    locals_ = []
    for rank in range(world):
        with open(tmp_path / f"rank{rank}.json") as f:
            got = json.load(f)
        assert len(got) == len(want)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g == (json.loads(json.dumps(w))), f"rank {rank}, answer {i}"
        with open(tmp_path / f"local{rank}.json") as f:
            locals_.append(json.load(f))
    assert sum(locals_) == 60 and all(0 < n < 60 for n in locals_), locals_


def test_a_sharded_index_without_gpus_fails_loudly():
    """
    ONE process asking for ``devices=4`` becomes the leader of a front (``shard_front.LeaderEngine``) and starts the other ranks
    itself; on a box without GPUs the RCCL group cannot be formed: the call raises, the workers it had started are stopped, and
    nothing pretends to be a sharded index (no CPU fallback).
    """
    from iscc_search_amd import shard_front

    m = HipIndexManager("hip:///?devices=4")
    with pytest.raises((ValueError, RuntimeError)):
        m._get_engine()
    assert m._engine is None and not shard_front.leads_this_process() and not dist.is_initialized()
    with pytest.raises(ValueError, match="devices must be >= 1"):
        HipIndexManager("hip:///?devices=0")
