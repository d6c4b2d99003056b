"""
N1 on the GPU box: ONE process opens ``hip:///path?devices=2`` with the PRODUCT engine; it becomes the leader, starts one
shard worker (a fresh interpreter) and both ranks share the box's one GPU (``same_gpu=1``: the exchange then runs over gloo,
staged through the host by ``ShardedTable`` -- ranks that share a GPU cannot form an RCCL communicator; everything else is
the multi-GPU code path).  Only the leader makes protocol calls; every answer must equal the unsharded oracle-backed manager's.
"""

import json

import pytest

from test_shard_leader import protocol_scenario, run_leader

pytestmark = pytest.mark.gpu


def test_one_process_leads_two_ranks_on_the_product_engine(tmp_path):
    from iscc_search_amd.index import HipIndexManager
    from oracle_engine import OracleEngine

    got = run_leader(f"hip://{tmp_path}/sharded?devices=2&backend=gloo&same_gpu=1", "-", tmp_path / "leader.json")
    uri = f"hip://{tmp_path}/single"
    want = protocol_scenario(lambda: HipIndexManager(uri, engine=OracleEngine()))
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == json.loads(json.dumps(w)), f"answer {i}"


def test_a_dead_worker_breaks_the_front_on_the_product_engine(tmp_path):
    out = run_leader(f"hip://{tmp_path}/sharded?devices=2&backend=gloo&same_gpu=1", "-", tmp_path / "kill.json", "kill", timeout=300)
    assert out["first"].startswith("the sharded index is down") and out["again"].startswith("the sharded index is down"), out
