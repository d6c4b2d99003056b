"""
N1: ``HipIndexManager("hip:///path?devices=2")`` constructed in ONE process -- as the reference's server
(``iscc_search/server/__init__.py:75-135``) and CLI (``iscc_search/cli/common.py:41-97``) construct their index -- must work:
that process becomes the leader, starts the shard worker itself and broadcasts every protocol call to it
(``iscc_search_amd/shard_front.py``).  Only the leader makes calls; its answers must equal the unsharded manager's.

CPU tier: gloo, oracle-backed engines on both ranks.  The same scenario with the product engine, two processes on the one
GPU of the box: ``test_gpu_shard_leader.py``.
"""

import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from run_leader_scenario import protocol_scenario  # noqa: E402


def run_leader(uri, factory, out_path, *extra, timeout=600, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    env["PYTHONPATH"] = os.pathsep.join([HERE, os.path.dirname(HERE), env.get("PYTHONPATH", "")])
    proc = subprocess.run([sys.executable, os.path.join(HERE, "run_leader_scenario.py"), uri, factory, str(out_path), *extra],
                          capture_output=True, text=True, timeout=timeout, env=env)
    assert proc.returncode == 0, proc.stderr[-4000:]
    with open(out_path) as f:
        return json.load(f)


def test_one_process_leads_a_two_rank_index(tmp_path):
    from iscc_search_amd.index import HipIndexManager
    from oracle_engine import OracleEngine

    got = run_leader(f"hip://{tmp_path}/sharded?devices=2&backend=gloo", "shard_factories:oracle", tmp_path / "leader.json")
    uri = f"hip://{tmp_path}/single"
    want = protocol_scenario(lambda: HipIndexManager(uri, engine=OracleEngine()))
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == json.loads(json.dumps(w)), f"answer {i}"
    # both ranks hold rows: the leader's snapshot has one shard directory per rank
    units = os.path.join(tmp_path, "sharded", "main", "units")
    some_type = sorted(os.listdir(units))[0]
    assert sorted(os.listdir(os.path.join(units, some_type))) == ["shard-0-of-2", "shard-1-of-2"]


def test_a_dead_worker_breaks_the_front_instead_of_hanging_it(tmp_path):
    out = run_leader(f"hip://{tmp_path}/sharded?devices=2&backend=gloo", "shard_factories:oracle", tmp_path / "kill.json", "kill", timeout=300)
    assert out["first"].startswith("the sharded index is down"), out
    assert out["again"].startswith("the sharded index is down"), out
    assert out["seconds"] < 120, out


def test_an_idle_front_outlives_the_collective_deadline(tmp_path):
    """workers wait for the next request inside a collective with a deadline; the leader's heartbeat keeps it from expiring."""
    out = run_leader(f"hip://{tmp_path}/sharded?devices=2&backend=gloo", "shard_factories:oracle", tmp_path / "idle.json", "idle",
                     timeout=300, env_extra={"ISCC_HIP_SHARD_TIMEOUT_S": "4"})
    assert out["idle_seconds"] > 8 and out["top"] == out["want"] and out["score"] == 1.0, out


def test_eight_caller_threads_get_the_single_threaded_answers(tmp_path):
    """the leader front combines concurrent callers; every thread's answer equals the sequential one."""
    out = run_leader(f"hip://{tmp_path}/sharded?devices=2&backend=gloo", "shard_factories:oracle", tmp_path / "threads.json", "threads", timeout=300)
    assert out["errors"] == [] and out["equal"] is True, out


def test_outcomes_alike_reach_the_caller_outcomes_that_differ_end_the_front(tmp_path):
    """errors are classified by what the RANKS report, not by exception type alone."""
    out = run_leader(f"hip://{tmp_path}/sharded?devices=2&backend=gloo", "shard_factories:oracle", tmp_path / "outcomes.json", "outcomes", timeout=300)
    assert out["alike"].startswith("ValueError") and out["still_up"] is True, out
    assert out["invalid"] == ["ValueError"] * 3 and out["up_after_invalid"] is True, out
    assert out["differ"].startswith("the sharded index is down") and "disagree" in out["differ"], out
    assert out["after"].startswith("the sharded index is down"), out
