"""
The driver's contract for ``bench.py``: ONE JSON line on stdout with the agreed keys, `roofline` and `cpu_baseline`
objects included.  Runs the real script on a small index (2 M rows) so that it finishes in seconds.
"""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    proc = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "2000000", "--steps", "3", "--warmup", "1", "--cpu-queries", "64"],
        capture_output=True, text=True, timeout=600, cwd=ROOT,
    )
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1
    assert out["unit"] == "queries/s" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert abs(out["value"] - 1024 / (out["ms_per_step"] / 1e3)) / out["value"] < 1e-6
    assert out["config"]["scan"] == "FP4 MFMA"
    assert "workload" in out["config"] and "model" not in out["config"]
    roof = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_measured_in_run"):
        assert key in roof, key
    # 1 024-query steps run on the matrix cores by default; whatever the bound, the fraction is a fraction
    assert roof["bound"] == "mfma" and roof["unit"].startswith("TOP/s") and abs(roof["peak"] - 10066.3) < 1.0
    assert roof["traffic_measured_in_run"] is False
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0 < roof["frac"] <= 1.0
    assert roof["launches"] >= 3 and roof["avg_launch_ms"] > 0
    # the two other regimes of the same step, measured outside the timed region
    valu, stream = out["roofline_valu"], out["roofline_streaming"]
    assert valu["bound"] == "valu" and 0 < valu["frac"] <= 1.0
    assert stream["bound"] == "hbm" and stream["unit"] == "GB/s" and stream["peak"] == 8000.0 and 0 < stream["frac"] <= 1.0
    cpu = out["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "py_memory_style", "usearch"):
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0
    assert cpu["py_memory_style"]["value"] > 0 and cpu["usearch"].startswith("unavailable")
    # the oracle's answers of the CPU leg were compared bit for bit with the GPU's
    assert out["parity_checked_queries"] == 64
    # round 3: the unsettled figure beside `value`, the memory picture of the timed kernel, no field that
    # reads as bytes the matrix-core kernel never moved, and BASELINE configs 3 and 5 in the same run
    assert out["value_unsettled"] > 0 and "settle_steps" in out
    assert "whole_step_algorithmic_GBs" not in out and "algorithmic_bytes_per_launch" not in roof and roof["section8d_bytes_equivalent"] > 0
    assert roof["traffic"] is None or 0 < roof["hbm_traffic_frac"] < 1.0
    assert "algorithmic_bytes_per_launch" in stream
    for name, bits in (("config3", 256), ("config5_shape", 128)):
        oc = out["other_configs"][name]
        assert oc["queries_per_s"] > 0 and oc["parity_checked_queries"] == 4 and str(bits) in oc["workload"]
        assert oc["roofline"] is None or 0 < oc["roofline"]["frac"] <= 1.0
    # the steps of the timed region start under the previous step's k-th distance (a hint carried from step to step, verified):
    # the line says so and times the same steps without it
    hint = out["threshold_hint"]
    assert hint["steps_started_under_a_hint"] + hint["hints_that_did_not_hold"] >= 1
    assert hint["value_without_hints"] > 0 and hint["ms_per_step_without_hints"] > 0
    # round 4: the steps rotate through DISTINCT query batches, each gated once and compared after every
    # measured leg, two of them against the oracle; config 5 end to end and config 1 through the protocol ride in other_configs
    assert out["config"]["distinct_query_batches"] == 8 and out["batches_gated"] == 8 and hint["distinct_batches_in_rotation"] == 8
    assert len(out["batches_compared_after_measurement"]) >= 3
    assert cpu["parity_checked_queries_second_batch"] == 64
    c5 = out["other_configs"]["config5_end_to_end"]
    assert c5["scores_equal_host_scoring"] is True and c5["ms_per_request"] > 0 and c5["ms_inside_the_library"] <= c5["ms_per_request"]
    c1 = out["other_configs"]["config1_protocol"]
    assert c1["add_assets_per_s"] > 0 and c1["search_assets_per_s"] > 0


@pytest.mark.parametrize("one_queue", ["1", "0"])
def test_a_one_rank_collective_run_reports_what_the_communicator_saw(one_queue):
    """... with the step's search, all-gather and merge on the library's stream (default) and on torch's own stream, ordered by events:
    bench.py gates every batch (planted neighbour first, sorted, full) and compares every batch's last answer with its gate answer."""
    proc = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "2000000", "--steps", "3", "--warmup", "1", "--force-collective",
         "--no-cpu-baseline", "--no-extra-legs"],
        capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, ISCC_HIP_SHARD_ONE_QUEUE=one_queue),
    )
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = json.loads([l for l in proc.stdout.splitlines() if l.strip()][-1])
    assert out["world_size_seen"] == 1 and out["collective_backend"] == "nccl" and out["rccl_version"]
    assert out["batches_gated"] == 8 and len(out["batches_compared_after_measurement"]) >= 3      # three measured steps (and what the other legs rotated through)
