#!/usr/bin/env python3
"""
bench.py -- queries/sec and achieved HBM GB/s of the brute-force 64-bit Hamming k=10 search.

Workload (BASELINE.json `metric`): 100 M synthetic 64-bit codes resident in HBM, k = 10, batches of
1 024 queries streamed as passes of T_q (=8) queries.  A "step" is one 1 024-query search through
the C-ABI (threshold bootstrap + sample scan + streaming scan + select [+ all-gather + merge]).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`;
   one rank per GPU over RCCL; STRONG scaling: the 100 M rows are row-range sharded over the ranks,
   every rank answers every query on its shard, ONE all-gather exchanges the per-shard top-k.)

Prints ONE JSON line on rank 0.  `roofline.achieved` = algorithmic bytes of the streaming-scan
launches (rows x 8 x words x query groups) / their summed device time, measured
with HIP events on the library's own stream inside the timed region.  `cpu_baseline` = the oracle
(CPU restatement, OpenMP) timed on the host cores over the same 100 M rows for a bounded number of
queries (rank 0, N = 1 only).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED_CODES = 0x1511CC00
SEED_Q = 0x1511CC02
SEED_P = 0x1511CC03
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6.29 TB/s is the measured copy ceiling
VALU_PEAK_TLANEOPS = 39.3  # 1024 SIMDs x 16 lanes x 2.4 GHz (profiles/r01_micro_valu.txt: 4 cycles per wave64 integer op)
MASK64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & MASK64
    return x ^ (x >> 31)


def make_queries(nq, rows, words):
    """random queries; every 4th is a stored code with f in {0,1,3,7} low bits flipped."""
    q = np.zeros((nq, words), dtype=np.uint64)
    planted = {}
    for j in range(nq):
        if j % 4 == 0:
            r = splitmix64(SEED_P + j) % rows
            f = (0, 1, 3, 7)[(j // 4) % 4]
            for w in range(words):
                q[j, w] = splitmix64(SEED_CODES + 4 * r + w)
            q[j, words - 1] ^= np.uint64(f)
            planted[j] = (r, bin(f).count("1"))
        else:
            for w in range(words):
                q[j, w] = splitmix64(SEED_Q + 4 * j + w)
    return q, planted


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=100_000_000, help="total rows of the index (all ranks together)")
    ap.add_argument("--queries", type=int, default=1024, help="queries per step")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--nbytes", type=int, default=8, help="code length in bytes (8 = 64-bit)")
    ap.add_argument("--metric", choices=["hamming", "nphd"], default="hamming", help="table metric (nphd: every row --nbytes long, queries too)")
    ap.add_argument("--tq", type=int, default=8, help="queries per pass (8|16); 8 keeps a streaming scan HBM-bound")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=0, help="queries timed on the CPU (0 = auto, ~10-30 s)")
    ap.add_argument("--no-profile", action="store_true", help="do not time scan launches with HIP events")
    ap.add_argument("--no-streaming-check", action="store_true", help="skip the extra HBM-streaming measurement (stretch_mb=0) after the timed region")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rehearsal of the N > 1 code path on a one-GPU box: every rank uses cuda:0 and the process group runs over gloo "
                         "(ranks sharing a GPU cannot form an RCCL communicator); the number it prints is not a scaling result")
    ap.add_argument("--force-collective", action="store_true", help="initialise RCCL and run the all-gather even with one rank (rehearsal)")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (tuning experiments)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from iscc_search_amd import _lib
    from iscc_search_amd.engine import HipEngine
    from iscc_search_amd.sharded import HipShardOps, ShardedTable, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with python -m torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:   # only the one-rank rehearsal gets here without torchrun
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    words = (args.nbytes + 7) // 8
    engine = HipEngine(local_rank)
    engine.set_option("queries_per_pass", args.tq)
    for kv in args.opt:
        name, _, val = kv.partition("=")
        engine.set_option(name, int(val))
    stretch_mb = 128
    for kv in args.opt:
        if kv.startswith("stretch_mb="):
            stretch_mb = int(kv.split("=")[1])
    nphd = args.metric == "nphd"
    table = engine.open_table(_lib.METRIC_NPHD if nphd else _lib.METRIC_HAMMING, 1, args.nbytes)
    q_nbytes = np.full(args.queries, args.nbytes, dtype=np.uint8) if nphd else None
    lo, hi = shard_range(args.rows, rank, world)
    table.add_synthetic(args.nbytes, hi - lo, SEED_CODES, first_row=lo, key_base=0)
    sharded = ShardedTable(HipShardOps(table, device), always_gather=args.force_collective)

    q, planted = make_queries(args.queries, args.rows, words)

    def step():
        return sharded.search(q, q_nbytes, args.k)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # correctness gate outside the timed region: planted neighbours must come back first
    keys, ham, pbits, cnt = step()
    for j, (r, f) in planted.items():
        assert int(cnt[j]) == min(args.k, args.rows), (j, cnt[j])
        assert int(ham[j, 0]) <= f, f"planted neighbour of query {j} not found: {ham[j, :3]} vs {f}"
        if f == 0:
            assert int(keys[j, 0]) == r or int(ham[j, 0]) == 0
    assert np.all(np.diff(ham.astype(np.int64), axis=1)[:, : args.k - 1] >= 0), "results not sorted"

    for _ in range(args.warmup):
        step()
    fence()
    engine.stats(reset=True)
    if not args.no_profile:
        engine.set_option("profile", 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    engine.set_option("profile", 0)
    st = engine.stats(reset=True)

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # Outside the timed region, rank 0 of a one-GPU run: the same steps with cache blocking OFF, i.e. every query
    # group streams the table from HBM -- the measurement that evidences the HBM roofline of the scan kernel
    # (with blocking on, all but the first group of a launch read their stretch from the 256 MB Infinity Cache).
    streaming = None
    if world == 1 and not args.no_profile and not args.no_streaming_check:
        engine.set_option("stretch_mb", 0)
        step()
        torch.cuda.synchronize()
        engine.stats(reset=True)
        engine.set_option("profile", 1)
        t1 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        engine.set_option("profile", 0)
        s2 = engine.stats(reset=True)
        if s2["scan_ms"] > 0:
            a2 = (s2["scan_bytes"] / 1e9) / (s2["scan_ms"] / 1e3)
            streaming = {
                "what": "same workload with stretch_mb=0 (no cache blocking): every pass streams the rows from HBM",
                "bound": "hbm", "achieved": a2, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a2 / HBM_PEAK_GBS,
                "launches": s2["scan_launches"], "avg_launch_ms": s2["scan_ms"] / s2["scan_launches"],
                "algorithmic_bytes_per_launch": s2["scan_bytes"] / s2["scan_launches"],
                "queries_per_s": args.queries * 5 / el,
            }
            try:
                with open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_size_scan_tq8.json")) as f:
                    per_row = json.load(f)["corrected_bytes_per_row_pass_streaming"]
                if args.nbytes == 8 and args.tq == 8:
                    streaming["traffic"] = per_row * streaming["algorithmic_bytes_per_launch"] / 8.0
            except (OSError, KeyError, ValueError):
                pass

    # HBM traffic of the dominant kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE, corrected x2
    # for gfx950's half-count of 16 B/lane streams): bytes per pass for this workload, scaled to one launch.
    traffic = None
    traffic_src = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_size_scan_tq8.json")) as f:
            pmc = json.load(f)
        per_row = None
        if args.nbytes == 8 and args.tq == 8 and world == 1:
            if stretch_mb == 0 or args.queries <= args.tq:
                per_row = pmc["corrected_bytes_per_row_pass_streaming"]
            elif stretch_mb == 128 and args.queries == 1024:
                per_row = pmc["corrected_bytes_per_row_pass"]
        if per_row and st["scan_launches"]:
            # bytes the L2s fetched per (row, query group) of the collect scan, scaled to one launch of this run
            traffic = per_row * (st["scan_bytes"] / 8.0) / st["scan_launches"]
            traffic_src = ("profiles/r01_pmc_fetch_size_scan_tq8.json (separate rocprofv3 --pmc FETCH_SIZE pass, x2 gfx950 correction; "
                           "FETCH_SIZE counts L2 misses, Infinity-Cache hits included)")
    except (OSError, KeyError, ValueError):
        traffic = None

    total_queries = args.queries * args.steps
    qps = total_queries / elapsed
    achieved = (st["scan_bytes"] / 1e9) / (st["scan_ms"] / 1e3) if st["scan_ms"] > 0 else None
    out = {
        "metric": "queries/sec (+ achieved HBM GB/s in `roofline`), 64-bit Hamming k=10 over 100M codes, exact top-k",
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.rows} x {args.nbytes * 8}-bit codes, brute-force {'NPHD' if nphd else 'Hamming'} k={args.k}, "
                        f"{args.queries} queries/step in passes of T_q={args.tq}, rows sharded over {world} GPU(s)",
            "rows_total": args.rows,
            "rows_per_gpu": hi - lo,
            "code_bits": args.nbytes * 8,
            "k": args.k,
            "queries_per_step": args.queries,
            "queries_per_pass": args.tq,
            "parallelism": f"row-shard x{world}, one all-gather of per-shard top-k",
        },
        "roofline": {
            "bound": "hbm",
            # the collect pass: scan_adapt_kernel<T_q, 3> for whole 64-bit codes, scan_kernel<W, mask, T_q, 3> otherwise
            # (mode 3 = MODE_STRETCH; the threshold levels in front of it are mode 2 launches of the same code)
            "kernel": ("scan_adapt_kernel<%d,STRETCH>" % args.tq) if words == 1 and args.nbytes % 8 == 0 else "scan_kernel<W=%d,STRETCH>" % words,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "launches": st["scan_launches"],
            "avg_launch_ms": (st["scan_ms"] / st["scan_launches"]) if st["scan_launches"] else None,
            "algorithmic_bytes_per_launch": (st["scan_bytes"] / st["scan_launches"]) if st["scan_launches"] else None,
            "regime": ("cache-blocked: %d query groups per launch share stretches of <= %d MB, read from HBM once and from the "
                       "256 MB Infinity Cache afterwards, so the algorithmic rate may exceed what HBM alone delivers; the kernel "
                       "is then bound by integer VALU issue (see `valu`); `roofline_streaming` is the HBM-bound measurement"
                       % (st["scan_passes"] // max(1, st["scan_launches"]), stretch_mb)) if stretch_mb and args.queries > args.tq else "streaming",
            "measured_copy_ceiling_GBs": 6290.0,          # MI355X_MICROARCH.md: measured streaming copy
            "measured_read_ceiling_GBs": 7050.0,          # profiles/r01_micro_read.txt: pure nontemporal read, same device
        },
        # every row is read exactly once per pass: the threshold levels (sample_bytes) stream the first stretch of
        # the table, the collect scan (scan_bytes) the rest
        "whole_step_GBs": ((st["scan_bytes"] + st["sample_bytes"]) / 1e9) / elapsed,
        "fallback_queries": st["fallback_queries"],
    }
    if achieved:
        # 4.5 VALU lane-operations per (row, query, 64-bit word): the other roofline of this kernel
        lane_ops = achieved * 1e9 / 8.0 * args.tq * 4.5
        out["roofline"]["valu"] = {"achieved": lane_ops / 1e12, "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s",
                                   "frac": lane_ops / 1e12 / VALU_PEAK_TLANEOPS}
    if streaming:
        out["roofline_streaming"] = streaming

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, q, words)

    if rank == 0:
        print(json.dumps(out))
    table.drop()
    engine.close()
    if dist.is_initialized():
        dist.destroy_process_group()


def usable_cores(omp_threads):
    """CPUs this process may really use: affinity mask and cgroup quota, not the host's core count."""
    n = omp_threads
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(args, q, words):
    """The oracle (CPU restatement of the same exact search) on the host cores, bounded to ~10-30 s."""
    from oracle import oracle_num_threads, oracle_splitmix64_fill, oracle_topk

    rows = args.rows
    cols = [oracle_splitmix64_fill(rows, SEED_CODES, stride=4, lane=w) for w in range(words)]
    code_words = np.ascontiguousarray(np.stack(cols, axis=1))
    if args.nbytes % 8:
        code_words[:, -1] &= np.uint64((MASK64 << (8 * (8 - args.nbytes % 8))) & MASK64)
    keys = np.arange(rows, dtype=np.uint64)
    threads = usable_cores(oracle_num_threads())
    nq = args.cpu_queries
    if nq <= 0:
        # calibrate on 2 queries, then size the sample for ~15 s
        t0 = time.perf_counter()
        oracle_topk(0, keys, code_words, None, q[:8], None, args.k, fixed_nbytes=args.nbytes, threads=threads)
        per_q = (time.perf_counter() - t0) / 8
        nq = int(max(8, min(args.queries, 15.0 / max(per_q, 1e-6))))
    reps = 1
    if args.cpu_queries <= 0 and nq == args.queries:
        reps = int(max(1, min(16, round(12.0 / max(per_q * nq, 1e-6)))))    # ~12 s of CPU work
    t0 = time.perf_counter()
    for _ in range(reps):
        oracle_topk(0, keys, code_words, None, q[:nq], None, args.k, fixed_nbytes=args.nbytes, threads=threads)
    dt = time.perf_counter() - t0
    nq = nq * reps
    # one core, on a 10 M-row slice (scaled linearly to the full table; labelled as extrapolated)
    slice_rows = min(rows, 10_000_000)
    t1 = time.perf_counter()
    oracle_topk(0, keys[:slice_rows], code_words[:slice_rows], None, q[:4], None, args.k, fixed_nbytes=args.nbytes, threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "single_core": {"value": 4 / dt1 * slice_rows / rows, "unit": "queries/s", "cores": 1,
                        "sample": f"4 queries over a {slice_rows}-row slice, extrapolated linearly to {rows} rows"},
        "value": nq / dt,
        "unit": "queries/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{nq} queries ({reps} x the step's batch) over all {rows} rows ({dt:.1f} s of CPU work, {threads} OpenMP threads = "
                  f"the CPUs this process may use; rows split across threads, cache-blocked over all queries)",
        "GBs": nq * rows * 8 * words / dt / 1e9,
    }


if __name__ == "__main__":
    main()
