#!/usr/bin/env python3
"""
bench.py -- queries/sec of the exact brute-force 64-bit Hamming k=10 search, with the roofline that binds the timed kernel.

Workload (BASELINE.json `metric`): 100 M synthetic 64-bit codes resident in HBM, k = 10, batches of 1 024 queries.
A "step" is one 1 024-query search through the C-ABI (threshold bootstrap + ONE self-tightening scan pass + select
[+ all-gather + merge]; batches of <= 16 queries and k > 512 take threshold levels + a collect pass instead).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`;
   one rank per GPU over RCCL; STRONG scaling: the 100 M rows are row-range sharded over the ranks,
   every rank answers every query on its shard, ONE all-gather exchanges the per-shard top-k.)

Prints ONE JSON line on rank 0.

`roofline` describes the dominant kernel of the TIMED steps, measured with HIP events on the library's own stream
inside the timed region:
  * batches of > 16 queries run the collect scan on the matrix cores in FP4 (csrc/mfma_scan.hip): bound "mfma",
    achieved = (rows x queries x 64-bit words) x 128 operations / launch time, peak = dense FP4 MFMA rate;
  * with `--opt mfma=0` the XOR + popcount kernel runs: cache-blocked it is bound by VALU issue (bound "valu",
    in (row, query, word) triples per second against the measured issue rates of v_xor / v_bcnt / v_min3), and with
    `--opt stretch_mb=0` every pass streams from HBM (bound "hbm", algorithmic bytes / time).
After the timed region (N = 1) the same step runs a few more times in the two other regimes, so that every default run
also carries the HBM-roofline evidence `north_star` asks for (`roofline_streaming`) and the VALU kernel's figure
(`roofline_valu`).

`cpu_baseline` = the oracle (CPU restatement, OpenMP) timed on the host cores over the same 100 M rows (rank 0, N = 1
only); its answers are compared bit for bit with the GPU's for the same queries (`parity_checked_queries`; a mismatch
fails the run).  Beside it: `py_memory_style` (the reference's memory:// loop restated, one core) and `usearch`
("unavailable": the HNSW wheels cannot be installed offline) -- SURVEY.md section 8d.
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
MASK64 = (1 << 64) - 1

# peaks (MI355X_MICROARCH.md; profiles/r02_micro_valu2.txt for the integer VALU issue rates)
HBM_PEAK_GBS = 8000.0                       # HBM3E spec; 6.29 TB/s is the guide's measured copy ceiling
SIMDS, CLOCK_HZ = 1024, 2.4e9               # 256 CUs x 4 SIMDs, max clock
MFMA_FP4_PEAK_TOPS = SIMDS * 4096 * CLOCK_HZ / 1e12     # v_mfma_scale_f32_32x32x64_f8f6f4 (FP4): 32*32*64*2 ops per 32 cycles per SIMD = 10 066 TOP/s dense
# XOR + popcount kernel, per wave64 and (row, query): 2 v_xor + 2 v_bcnt per 64-bit word and half a v_min3 per pair, each at the
# 4 cycles a wave64 instruction takes in this mix (profiles/r02_micro_valu3.txt: 35 cycles per 9 instructions)


def valu_peak_gtriples(words):
    """(row, query, word) triples per second at full VALU issue: 64 lanes / ((4 + 0.5 / W) instructions x 4 cycles) per SIMD."""
    return SIMDS * CLOCK_HZ * 64 / ((4.0 + 0.5 / words) * 4.0) / 1e9
PMC_PROFILE = os.path.join("profiles", "r03_pmc_fetch_size.json")
if os.path.exists(os.path.join(ROOT, "profiles", "r04_pmc_fetch_size.json")):
    PMC_PROFILE = os.path.join("profiles", "r04_pmc_fetch_size.json")


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & MASK64
    return x ^ (x >> 31)


def make_queries(nq, rows, words, batch=0):
    """
    SURVEY.md section 8d: random queries; every 4th is a stored code with f in {0,1,3,7} low bits flipped.  `batch` b draws from
    seeds SEED_Q + 1000 b / SEED_P + 1000 b: the timed region rotates through several DISTINCT batches.
    """
    q = np.zeros((nq, words), dtype=np.uint64)
    planted = {}
    seed_q, seed_p = SEED_Q + 1000 * batch, SEED_P + 1000 * batch
    for j in range(nq):
        if j % 4 == 0:
            r = splitmix64(seed_p + j) % rows
            f = (0, 1, 3, 7)[(j // 4) % 4]
            for w in range(words):
                q[j, w] = splitmix64(SEED_CODES + 4 * r + w)
            q[j, words - 1] ^= np.uint64(f)
            planted[j] = (r, bin(f).count("1"))
        else:
            for w in range(words):
                q[j, w] = splitmix64(seed_q + 4 * j + w)
    return q, planted


def roofline_of(st, args, words, regime):
    """The roofline object of one measured leg from the engine's statistics (HIP events around every collect launch)."""
    if not st["scan_launches"] or st["scan_ms"] <= 0:
        return None
    streaming = regime.startswith("streaming")
    mfma = st["scan_mfma_launches"] == st["scan_launches"]          # the collect launches ran on the matrix cores
    if not streaming:
        # Levels and collect pass are the SAME kernel (MODE_BOTH / MODE_STRETCH instantiations of one body) walking consecutive
        # stretches of the rows; together they are ~96 % of a step.  The roofline covers all of their launches.  (The HBM-streaming
        # leg keeps to the collect launches: its levels are cache-resident by construction.)
        st = dict(st)
        for a, b in (("scan_launches", "level_launches"), ("scan_pair_words", "level_pair_words"), ("scan_mfma_launches", "level_mfma_launches"),
                     ("scan_ms", "level_ms"), ("scan_bytes", "sample_bytes")):
            st[a] = st[a] + st[b]
    secs = st["scan_ms"] / 1e3
    launches = st["scan_launches"]
    out = {
        "launches": launches,
        "avg_launch_ms": st["scan_ms"] / launches,
        # SURVEY 8d's accounting: rows x 8 x words x passes of T_q queries.  The XOR + popcount kernel really streams that;
        # the matrix-core kernel reads the rows ONCE per chunk of up to 1 024 queries, so there the figure is only what the
        # same work would have cost in passes of T_q queries (`section8d_bytes_equivalent`), never bytes it moved
        ("section8d_bytes_equivalent" if mfma else "algorithmic_bytes_per_launch"): st["scan_bytes"] / launches,
        "triples_per_launch": st["scan_pair_words"] / launches,            # (row, query, 64-bit word)
        "scan_ms_per_step": st["scan_ms"] / max(1, st["searches"]),        # device time of the scan launches of one step
        "regime": regime,
    }
    if mfma:
        ops = st["scan_pair_words"] * 128.0                                # 64 multiply-adds per triple
        # SQ counters of the same kernel (profiles/r02_pmc_sq.json): the matrix pipe is busy ~45 % of the cycles and vector issue
        # (the fold: 16 results per lane per 1 024 pairs + the MFMAs' own issue slots) ~90 %; the chip holds ~2.05 GHz.
        # k <= 512: ONE launch per step (MODE_SELF, thresholds tighten themselves); larger k: threshold levels + collect pass.
        single = st["level_launches"] == 0
        packed = st.get("mfma_pack_launches", 0) > 0
        out.update({"bound": "mfma", "kernel": "isk::%s (v_mfma_f32_32x32x64_f8f6f4, FP4 operands), %s" % (
                        "mfma_pack_kernel: two row tiles per accumulator, v_pk_minimum3_f16 fold" if packed else "mfma_scan_kernel<W=%d>" % words,
                        "one self-tightening pass per step" if single else "levels + collect"),
                    "achieved": ops / secs / 1e12, "peak": MFMA_FP4_PEAK_TOPS, "unit": "TOP/s (FP4, dense)",
                    "rows_bytes_per_launch": st["scan_pair_words"] / max(1, st["queries"] // max(1, st["searches"])) * 8 / launches,
                    "co_limiter": "the chip's clock under matrix load (~1.9 of the 2.4 GHz the peak assumes: profiles/r03_proto_pack_scan.txt)" if packed
                                  else "vector issue: the per-result fold shares the SIMD's issue port with the MFMAs"})
    elif streaming:
        out.update({"bound": "hbm", "kernel": "isk::scan_adapt_kernel / scan_kernel (XOR + popcount), one pass of T_q queries per table read",
                    "achieved": st["scan_bytes"] / 1e9 / secs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "measured_copy_ceiling_GBs": 6290.0, "measured_read_ceiling_GBs": 7050.0})
    else:
        out.update({"bound": "valu", "kernel": "isk::scan_adapt_kernel / scan_kernel (XOR + popcount), rows served by L2 / Infinity Cache",
                    "achieved": st["scan_pair_words"] / 1e9 / secs, "peak": valu_peak_gtriples(words), "unit": "G (row, query, word) triples/s",
                    "peak_derivation": "1024 SIMDs x 2.4 GHz x 64 lanes / ((2 v_xor + 2 v_bcnt per word + 1/2 v_min3 per pair) x 4 cycles per wave64 "
                                       "instruction in this mix; profiles/r02_micro_valu3.txt)",
                    "algorithmic_GBs": st["scan_bytes"] / 1e9 / secs})
    out["frac"] = out["achieved"] / out["peak"]
    return out


def attach_traffic(roof, key):
    """
    HBM traffic of the kernel per launch, from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE in a run of its own,
    gfx950 x2 correction), scaled to this run's launch size.  Never measured inside this run: the profile names the
    commit it was taken at and must be regenerated when the kernels change.
    """
    roof["traffic"] = None
    roof["traffic_measured_in_run"] = False
    try:
        with open(os.path.join(ROOT, PMC_PROFILE)) as f:
            pmc = json.load(f)
        entry = pmc["regimes"][key]
        roof["traffic"] = entry["corrected_bytes_per_launch"] * roof["triples_per_launch"] / entry["triples_per_launch"]
        roof["traffic_source"] = "%s [%s], kernels at commit %s" % (PMC_PROFILE, key, pmc.get("commit", "?"))
        # the memory picture of the same launch: HBM bytes / launch time against the 8 TB/s peak
        roof["hbm_traffic_frac"] = roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / (HBM_PEAK_GBS * 1e9)
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=100_000_000, help="total rows of the index (all ranks together)")
    ap.add_argument("--queries", type=int, default=1024, help="queries per step")
    ap.add_argument("--batches", type=int, default=8, help="distinct query batches the steps rotate through (gate, settle, warm-up, timed)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--nbytes", type=int, default=8, help="code length in bytes (8 = 64-bit)")
    ap.add_argument("--metric", choices=["hamming", "nphd"], default="hamming", help="table metric (nphd: every row --nbytes long, queries too)")
    ap.add_argument("--tq", type=int, default=8, help="queries per pass of the XOR + popcount kernel (8|16); 8 keeps a streaming scan HBM-bound")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=0, help="queries timed (and parity-checked) on the CPU (0 = auto, ~10-30 s)")
    ap.add_argument("--no-profile", action="store_true", help="do not time scan launches with HIP events")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE configs 3 and 5 after the timed region")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the VALU / HBM-streaming measurements after the timed region")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rehearsal of the N > 1 code path on a one-GPU box: every rank uses cuda:0 and the process group runs over gloo "
                         "(ranks sharing a GPU cannot form an RCCL communicator); the number it prints is not a scaling result")
    ap.add_argument("--force-collective", action="store_true", help="initialise RCCL and run the all-gather even with one rank (rehearsal)")
    ap.add_argument("--settle-steps", type=int, default=40, help="untimed steps before the warm-up while the GPU leaves its idle power state (0: none)")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (tuning experiments)")
    args = ap.parse_args()

    # ONE JSON line on stdout: RCCL prints a version banner to stdout when a communicator is created, so everything
    # below runs with fd 1 pointing at stderr and the line is written to the real stdout at the end
    real_stdout = os.dup(1)
    os.dup2(2, 1)

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
    opts = {"mfma": 1, "stretch_mb": 128, "mfma_min_queries": 17}
    for kv in args.opt:
        name, _, val = kv.partition("=")
        engine.set_option(name, int(val))
        opts[name] = int(val)
    nphd = args.metric == "nphd"
    table = engine.open_table(_lib.METRIC_NPHD if nphd else _lib.METRIC_HAMMING, 1, args.nbytes)
    q_nbytes = np.full(args.queries, args.nbytes, dtype=np.uint8) if nphd else None
    lo, hi = shard_range(args.rows, rank, world)
    table.add_synthetic(args.nbytes, hi - lo, SEED_CODES, first_row=lo, key_base=0)
    sharded = ShardedTable(HipShardOps(table, device), always_gather=args.force_collective)

    # Several DISTINCT batches, taken in rotation by every step of the run (VERDICT r3 item 2 / ADVICE r3): a step starts under the
    # k-th distance the PREVIOUS batch ended at (the engine's threshold hint), so identical batches would make that hint exact.
    n_batches = max(1, args.batches)
    batches = [make_queries(args.queries, args.rows, words, b) for b in range(n_batches)]
    q = batches[0][0]
    turn = [0]

    def step():
        b = turn[0] % n_batches
        turn[0] += 1
        return b, sharded.search(batches[b][0], q_nbytes, args.k)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # correctness gate outside the timed region, EVERY batch once: planted neighbours must come back first, lists sorted and
    # full; the answers are kept -- two batches are compared with the oracle's below (cpu_baseline), and whatever a measured
    # leg returns for a batch last must equal its gate answer
    gate = []
    for b in range(n_batches):
        _, ans = step()
        keys, ham, pbits, cnt = ans
        for j, (r, f) in batches[b][1].items():
            assert int(cnt[j]) == min(args.k, args.rows), (b, j, cnt[j])
            assert int(ham[j, 0]) <= f, f"batch {b}: planted neighbour of query {j} not found: {ham[j, :3]} vs {f}"
            if f == 0:
                assert int(keys[j, 0]) == r or int(ham[j, 0]) == 0
        assert np.all(np.diff(ham.astype(np.int64), axis=1)[:, : args.k - 1] >= 0), "results not sorted"
        gate.append(ans)
    first = gate[0]
    batches_compared = set()

    def measure(steps):
        """`steps` steps with per-launch HIP events; returns (seconds, statistics).  Every batch's LAST answer must equal its gate answer."""
        fence()
        engine.stats(reset=True)
        if not args.no_profile:
            engine.set_option("profile", 1)
        fence()
        last = {}
        t0 = time.perf_counter()
        for _ in range(steps):
            b, out = step()
            last[b] = out
        fence()
        el = time.perf_counter() - t0
        engine.set_option("profile", 0)
        st = engine.stats(reset=True)
        for b, out in last.items():
            for name, x, y in zip(("keys", "hamming", "prefix_bits", "count"), out, gate[b]):
                if not np.array_equal(x, y):
                    raise SystemExit(f"PARITY FAILURE: the last measured step of batch {b} returned different {name} than its gate step")
            batches_compared.add(b)
        return el, st

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # The UNSETTLED figure first (ADVICE r2 / VERDICT r2): W warm-up steps and K timed steps of a process that has done nothing
    # but build the index and answer the gate step -- what `--steps K --warmup W` meant before the settle loop existed.
    for _ in range(args.warmup):
        step()
    el_unsettled, _ = measure(args.steps)
    el_unsettled = max_over_ranks(el_unsettled)

    # Clock settle, before the W warm-up steps and outside every timing: a fresh process finds the GPU in its idle power
    # state, and the matrix-core kernel of this step takes ~15 steps (~50 ms) to reach its steady clock (3.46 -> 2.86 ms per
    # launch in profiles/r02_kernel_stats_mfma.csv: first / fastest launch of a fresh process).  A serving process is never in that
    # state for longer than its first request, so the bench runs untimed steps until five consecutive steps are within 1 % of
    # the five before them (at most `--settle-steps`, the same number on every rank), and says how many it took.
    settle_steps = 0
    if args.settle_steps > 0:
        hist = []
        for _ in range(args.settle_steps):
            t0 = time.perf_counter()
            step()
            hist.append(time.perf_counter() - t0)
            settle_steps += 1
            if world == 1 and len(hist) >= 10 and abs(sum(hist[-5:]) - sum(hist[-10:-5])) <= 0.01 * sum(hist[-10:-5]):
                break
    for _ in range(args.warmup):
        step()
    elapsed, st = measure(args.steps)
    elapsed = max_over_ranks(elapsed)

    batched = args.queries > args.tq
    # (64-bit codes take the packed matrix-core kernel from 9 queries: engine option mfma_pack_min_queries)
    mfma_from = min(opts["mfma_min_queries"], opts.get("mfma_pack_min_queries", 9)) if words == 1 and opts.get("mfma_pack", 1) else opts["mfma_min_queries"]
    mfma_on = bool(opts["mfma"]) and args.queries >= mfma_from and st["scan_mfma_launches"] > 0

    def regime_name(mfma, stretch_mb):
        if mfma:
            return "matrix cores: each block keeps up to %d expanded queries in LDS, the rows cross the memory system once per chunk" % (1024 // words)
        if stretch_mb and batched:
            return ("cache-blocked: all query groups of a launch share stretches of <= %d MB, read from HBM once and from the 256 MB Infinity "
                    "Cache afterwards" % stretch_mb)
        return "streaming: every pass of T_q queries reads the rows from HBM"

    roof = roofline_of(st, args, words, regime_name(mfma_on, opts["stretch_mb"])) if not args.no_profile else None
    if roof:
        attach_traffic(roof, "mfma" if mfma_on else ("valu_blocked" if opts["stretch_mb"] and batched else "valu_streaming"))

    # Outside the timed region (one GPU): the same step in the other regimes.
    extra = {}
    # The engine starts a step under the k-th distance the previous batch of its size ended at (+ 2 bits) instead of a bootstrap
    # sample's threshold, and verifies that this held k rows for every query ("hints"): every row is still
    # scanned, nothing is cached, and the hint is ONE number per batch -- but it is state carried from step to step, so the
    # same steps without it are timed beside the headline.
    if world == 1 and (st["spec_hits"] or st["spec_misses"]) and "speculate" not in opts:
        engine.set_option("speculate", 0)
        step()
        el0, s0_ = measure(min(args.steps, 10))
        engine.set_option("speculate", 1)
        step()
        extra["threshold_hint"] = {
            "steps_started_under_a_hint": int(st["spec_hits"]), "hints_that_did_not_hold": int(st["spec_misses"]),
            "what": "steps of the timed region whose thresholds started at the previous step's worst k-th distance + 2 (verified; exact either way)",
            "distinct_batches_in_rotation": n_batches,
            "value_without_hints": args.queries * min(args.steps, 10) / el0,
            "ms_per_step_without_hints": el0 / min(args.steps, 10) * 1e3,
            "without_hints": "engine option speculate = 0: bootstrap sample + single pass (batches <= 128 queries: bootstrap + levels), same process, after the timed region",
        }
    if world == 1 and not args.no_profile and not args.no_extra_legs and batched:
        legs = []
        if mfma_on:
            legs.append(("roofline_valu", {"mfma": 0, "stretch_mb": opts["stretch_mb"] or 128}))
        if mfma_on or opts["stretch_mb"]:
            legs.append(("roofline_streaming", {"mfma": 0, "stretch_mb": 0}))
        for name, o in legs:
            for k_, v_ in o.items():
                engine.set_option(k_, v_)
            step()
            el, s2 = measure(5)
            r2 = roofline_of(s2, args, words, regime_name(False, o["stretch_mb"]))
            if r2:
                r2["queries_per_s"] = args.queries * 5 / el
                r2["what"] = "same workload, outside the timed region, engine options %s" % o
                attach_traffic(r2, "valu_blocked" if o["stretch_mb"] else "valu_streaming")
                extra[name] = r2
        engine.set_option("mfma", opts["mfma"])
        engine.set_option("stretch_mb", opts["stretch_mb"])

    if sharded.hint_hits or sharded.hint_misses:
        # (sharded steps: every shard starts under the GLOBAL k-th distance of the previous step + 2; the merged lists are checked)
        extra["threshold_hint"] = {"steps_started_under_a_hint": sharded.hint_hits, "hints_that_did_not_hold": sharded.hint_misses,
                                   "what": "all steps of this process (gate, settle, warm-up, timed): shards started under the previous step's global k-th "
                                           "distance + 2 (+ 1 from k = 64); a step stands only if every merged list holds k rows (sharded.py)"}
    total_queries = args.queries * args.steps
    qps = total_queries / elapsed
    out = {
        "metric": "queries/sec (+ roofline of the dominant kernel), 64-bit Hamming k=10 over 100M codes, exact top-k",
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "settle_steps": settle_steps,          # untimed, before the warm-up: see the comment at the loop
        "value_unsettled": total_queries / el_unsettled,   # the first `steps` steps after `warmup` of the fresh process, no settle loop
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "fp4 (e2m1 0/1 x +-1 products, f32 accumulation: exact integers)" if mfma_on else "u64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.rows} x {args.nbytes * 8}-bit codes, brute-force {'NPHD' if nphd else 'Hamming'} k={args.k}, "
                        f"{args.queries} queries/step, rows sharded over {world} GPU(s)",
            "rows_total": args.rows,
            "rows_per_gpu": hi - lo,
            "code_bits": args.nbytes * 8,
            "k": args.k,
            "queries_per_step": args.queries,
            "distinct_query_batches": n_batches,
            "queries_per_pass": args.tq,
            "scan": "FP4 MFMA" if mfma_on else "XOR + popcount",
            "parallelism": f"row-shard x{world}, one all-gather of per-shard top-k",
        },
        "roofline": roof,
        "fallback_queries": st["fallback_queries"],
        # every batch of the rotation was gated before the timed region; these batches' last measured answers were compared with
        # their gate answers (all legs together)
        "batches_gated": n_batches,
        "batches_compared_after_measurement": sorted(batches_compared),
    }
    if dist.is_initialized():
        # what the communicator itself reports (a SCALE record can then show that RCCL saw N ranks)
        out["world_size_seen"] = dist.get_world_size()
        out["collective_backend"] = dist.get_backend()
        try:
            out["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # pragma: no cover - depends on the torch build
            out["rccl_version"] = f"unavailable ({type(e).__name__})"
    out.update(extra)

    if rank == 0 and world == 1 and not args.no_other_configs and not args.no_cpu_baseline and args.nbytes == 8 and not nphd:
        out["other_configs"] = other_configs(engine, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        second = (batches[1][0], gate[1]) if n_batches > 1 else None
        out["cpu_baseline"] = cpu_baseline(args, q, words, first, second)
        out["parity_checked_queries"] = out["cpu_baseline"]["parity_checked_queries"]

    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
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


def py_memory_style(n_assets=2500, n_queries=200):
    """
    SURVEY.md section 8d-ii: what the reference's memory:// backend does per search (iscc_search/indexes/memory/index.py:204-232),
    restated: normalise the query, compare the iscc_code STRING with every stored asset, score 1.0.  No distance is
    computed; one core, GIL-bound by construction.  Config 1's shape: 2 500 assets x 4 units = 10 000 64-bit units.
    """
    import base64

    rng = np.random.default_rng(0)
    codes = ["ISCC:" + base64.b32encode(rng.integers(0, 256, size=34, dtype=np.uint8).tobytes()).decode().rstrip("=") for _ in range(n_assets)]
    store = {i: {"iscc_id": i, "iscc_code": c, "units": [c[:21]] * 4, "metadata": None} for i, c in enumerate(codes)}
    t0 = time.perf_counter()
    for j in range(n_queries):
        code = codes[j]
        types = {u: 1.0 for u in store[j]["units"]}
        hits = [(a["iscc_id"], 1.0, types, a["metadata"]) for a in store.values() if code and a["iscc_code"] and a["iscc_code"] == code][:10]
        assert len(hits) == 1
    dt = time.perf_counter() - t0
    return {"value": n_queries / dt, "unit": "searches/s", "cores": 1, "kind": "port",
            "sample": f"{n_queries} iscc_code equality searches over {n_assets} assets (config 1: 10 000 units); no distance computed"}


def slab_oracle(rows, nbytes, q, k, nphd, key_words, slab=12_500_000):
    """The oracle over a synthetic table too large for one host array: exact top-k per slab of rows, merged under (distance, key)."""
    from oracle import oracle_splitmix64_fill, oracle_topk

    words = (nbytes + 7) // 8
    qn = np.full(len(q), nbytes, dtype=np.uint8) if nphd else None
    best = [[] for _ in range(len(q))]
    for lo in range(0, rows, slab):
        n = min(slab, rows - lo)
        cw = np.ascontiguousarray(np.stack([oracle_splitmix64_fill(n, SEED_CODES, first=lo, stride=4, lane=w) for w in range(words)], axis=1))
        rk = np.arange(lo, lo + n, dtype=np.uint64)
        keys = np.stack([np.zeros(n, dtype=np.uint64), rk], axis=1) if key_words == 2 else rk
        lens = np.full(n, nbytes, dtype=np.uint8) if nphd else None
        kk, hh, _, cc = oracle_topk(1 if nphd else 0, keys, cw, lens, q, qn, k, fixed_nbytes=0 if nphd else nbytes)
        for i in range(len(q)):
            c = int(cc[i])
            low = kk[i, :c, 1] if key_words == 2 else kk[i, :c]
            best[i].extend(zip(hh[i, :c].tolist(), low.tolist()))
    return [sorted(c)[:k] for c in best]


def other_configs(engine, args):
    """
    BASELINE configs 3 and 5 in the same process, after the timed region (N = 1): a few steps each with HIP events around the
    scan launches, four queries checked against the oracle (slab-wise), the last step compared with the first.
    """
    from iscc_search_amd import _lib

    out = {}
    # sized from --rows: the default 100 M-row run carries config 3 at its full size and config 5's table at 10 M rows
    rows3, rows5 = args.rows, max(1, args.rows // 10)
    cases = (
        ("config3", f"{rows3} x 256-bit ISCC-UNITs, NPHD, 1 024 queries, k = 10", dict(rows=rows3, nbytes=32, nphd=True, key_words=1, nq=1024, k=10, steps=3)),
        ("config5_shape", f"{rows5} x 128-bit chunk fingerprints, 128-bit keys, 512 queries, k = 400 (one of config 5's three tables)",
         dict(rows=rows5, nbytes=16, nphd=False, key_words=2, nq=512, k=400, steps=5)),
    )
    for name, what, c in cases:
        words = (c["nbytes"] + 7) // 8
        t = engine.open_table(_lib.METRIC_NPHD if c["nphd"] else _lib.METRIC_HAMMING, c["key_words"], c["nbytes"])
        try:
            t.add_synthetic(c["nbytes"], c["rows"], SEED_CODES)
            q, planted = make_queries(c["nq"], c["rows"], words)
            qn = np.full(c["nq"], c["nbytes"], dtype=np.uint8) if c["nphd"] else None
            first = t.search(q, qn, c["k"])
            engine.stats(reset=True)
            engine.set_option("profile", 1)
            t0 = time.perf_counter()
            for _ in range(c["steps"]):
                last = t.search(q, qn, c["k"])
            el = time.perf_counter() - t0
            engine.set_option("profile", 0)
            st = engine.stats(reset=True)
            for a, b in zip(first, last):
                if not np.array_equal(a, b):
                    raise SystemExit(f"PARITY FAILURE ({name}): two steps returned different results")
            pick = [0, 1, 2, c["nq"] - 1]
            exp = slab_oracle(c["rows"], c["nbytes"], q[pick], c["k"], c["nphd"], c["key_words"])
            keys, ham, _, cnt = first
            for j, e in zip(pick, exp):
                low = keys[j, :, 1] if c["key_words"] == 2 else keys[j]
                got = list(zip(ham[j, : int(cnt[j])].tolist(), low[: int(cnt[j])].tolist()))
                if got != e:
                    raise SystemExit(f"PARITY FAILURE ({name}): query {j} differs from the oracle")
            roof = roofline_of(st, args, words, "matrix cores")
            out[name] = {"workload": what, "queries_per_s": c["nq"] * c["steps"] / el, "ms_per_step": el / c["steps"] * 1e3, "steps": c["steps"],
                         "parity_checked_queries": len(pick), "fallback_queries": st["fallback_queries"],
                         "roofline": None if roof is None else {k_: roof[k_] for k_ in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches")}}
        finally:
            t.drop()
    out["config5_end_to_end"] = config5_end_to_end(engine, rows5)
    out["config1_protocol"] = config1_protocol()
    return out


def config5_end_to_end(engine, chunks):
    """
    BASELINE config 5 through the simprint interface (`UsearchSimprintIndex.search_raw`, usearch_core.py:137-269): one table of
    `chunks` 128-bit chunk fingerprints (40 per asset, 128-bit chunk-pointer keys), 512 query simprints, limit 20 x oversampling 20
    = 400 neighbours each, threshold 0.75, chunk detail, device document frequencies.  ONE library call per request
    (isccsearch_simprint_score): search, threshold, best chunk per asset and query, IDF-weighted scores, sort, cut -- all on the device.
    Checked once against the same request scored on the host from the neighbour lists (round 3's path): equal assets, equal float64 scores.
    """
    from iscc_search_amd.simprint import HipSimprintIndex

    per_asset, nq, reps = 40, 512, 9
    rng = np.random.default_rng(0)
    idx = HipSimprintIndex(engine, ndim=128)
    try:
        first = None
        for lo in range(0, chunks, 1 << 20):
            n = min(1 << 20, chunks - lo)
            rows = np.arange(lo, lo + n, dtype=np.uint64)
            keys = np.stack([rows // np.uint64(per_asset) + np.uint64(1),
                             ((rows % np.uint64(per_asset)) * np.uint64(100) << np.uint64(32)) | np.uint64(100)], axis=1)
            vecs = rng.integers(0, 256, size=(n, 16), dtype=np.uint8)
            idx._index.add(keys, vecs, trusted_unique=True)
            if first is None:
                first = vecs[:nq].copy()
        first[:, 0] ^= 3                                          # two flipped bits: near, not equal
        simprints = [bytes(r) for r in first]
        kw = dict(limit=20, threshold=0.75, detailed=True, total_assets=chunks // per_asset, device_doc_freq=True)
        inner, inside = idx._index.score_assets, [0.0]

        def timed_call(*a, **k):
            t0 = time.perf_counter()
            try:
                return inner(*a, **k)
            finally:
                inside[0] += time.perf_counter() - t0

        idx._index.score_assets = timed_call
        res = idx.search_raw(simprints, **kw)                     # builds the frequency column
        rows_t = []
        for _ in range(reps):
            inside[0] = 0.0
            t0 = time.perf_counter()
            res = idx.search_raw(simprints, **kw)
            rows_t.append((time.perf_counter() - t0, inside[0]))
        rows_t.sort()
        total, dev = rows_t[reps // 2]
        host = idx._search_raw_host(simprints, 20, 0.75, True, None, chunks // per_asset, True)
        same = [(r.iscc_id_body, r.score, r.matches) for r in res] == [(r.iscc_id_body, r.score, r.matches) for r in host]
        if not same:
            raise SystemExit("PARITY FAILURE (config5_end_to_end): device scoring differs from the host scoring of the same neighbour lists")
        return {"workload": f"{chunks} x 128-bit chunk fingerprints of {chunks // per_asset} assets, {nq} query simprints, limit 20 x oversampling 20, "
                            "threshold 0.75, chunk detail, device document frequencies (search_raw end to end)",
                "ms_per_request": total * 1e3, "ms_inside_the_library": dev * 1e3, "ms_python": (total - dev) * 1e3,
                "requests_per_s": 1.0 / total, "query_simprints_per_s": nq / total, "assets_returned": len(res),
                "scores_equal_host_scoring": same, "reps": reps, "statistic": "median"}
    finally:
        idx.close()


def config1_protocol(n_assets=2500, n_queries=200):
    """
    BASELINE config 1 through the real backend: 10 000 random 64-bit units (2 500 assets x 4) via IsccIndexProtocol on the GPU
    (HipIndexManager: `add_assets` iscc_search/indexes/usearch/index.py:194-537, `search_assets` :735-881), one thread.
    """
    from iscc_search_amd import codec
    from iscc_search_amd.index import HipIndexManager
    from iscc_search_amd.schema import IsccEntry, IsccIndex, IsccQuery

    rng = np.random.default_rng(0)
    assets = []
    for i in range(n_assets):
        units = [codec.encode_unit(mt, 0, 0, rng.integers(0, 256, size=8, dtype=np.uint8).tobytes())
                 for mt in (codec.MT_META, codec.MT_CONTENT, codec.MT_DATA, codec.MT_INSTANCE)]
        assets.append(IsccEntry(iscc_id=codec.iscc_id_from_int(((1_000_000 + i) << 12) | (i & 0xFFF), 0), iscc_code=codec.gen_iscc_code(units), units=units))
    m = HipIndexManager("hip:///")
    try:
        m.create_index(IsccIndex(name="c1"))
        t0 = time.perf_counter()
        for i in range(0, n_assets, 500):
            m.add_assets("c1", assets[i : i + 500])
        add_s = time.perf_counter() - t0
        queries = [IsccQuery(iscc_code=a.iscc_code) for a in assets[:n_queries]]
        m.search_assets("c1", queries[0], limit=10)
        t0 = time.perf_counter()
        for qy in queries:
            r = m.search_assets("c1", qy, limit=10)
            if r.global_matches[0].score != 1.0:
                raise SystemExit("PARITY FAILURE (config1_protocol): an indexed asset is not its own best match")
        search_s = time.perf_counter() - t0
    finally:
        m.close()
    return {"workload": f"{n_assets} assets x 4 units = {n_assets * 4} random 64-bit ISCC-UNITs through HipIndexManager (hip:///), one thread",
            "add_assets_per_s": n_assets / add_s, "search_assets_per_s": n_queries / search_s, "ms_per_search": search_s / n_queries * 1e3,
            "units_per_query": 4, "limit": 10}


def cpu_baseline(args, q, words, gpu, second=None):
    """
    The oracle (CPU restatement of the same exact search) on the host cores, bounded to ~10-30 s; its answers check the GPU's for
    the first batch of the rotation and, `second` = (queries, GPU answer), for up to 64 queries of another one.
    """
    from oracle import oracle_num_threads, oracle_splitmix64_fill, oracle_topk

    rows = args.rows
    cols = [oracle_splitmix64_fill(rows, SEED_CODES, stride=4, lane=w) for w in range(words)]
    code_words = np.ascontiguousarray(np.stack(cols, axis=1))
    if args.nbytes % 8:
        code_words[:, -1] &= np.uint64((MASK64 << (8 * (8 - args.nbytes % 8))) & MASK64)
    keys = np.arange(rows, dtype=np.uint64)
    threads = usable_cores(oracle_num_threads())
    nphd = args.metric == "nphd"
    lens = np.full(rows, args.nbytes, dtype=np.uint8) if nphd else None

    def run(qs, nthreads):
        qn = np.full(len(qs), args.nbytes, dtype=np.uint8) if nphd else None
        return oracle_topk(1 if nphd else 0, keys, code_words, lens, qs, qn, args.k, fixed_nbytes=0 if nphd else args.nbytes, threads=nthreads)

    nq = args.cpu_queries
    if nq <= 0:
        # calibrate on 8 queries, then size the sample for ~15 s
        t0 = time.perf_counter()
        run(q[:8], threads)
        per_q = (time.perf_counter() - t0) / 8
        nq = int(max(8, min(args.queries, 15.0 / max(per_q, 1e-6))))
    nq = min(nq, args.queries)
    reps = 1
    if args.cpu_queries <= 0 and nq == args.queries:
        reps = int(max(1, min(16, round(12.0 / max(per_q * nq, 1e-6)))))    # ~12 s of CPU work
    t0 = time.perf_counter()
    for _ in range(reps):
        exp = run(q[:nq], threads)
    dt = time.perf_counter() - t0
    # parity: the oracle's answers for those queries against what the GPU returned for them before the timed region
    for name, g, e in zip(("keys", "hamming", "prefix_bits", "count"), gpu, exp):
        if not np.array_equal(g[:nq], e):
            bad = int(np.nonzero(np.any((g[:nq] != e).reshape(nq, -1), axis=1))[0][0])
            raise SystemExit(f"PARITY FAILURE: GPU {name} differ from the oracle for query {bad}: {g[bad]} vs {e[bad]}")
    second_checked = 0
    if second is not None:
        q2, gpu2 = second
        second_checked = min(64, len(q2), nq)
        exp2 = run(q2[:second_checked], threads)
        for name, g, e in zip(("keys", "hamming", "prefix_bits", "count"), gpu2, exp2):
            if not np.array_equal(g[:second_checked], e):
                raise SystemExit(f"PARITY FAILURE: GPU {name} of the second query batch differ from the oracle")
    # one core, on a 10 M-row slice (scaled linearly to the full table; labelled as extrapolated)
    slice_rows = min(rows, 10_000_000)
    t1 = time.perf_counter()
    oracle_topk(1 if nphd else 0, keys[:slice_rows], code_words[:slice_rows], None if lens is None else lens[:slice_rows], q[:4],
                np.full(4, args.nbytes, dtype=np.uint8) if nphd else None, args.k, fixed_nbytes=0 if nphd else args.nbytes, threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "single_core": {"value": 4 / dt1 * slice_rows / rows, "unit": "queries/s", "cores": 1,
                        "sample": f"4 queries over a {slice_rows}-row slice, extrapolated linearly to {rows} rows"},
        "value": nq * reps / dt,
        "unit": "queries/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{nq * reps} queries ({reps} x {nq} of the step's batch) over all {rows} rows ({dt:.1f} s of CPU work, {threads} OpenMP threads = "
                  f"the CPUs this process may use; rows split across threads, cache-blocked over all queries)",
        "GBs": nq * reps * rows * 8 * words / dt / 1e9,
        "parity_checked_queries": nq,
        "parity_checked_queries_second_batch": second_checked,
        "parity": f"keys, hamming, prefix_bits and counts of the GPU's answer to the first {nq} queries of batch 0 (and {second_checked} of batch 1) are bit-identical to the oracle's",
        "py_memory_style": py_memory_style(),
        "usearch": "unavailable (iscc-usearch 0.8.1 / usearch-iscc 2.24.6 wheels are not in the image and cannot be installed offline; not estimated)",
    }


if __name__ == "__main__":
    main()
