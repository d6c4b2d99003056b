#!/usr/bin/env python3
"""
Turns what tools/profile_round.sh left under gpurun_out/<tag>/ into the evidence files committed under profiles/:

  <tag>_kernel_stats_<regime>.csv        rocprofv3 --kernel-trace --stats of bench.py in that regime
  <tag>_bench_under_rocprof_<regime>.json the JSON line bench.py printed in that same run (HIP events vs rocprofv3)
  <tag>_pmc_fetch_size.json              FETCH_SIZE per dispatch of the dominant kernel, gfx950 x2 correction, per regime:
                                          what bench.py's `roofline.traffic` is scaled from
  <tag>_pmc_sq.json                      SQ counters of the dominant kernel (issue / matrix-pipe occupancy, effective clock)

usage: python tools/summarise_profiles.py r02 [commit]
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
# kernel name prefixes: levels (<.., 2>) and collect (<.., 3>) are two instantiations of one body; the streaming leg keeps to the collect launches
DOMINANT = {"mfma": "mfma_pack_kernel<", "valu": "scan_adapt_kernel<8,", "streaming": "scan_adapt_kernel<8, 3>", "config3": "mfma_scan_kernel<4,"}


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)      # gpurun MERGES outputs: an older run's files may still be there
    return max(hits, key=os.path.getmtime) if hits else None


def dispatches(path, kernel):
    """{counter: [values per dispatch]}, [duration ms per dispatch] for the dispatches of `kernel` in a counter_collection.csv"""
    per = collections.defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel_Name"]:
                d = per[int(r["Dispatch_Id"])]
                d[r["Counter_Name"]] = float(r["Counter_Value"])
                d["_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                d["_grid"] = int(r["Grid_Size"])
    ids = sorted(per)
    counters = collections.defaultdict(list)
    for i in ids:
        for k, v in per[i].items():
            counters[k].append(v)
    return counters


# 1. kernel statistics + the bench line of the same run
for regime in DOMINANT:
    st = one(f"prof_stats_{regime}/**/*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(dst, f"{tag}_kernel_stats_{regime}.csv"))
    bj = os.path.join(src, f"bench_under_rocprof_{regime}.json")
    if os.path.exists(bj) and os.path.getsize(bj):
        shutil.copy(bj, os.path.join(dst, f"{tag}_bench_under_rocprof_{regime}.json"))

# 2. FETCH_SIZE
ROWS, NQ = 100_000_000, 1024
fetch = {"commit": commit, "command": "tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 [--opt mfma=0] [--opt stretch_mb=0]",
         "correction": "gfx950 FETCH_SIZE reports exactly half of a 16 B/lane coalesced stream (MI355X_MICROARCH.md, HBM): bytes = FETCH_SIZE(KB) x 1024 x 2 for the "
                       "dwordx4 loads of the XOR + popcount kernel and -- since the packed matrix-core kernel reads its rows with one dwordx4 per lane and step too "
                       "(round 3; 4 B per lane before, with the same counter value) -- of both kernels: raw and x2 figure are given, `corrected_bytes_per_launch` is x2.  "
                       "FETCH_SIZE counts what the L2s fetch from the fabric: Infinity-Cache hits are included.",
         "regimes": {}}
for regime, key in (("mfma", "mfma"), ("valu", "valu_blocked"), ("streaming", "valu_streaming")):
    cc = one(f"prof_fetch_{regime}/**/*counter_collection.csv")
    if not cc:
        continue
    c = dispatches(cc, DOMINANT[regime])
    if not c.get("FETCH_SIZE"):
        continue
    # the collect launches of one step differ in size (stretches / the remainder after the levels): report the per-step sum too
    kb = c["FETCH_SIZE"]
    n_launch = len(kb)
    with open(os.path.join(src, f"bench_under_pmc_{regime}.json")) as f:
        roof = json.load(f)["roofline"]          # the same run's own launch accounting
    entry = {
        "kernel": DOMINANT[regime], "dispatches": n_launch, "FETCH_SIZE_KB_per_dispatch": kb, "kernel_ms_per_dispatch_under_pmc": c["_ms"],
        "triples_per_launch": roof["triples_per_launch"],
        "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch", roof.get("rows_bytes_per_launch")),   # matrix cores: the rows, once
        "raw_bytes_per_launch": sum(kb) / n_launch * 1024,
        "corrected_bytes_per_launch": sum(kb) / n_launch * 1024 * 2,
    }
    fetch["regimes"][key] = entry
json.dump(fetch, open(os.path.join(dst, f"{tag}_pmc_fetch_size.json"), "w"), indent=1)

# 3. SQ counters
sq = {"commit": commit, "command": "tools/profile_round.sh: rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY "
      "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py ... --steps 3",
      "notes": "SQ_* wave counters are in quad-cycles summed over all waves; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the 1 024 SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs "
               "(MI355X_MICROARCH.md)", "kernels": {}}
DOMINANT_SQ = dict(DOMINANT, mfma_unpacked="mfma_scan_kernel<1,")        # the round-2 kernel, same box, option mfma_pack=0
for regime in ("mfma", "valu", "mfma_unpacked"):
    cc = one(f"prof_sq_{regime}/**/*counter_collection.csv")
    if not cc:
        continue
    c = dispatches(cc, DOMINANT_SQ[regime])
    if not c.get("SQ_WAVE_CYCLES"):
        continue
    mean = {k: sum(v) / len(v) for k, v in c.items() if not k.startswith("_")}
    ms = sum(c["_ms"]) / len(c["_ms"])
    cycles_per_xcd = mean["GRBM_GUI_ACTIVE"] / 8
    derived = {"kernel_ms_under_pmc": ms, "effective_clock_GHz": cycles_per_xcd / (ms * 1e-3) / 1e9,
               "valu_wave_instructions_per_simd_cycle": mean["SQ_INSTS_VALU"] / 1024 / cycles_per_xcd,
               "valu_issue_occupancy_at_4_cycles_per_instruction": mean["SQ_INSTS_VALU"] * 4 / 1024 / cycles_per_xcd,
               "matrix_pipe_busy_fraction": mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cycles_per_xcd}
    entry = {"dispatches": len(c["_ms"]), "mean_per_dispatch": mean, "derived": derived}
    # the second pass of the same kernel (other counters, own run): instruction mix per wave-level MFMA
    cc2 = one(f"prof_sq2_{regime}/**/*counter_collection.csv")
    if cc2:
        c2 = dispatches(cc2, DOMINANT_SQ[regime])
        entry["second_pass_mean_per_dispatch"] = {k: sum(v) / len(v) for k, v in c2.items() if not k.startswith("_")}
    sq["kernels"][DOMINANT_SQ[regime] + (" [option mfma_pack=0]" if regime == "mfma_unpacked" else "")] = entry
json.dump(sq, open(os.path.join(dst, f"{tag}_pmc_sq.json"), "w"), indent=1)
print(json.dumps({k: {kk: round(vv, 4) for kk, vv in v["derived"].items()} for k, v in sq["kernels"].items()}, indent=1))
print({k: (round(v["raw_bytes_per_launch"] / 1e9, 3), v["dispatches"]) for k, v in fetch["regimes"].items()})
