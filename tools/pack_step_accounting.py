#!/usr/bin/env python3
"""Summary of tools/pack_step_accounting.sh: per ISK_EXP_PACK variant, the launch time of mfma_pack_kernel<4, 1, 0> and its counters."""
import collections
import csv
import glob
import os
import sys

O = sys.argv[1]
KERNEL = "mfma_pack_kernel<4, 1, 0>"
NAMES = {0: "product kernel", 1: "no fold (16 v_pk_minimum3_f16 + v_cmp per stage)", 2: "rows expanded once (loads kept)", 4: "no looks (threshold refresh, checkers)",
         3: "no fold, rows expanded once", 15: "as 7, and no block scale on the second tile's MFMAs (no v_mfma_ld_scale_b32)", 7: "no fold, rows expanded once, no looks: MFMAs + fragment reads + row loads"}


def newest(pattern):
    hits = glob.glob(os.path.join(O, pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


print("variant                                                                    launches   ms (trace)   | under counters: ms   clock GHz   pipe busy   busy GHz   VALU/MFMA   issue occ.   wait_inst   wait_any")
for v in (0, 1, 2, 4, 3, 7, 15):
    st = newest(f"stats_{v}/**/*kernel_stats.csv")
    ms_trace, calls = float("nan"), 0
    if st:
        for r in csv.DictReader(open(st)):
            if KERNEL in r["Name"]:
                ms_trace, calls = float(r["AverageNs"]) / 1e6, int(r["Calls"])
    cc = newest(f"pmc_{v}/**/*counter_collection.csv")
    line = "%-74s %8d   %10.3f   |" % (f"{v}: {NAMES[v]}", calls, ms_trace)
    if cc:
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(cc)):
            if KERNEL in r["Kernel_Name"]:
                d = per[int(r["Dispatch_Id"])]
                d[r["Counter_Name"]] = float(r["Counter_Value"])
                d["_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if per:
            n = len(per)
            mean = collections.defaultdict(float)
            for d in per.values():
                for k, x in d.items():
                    mean[k] += x / n
            cyc = mean["GRBM_GUI_ACTIVE"] / 8            # summed over the 8 XCDs
            busy = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
            ghz = cyc / (mean["_ms"] * 1e6)
            line += "  %15.3f   %9.3f   %9.3f   %8.3f   %9.2f   %10.3f   %9.3f   %8.3f" % (
                mean["_ms"], ghz, busy, busy * ghz, mean["SQ_INSTS_VALU"] / max(mean["SQ_INSTS_MFMA"], 1), mean["SQ_INSTS_VALU"] * 4 / 1024 / cyc,
                mean["SQ_WAIT_INST_ANY"] / max(mean["SQ_WAVE_CYCLES"], 1), mean["SQ_WAIT_ANY"] / max(mean["SQ_WAVE_CYCLES"], 1))
    print(line)
