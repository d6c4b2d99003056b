#!/usr/bin/env python3
"""
Timeline of one search step from a rocprofv3 --kernel-trace CSV: every kernel of the LAST complete step (a step starts
at boot_kernel / boot_multi_kernel / radius_init[_inline]_kernel) with its duration and the gap to the previous kernel's end.

usage: timeline.py <..._kernel_trace.csv> [steps back from the last, default 1]
"""
import csv
import sys


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    with open(path) as f:
        rows = list(csv.DictReader(f))
    seq = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"]) for r in rows))
    starts = [i for i, s in enumerate(seq) if "boot_kernel" in s[2] or "boot_multi_kernel" in s[2] or "radius_init_kernel" in s[2] or "radius_init_inline_kernel" in s[2]]
    if len(starts) < back + 1:
        raise SystemExit("not enough steps in the trace")
    a, b = starts[-back - 1], starts[-back]
    prev_end = None
    busy = 0.0
    for s0, s1, name, gx, gy in seq[a:b]:
        gap = (s0 - prev_end) / 1e3 if prev_end is not None else 0.0
        busy += (s1 - s0) / 1e3
        print("%-58s %9.1f us   gap %6.1f us   grid %s x %s" % (name[:58], (s1 - s0) / 1e3, gap, gx, gy))
        prev_end = s1
    print("step: %.1f us from first kernel start to the next step's first kernel; kernels busy %.1f us" % ((seq[b][0] - seq[a][0]) / 1e3, busy))


if __name__ == "__main__":
    main()
