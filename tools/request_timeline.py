#!/usr/bin/env python3
"""
Kernel timeline of single requests out of a rocprofv3 --kernel-trace CSV: the launches between two occurrences of a marker
kernel (default: emit_kernel, the last launch of an isccsearch_simprint_score call), with start offsets and durations.

usage: python tools/request_timeline.py <kernel_trace.csv> [marker substring] [request index ...]
"""
import csv
import re
import sys


def short(name):
    name = name.replace("isksp::(anonymous namespace)::", "").replace("void ", "")
    if "rocprim" in name:
        m = re.findall(r"detail::(\w+)", name)
        name = "rocprim:" + (m[1] if len(m) > 1 else m[0] if m else "?")
    return name[:100]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marker = sys.argv[2] if len(sys.argv) > 2 else "emit_kernel"
    ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    picks = [int(a) for a in sys.argv[3:]] or [len(ends) - 1]
    for idx in picks:
        a, b = ends[idx - 1] + 1, ends[idx]
        t0 = int(rows[a]["Start_Timestamp"])
        print(f"--- request {idx} of {len(ends)}: {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us from its first launch to the end of its last")
        for r in rows[a : b + 1]:
            print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {short(r['Kernel_Name'])}")


if __name__ == "__main__":
    main()
