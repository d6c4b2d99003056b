#!/bin/bash
# Kernel timelines of one step at the operating points DESIGN.md quotes (run through gpurun from the repo root):
#   bash tools/timelines.sh r03   ->  gpurun_out/r03_step_timelines.txt
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/${TAG}_tl
OUT=$ROOT/gpurun_out/${TAG}_step_timelines.txt
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --no-profile --settle-steps 10 --steps 10 --warmup 2 --opt speculate=1"     # (the option named: bench.py then skips its without-hints leg, so the trace ends with ordinary steady-state steps)
: > "$OUT"
one() {   # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --output-format csv -d "$O/$name" -- $B "$@" > "$O/$name.log" 2>&1 || echo "rocprofv3 $name failed" >> "$OUT"
    echo "== $name: bench.py $*" >> "$OUT"
    grep -h '^{' "$O/$name.log" | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench line: %.0f q/s, %.3f ms per step' % (d['value'], d['ms_per_step']))" >> "$OUT" 2>&1
    python3 $ROOT/tools/timeline.py "$(ls -t $O/$name/*/*_kernel_trace.csv | head -1)" >> "$OUT" 2>&1
}
one 100m
one 12m --rows 12500000 --force-collective
one 1q --queries 1
one 8q --queries 8
one 32q --queries 32
one 64q --queries 64
one 128q --queries 128
one c3 --nbytes 32 --metric nphd
cat "$OUT"
