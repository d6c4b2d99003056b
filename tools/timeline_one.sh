#!/bin/bash
# kernel timeline of one step of `bench.py <args>` (run through gpurun from the repo root): bash tools/timeline_one.sh <name> <bench args...>
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
name=$1; shift
O=$ROOT/gpurun_out/tl_$name
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O" -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --no-profile --settle-steps 10 --steps 10 --warmup 2 "$@" > "$O.log" 2>&1
grep -h '^{' "$O.log" | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench line: %.0f q/s, %.3f ms per step' % (d['value'], d['ms_per_step']))"
python3 $ROOT/tools/timeline.py "$(ls -t $O/*/*_kernel_trace.csv | head -1)"
