#!/bin/bash
# Regenerates the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r03
# Kernel-trace statistics and PMC passes are SEPARATE rocprofv3 runs (gpurun refuses the combination with other trace
# domains, and the counters perturb the timing); outputs land under gpurun_out/<tag>/prof_*; tools/summarise_profiles.py
# turns them into the files committed under profiles/.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs --no-other-configs"
run() {   # name, rocprofv3 args..., -- program args
    local name=$1; shift
    rocprofv3 "$@" > "$O/$name.log" 2>&1 || echo "rocprofv3 $name failed (see $O/$name.log)"
}
# 1. per-kernel time of the three regimes of the default workload (HIP events inside bench.py must agree)
run prof_stats_mfma      --kernel-trace --stats --output-format csv -d "$O/prof_stats_mfma"      -- $B --steps 10
run prof_stats_valu      --kernel-trace --stats --output-format csv -d "$O/prof_stats_valu"      -- $B --steps 5 --opt mfma=0
run prof_stats_streaming --kernel-trace --stats --output-format csv -d "$O/prof_stats_streaming" -- $B --steps 5 --opt mfma=0 --opt stretch_mb=0
run prof_stats_config3   --kernel-trace --stats --output-format csv -d "$O/prof_stats_config3"   -- $B --steps 5 --nbytes 32 --metric nphd
for n in mfma valu streaming config3; do grep -h '^{' "$O/prof_stats_$n.log" | tail -1 > "$O/bench_under_rocprof_$n.json"; done
# 2. HBM traffic (FETCH_SIZE) of the same three regimes
P="--steps 3 --warmup 1"
run prof_fetch_mfma      --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/prof_fetch_mfma"      -- $B $P
run prof_fetch_valu      --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/prof_fetch_valu"      -- $B $P --opt mfma=0
run prof_fetch_streaming --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/prof_fetch_streaming" -- $B $P --opt mfma=0 --opt stretch_mb=0
for n in mfma valu streaming; do grep -h '^{' "$O/prof_fetch_$n.log" | tail -1 > "$O/bench_under_pmc_$n.json"; done
# 3. SQ counters of the matrix-core kernel and of the cache-blocked XOR + popcount kernel
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
# a second SQ pass (8 slots per pass): what else the waves issue and what they wait for
SQ2=${SQ2:-"SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"}
rocprofv3 -L > "$O/counters_available.txt" 2>&1
run prof_sq_mfma --pmc $SQ --kernel-trace --output-format csv -d "$O/prof_sq_mfma" -- $B $P
run prof_sq_valu --pmc $SQ --kernel-trace --output-format csv -d "$O/prof_sq_valu" -- $B $P --opt mfma=0
run prof_sq2_mfma --pmc $SQ2 --kernel-trace --output-format csv -d "$O/prof_sq2_mfma" -- $B $P
run prof_sq_mfma_unpacked --pmc $SQ --kernel-trace --output-format csv -d "$O/prof_sq_mfma_unpacked" -- $B $P --opt mfma_pack=0
ls "$O"
