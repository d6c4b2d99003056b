#!/bin/bash
# Accounting of one step of mfma_pack_kernel BY SUBTRACTION: the product kernel and builds of it with one
# part of the step left out (ISK_EXP_PACK, mfma_scan.hip), each timed (rocprofv3 --kernel-trace --stats) and counted (one --pmc
# pass: matrix pipe busy, clock, vector instructions) on the default step: 100 M x 64-bit rows x 1 024 queries, k = 10.
# Run through gpurun from the repo root:  bash tools/pack_step_accounting.sh   ->  gpurun_out/pack_acct/summary.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/pack_acct
mkdir -p "$O"
C=$ROOT/iscc_search_amd/csrc
cp $C/libisccsearch_hip.so /tmp/lib_product.so
SQ="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
for v in ${VARIANTS:-0 1 2 4 3 7}; do
    (cd $C && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -ffinite-math-only -DISK_EXP_PACK=$v -c -o /tmp/mfma_scan_exp.o mfma_scan.hip 2>/dev/null \
        && hipcc --offload-arch=gfx950 -shared -o libisccsearch_hip.so isccsearch.o /tmp/mfma_scan_exp.o docfreq.o simprint_score.o) || { echo "variant $v did not build"; continue; }
    cd /tmp && export TMPDIR=/tmp
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$v" -o t -- python3 $ROOT/tools/bench_one.py 100000000 8 1024 10 6 > "$O/stats_$v.log" 2>&1 || echo "variant $v: stats run failed" 
    timeout -k 10 240 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d "$O/pmc_$v" -o t -- python3 $ROOT/tools/bench_one.py 100000000 8 1024 10 3 > "$O/pmc_$v.log" 2>&1 || echo "variant $v: counter run failed"
    echo "variant $v done: $(grep -h 'ms/call' $O/stats_$v.log | tail -1)"
    cd $ROOT
done
cp /tmp/lib_product.so $C/libisccsearch_hip.so
python3 $ROOT/tools/pack_step_accounting.py "$O" | tee "$O/summary.txt"
