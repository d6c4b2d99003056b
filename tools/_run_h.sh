set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
python -m pytest tests/test_gpu_pack.py tests/test_gpu_mfma.py tests/test_gpu_many.py -x -q > gpurun_out/r04j/tests.txt 2>&1 || { tail -30 gpurun_out/r04j/tests.txt; exit 1; }
tail -2 gpurun_out/r04j/tests.txt
B="python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20"
for q in 9 16 24 32 64; do
  $B --queries $q > gpurun_out/r04j/q_$q.json 2> gpurun_out/r04j/q_$q.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/r04j/q_$q.json').read().strip().splitlines()[-1]); print($q, round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"
done
