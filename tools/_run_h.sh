cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for rows in 25000000 50000000; do for f in 0 100000000; do
python tools/bench_one.py $rows 16 16 10 60 2 mfma_few_rows=$f
python tools/bench_one.py $rows 32 12 10 60 2 mfma_few_rows=$f
done; done
python tools/bench_simprint.py --raw-only 2>&1 | grep "nq= 16"
