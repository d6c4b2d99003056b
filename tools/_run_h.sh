cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for r in 1000 10000 16384; do python tools/probe_host_overhead.py $r 1 2>&1 | tail -1; done
python tools/probe_host_overhead.py 10000 4 2>&1 | tail -1
python tools/probe_host_overhead.py 10000 64 2>&1 | tail -1
python tools/bench_protocol.py 2>&1 | tail -4
