cd $GRAFT_REPO_ROOT
for n in 2500 100000 1000000; do echo "== $n assets"; timeout -k 10 500 python tools/bench_protocol.py $n 2>&1 | grep -v "^$" | tail -5; done
