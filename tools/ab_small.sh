# small and middle batches over 100 M rows: step and scan time per batch size (one box, one library)
run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-60s q/s %.0f  step %.3f ms  scan %.3f ms x %d' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches']))" "$@"; }
for q in 17 32 64 96 128 192 256 1024; do run --queries $q; done
run --queries 32 --opt spec_max_queries=1024; run --queries 64 --opt spec_max_queries=1024; run --queries 128 --opt spec_max_queries=1024
run --queries 32 --k 100; run --rows 12500000 --force-collective
