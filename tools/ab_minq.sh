# 5..16 queries: the XOR + popcount kernel (default, mfma_min_queries = 17) against the packed matrix-core kernel (mfma_min_queries = 1), same box
run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-60s q/s %.0f  step %.3f ms  scan %.3f ms x %d' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches']))" "$@"; }
for q in 1 4 6 8 9 12 16; do run --queries $q; run --queries $q --opt mfma_min_queries=1; done
for q in 8 16; do run --queries $q --k 100; run --queries $q --k 100 --opt mfma_min_queries=1; done
for q in 8 16; do run --queries $q --nbytes 16; run --queries $q --nbytes 16 --opt mfma_min_queries=1;  run --queries $q --nbytes 32; run --queries $q --nbytes 32 --opt mfma_min_queries=1; done
