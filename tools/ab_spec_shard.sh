# does a range-limited pass under the previous step's k-th distance (+ 2) pay for LARGE batches?  (spec_max_queries = 1024 against the default 128)
run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-75s q/s %.0f  step %.3f ms  scan %.3f ms x %d' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches']))" "$@"; }
ab() { run "$@"; run "$@" --opt spec_max_queries=1024; run "$@"; run "$@" --opt spec_max_queries=1024; }
ab; ab --queries 512; ab --queries 256; ab --k 100; ab --k 1; ab --rows 12500000 --force-collective; ab --nbytes 32 --metric nphd; ab --nbytes 16
