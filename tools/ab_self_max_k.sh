# k above 512: the level design (default self_max_k = 512) against the single self-tightening pass (--opt self_max_k=N), same box
run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-75s q/s %.0f  step %.3f ms  scan %.3f ms x %d  fallbacks %d' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches'], d['fallback_queries']))" "$@"; }
run --k 700; run --k 700 --opt self_max_k=4096
run --k 1000; run --k 1000 --opt self_max_k=4096
run --k 2000; run --k 2000 --opt self_max_k=4096
run --k 4096 --queries 256; run --k 4096 --queries 256 --opt self_max_k=4096
run --nbytes 16 --rows 10000000 --queries 512 --k 1000; run --nbytes 16 --rows 10000000 --queries 512 --k 1000 --opt self_max_k=4096
run --nbytes 32 --metric nphd --k 1000; run --nbytes 32 --metric nphd --k 1000 --opt self_max_k=4096
