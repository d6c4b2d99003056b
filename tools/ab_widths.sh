run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-60s q/s %.0f  step %.3f ms  scan %.3f ms x %d  frac %.3f' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches'], r['frac']))" "$@"; }
run --nbytes 32 --metric nphd
run --nbytes 16
run --nbytes 24
run --nbytes 32
run --nbytes 16 --rows 10000000 --queries 512 --k 400
