run() { python bench.py --no-cpu-baseline --no-extra-legs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-60s q/s %.0f  step %.3f ms  scan %.3f ms' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))" "$@"; }
run
run --opt self_boot_rows=262144
run --opt self_refresh_steps=4
run --opt self_tighten=0
run --k 1
run --k 100
run --k 100 --opt self_tighten=0
run --rows 12500000
run --rows 12500000 --opt self_tighten=0
run --queries 64
run --queries 64 --opt self_tighten=0
