run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-75s q/s %.0f  step %.3f ms  scan %.3f ms x %d' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches']))" "$@"; }
for b in 65536 32768 16384 8192; do
run --rows 12500000 --force-collective --opt self_boot_rows=$b
done
run --rows 12500000 --force-collective --no-profile 2>/dev/null || python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 --rows 12500000 --force-collective --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-profile 12.5M collective: step %.3f ms' % d['ms_per_step'])"
python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 --rows 125000000 --force-collective --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-profile 125M collective: step %.3f ms' % d['ms_per_step'])"
python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 --rows 1000000000 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-profile 1B one GPU: step %.3f ms' % d['ms_per_step'])"
