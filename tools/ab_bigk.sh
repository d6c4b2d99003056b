run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 10 --steps 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-70s q/s %.0f  step %.3f ms  scan %.3f ms x %d' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches']))" "$@"; }
for k in 10 100 256 400 512; do
run --k $k
done
run --k 100 --opt self_boot_rows=65536
run --k 512 --opt self_boot_rows=262144
run --nbytes 16 --rows 10000000 --queries 512 --k 400
run --nbytes 8 --rows 10000000 --queries 512 --k 400
run --rows 12500000 --k 100
