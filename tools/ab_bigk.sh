run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 10 --steps 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-70s q/s %.0f  step %.3f ms  scan %.3f ms x %d' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches']))" "$@"; }
for k in 256 400 512; do
run --k $k
run --k $k --opt self_tighten=0
run --k $k --opt candidate_cap=65536
run --k $k --opt self_boot_rows=262144
done
run --nbytes 16 --rows 10000000 --queries 512 --k 400
run --nbytes 16 --rows 10000000 --queries 512 --k 400 --opt self_tighten=0
run --nbytes 8 --rows 10000000 --queries 512 --k 400
run --nbytes 8 --rows 10000000 --queries 512 --k 400 --opt self_tighten=0
