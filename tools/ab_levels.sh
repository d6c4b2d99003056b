run() { python bench.py --no-cpu-baseline --no-extra-legs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-75s q/s %.0f  step %.3f ms  scan %.3f ms x %d' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches']))" "$@"; }
for g in 4 8 16 32 64; do
run --opt self_tighten=0 --opt mfma_level_growth=$g
run --opt self_tighten=0 --opt mfma_level_growth=$g --rows 12500000
run --opt self_tighten=0 --opt mfma_level_growth=$g --queries 64
run --opt self_tighten=0 --opt mfma_level_growth=$g --k 100
done
