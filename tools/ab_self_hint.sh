# same box: batches above spec_max_queries start their single self-tightening pass under the previous batch's k-th distance + 2
# (default) against the bootstrap sample (self_hint = 0)
run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-60s q/s %.0f  step %.3f ms  scan %.3f ms x %d  frac %.3f' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches'], r['frac']))" "$@"; }
ab() { run "$@"; run "$@" --opt self_hint=0; run "$@"; run "$@" --opt self_hint=0; }
ab; ab --queries 512; ab --queries 256; ab --k 100; ab --k 256; ab --nbytes 32 --metric nphd; ab --nbytes 16 --rows 10000000 --queries 512 --k 400; ab --rows 1000000
