# config 5's table shape (10 M x 128-bit, 512 queries, k = 400): rows of the bootstrap sample per wanted neighbour
run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-90s q/s %.0f  step %.3f ms  scan %.3f ms x %d  retries %s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches'], d.get('self_retries')))" "$@"; }
for b in 1024 512 256 128 64; do run --nbytes 16 --rows 10000000 --queries 512 --k 400 --opt self_boot_per_k=$b; done
for b in 1024 256 128; do run --nbytes 8 --rows 10000000 --queries 512 --k 400 --opt self_boot_per_k=$b; run --nbytes 32 --rows 10000000 --queries 512 --k 400 --opt self_boot_per_k=$b; done
for b in 1024 256 128; do run --k 256 --opt self_boot_per_k=$b; run --k 100 --opt self_boot_per_k=$b; done
