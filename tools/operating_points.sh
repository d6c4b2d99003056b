#!/bin/bash
# bench.py at the operating points DESIGN.md quotes (run through gpurun from the repo root; ~2 minutes):
#   bash tools/operating_points.sh > gpurun_out/r03_operating_points.txt
run() {
    python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline'] or {}
print('%-58s %10.0f q/s  %8.3f ms/step  %-5s frac %.3f  scan %8.3f ms x %-3d fallbacks %d' % (' '.join(sys.argv[1:]) or '(default: 100 M x 64-bit, 1 024 queries, k = 10)',
      d['value'], d['ms_per_step'], r.get('bound', '-'), r.get('frac', 0), r.get('avg_launch_ms', 0), r.get('launches', 0), d['fallback_queries']))" "$@"
}
echo "# bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 <args>: queries/s, ms per step, bound and fraction of its roofline, ms per scan launch x launches in 20 steps"
run
echo "# batch size (100 M x 64-bit, k = 10): <= 8 queries run the XOR + popcount kernel, more the packed matrix-core kernel; up to 128 queries one speculative range-limited pass (the ordinary path of the same sizes: --opt speculate=0 below)"
for q in 1 4 8 9 12 16 17 32 64 96 128 192 256 512; do run --queries $q; done
for q in 1 8 16 32 64 128; do run --queries $q --opt speculate=0; done
echo "# k"
for k in 1 100 256 512 1000 2000; do run --k $k; done
echo "# code length (Hamming tables of 128 / 192 / 256-bit codes) and config 3 (NPHD table of 256-bit units)"
run --nbytes 16; run --nbytes 24; run --nbytes 32; run --nbytes 32 --metric nphd
echo "# shards of the 100 M-row index with the collective enabled on ONE GPU (rehearsal, not a scaling result), and config 4's shard and index"
for r in 50000000 25000000 12500000; do run --rows $r --force-collective; done
run --rows 100000000 --force-collective
run --rows 125000000 --force-collective
run --rows 1000000000
echo "# config 2 (1 M x 64-bit: cache resident) and config 5's table shape (10 M x 128-bit, 512 queries, k = 400; 128-bit keys are in bench.py's other_configs)"
for q in 1 16 1024; do run --rows 1000000 --queries $q; done
run --nbytes 16 --rows 10000000 --queries 512 --k 400
run --nbytes 8 --rows 10000000 --queries 512 --k 400
run --nbytes 32 --rows 10000000 --queries 512 --k 400
