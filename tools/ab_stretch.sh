# same-box A/B: a wave's DEPTH steps in flight as ONE stretch of consecutive rows (committed build) against DEPTH steps a grid stride apart (-DISK_PACK_STRETCH=0)
run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-60s q/s %.0f  step %.3f ms  scan %.3f ms x %d' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches']))" "$@"; }
all() { for q in 32 64 128 192; do run --queries $q; done; }
echo "== stretches (committed build)"; all
cd iscc_search_amd/csrc && cp libisccsearch_hip.so /tmp/lib_committed.so && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -ffinite-math-only -DISK_PACK_STRETCH=0 -c -o /tmp/mfma_scan_s0.o mfma_scan.hip 2>/dev/null && hipcc --offload-arch=gfx950 -shared -o libisccsearch_hip.so isccsearch.o /tmp/mfma_scan_s0.o docfreq.o && cd ../..
echo "== a grid stride apart (-DISK_PACK_STRETCH=0)"; all
cp /tmp/lib_committed.so iscc_search_amd/csrc/libisccsearch_hip.so
echo "== stretches again"; all
