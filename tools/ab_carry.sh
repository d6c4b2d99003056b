# same-box A/B: the general loop of the packed scan carrying its accumulators across steps for even group counts (committed build)
# against -DISK_PACK_CARRY=0 (a step's first MFMAs and last fold stand alone)
run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-60s q/s %.0f  step %.3f ms  scan %.3f ms x %d  frac %.3f' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches'], r['frac']))" "$@"; }
all() { run; run --queries 512; run --queries 256; run --queries 192; run --rows 12500000 --force-collective; run --k 100; }
echo "== carried (committed build)"; all
cd iscc_search_amd/csrc && cp libisccsearch_hip.so /tmp/lib_committed.so && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -ffinite-math-only -DISK_PACK_CARRY=0 -c -o /tmp/mfma_scan_nc.o mfma_scan.hip 2>/dev/null && hipcc --offload-arch=gfx950 -shared -o libisccsearch_hip.so isccsearch.o /tmp/mfma_scan_nc.o docfreq.o && cd ../..
echo "== not carried (-DISK_PACK_CARRY=0)"; all
cp /tmp/lib_committed.so iscc_search_amd/csrc/libisccsearch_hip.so
echo "== carried again"; all
