# same-box A/B of the ordered (inline asm) stage of mfma_scan_kernel for codes of 2..4 words: the committed build against -DISK_ORDERED_STAGE=0
run() { python bench.py --no-cpu-baseline --no-extra-legs --no-other-configs --settle-steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-60s q/s %.0f  step %.3f ms  scan %.3f ms x %d  frac %.3f' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches'], r['frac']))" "$@"; }
all() { run --nbytes 32 --metric nphd; run --nbytes 16; run --nbytes 24; run --nbytes 16 --rows 10000000 --queries 512 --k 400; }
echo "== ordered stage (committed build)"; all
cd iscc_search_amd/csrc && cp libisccsearch_hip.so /tmp/lib_ordered.so && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -ffinite-math-only -DISK_ORDERED_STAGE=0 -c -o /tmp/mfma_scan_unordered.o mfma_scan.hip && hipcc --offload-arch=gfx950 -shared -o libisccsearch_hip.so isccsearch.o /tmp/mfma_scan_unordered.o docfreq.o && cd ../..
echo "== builtin-scheduled stage (-DISK_ORDERED_STAGE=0)"; all
cp /tmp/lib_ordered.so iscc_search_amd/csrc/libisccsearch_hip.so
echo "== ordered stage again"; all
