#!/usr/bin/env python3
"""Summarise -Rpass-analysis=kernel-resource-usage output for the scan kernel variants.

usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Rpass-analysis=kernel-resource-usage \
             -o /tmp/x.so iscc_search_amd/csrc/isccsearch.hip 2> res.txt ; python tools/kernel_resources.py res.txt
"""
import re
import sys

txt = open(sys.argv[1]).read()
failed = False
occ_key = "Occupancy \\[waves/SIMD\\]"
scr_key = "ScratchSize \\[bytes/lane\\]"
for b in re.split(r"Function Name: ", txt)[1:]:
    name = b.split()[0]
    m = re.match(r"_ZN3isk11scan_kernelILi(\d)ELb(\d)ELi(\d+)ELi(\d)ELb1E", name)
    if m:
        W, MASK, TQ, MODE = m.groups()
    else:
        m = re.match(r"_ZN3isk17scan_adapt_kernelILi(\d+)ELi(\d)E", name)   # 64-bit codes, both fast paths in one kernel
        if not m:
            continue
        W, MASK = "1*", "0"
        TQ, MODE = m.groups()

    def g(k):
        return re.search(k + r": (\d+)", b).group(1)

    bad = "" if (g("VGPRs Spill") == "0" and g(scr_key) == "0") else "   <-- FORBIDDEN with asm-issued loads (scratch / VGPR spill)"
    print("W=%s MASK=%s TQ=%-2s MODE=%s: SGPR=%-3s VGPR=%-3s waves/SIMD=%s sgpr_spill=%-3s vgpr_spill=%s scratch=%s%s" % (
        W, MASK, TQ, MODE, g("TotalSGPRs"), g("VGPRs"), g(occ_key), g("SGPRs Spill"), g("VGPRs Spill"), g(scr_key), bad))
    if bad:
        failed = True

sys.exit(1 if failed else 0)
