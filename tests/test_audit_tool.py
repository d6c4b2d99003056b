"""
tools/audit_kernels.py is part of the build (``make audit``): it must accept the pipelined load / counted-wait structure of
the scan kernels and reject the three ways that structure can silently break.  Synthetic gfx950 assembly, CPU tier.
"""

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "audit_kernels.py")

RES = """remark: Function Name: _ZN3isk11scan_kernelILi1ELb0ELi8ELi0ELb1EEEvNS_10ScanParamsE
remark:     TotalSGPRs: 96
remark:     VGPRs: 74
remark:     AGPRs: 0
remark:     ScratchSize [bytes/lane]: {scratch}
remark:     Occupancy [waves/SIMD]: 6
remark:     SGPRs Spill: 4
remark:     VGPRs Spill: {spill}
"""

ASM = """_ZN3isk11scan_kernelILi1ELb0ELi8ELi0ELb1EEEvNS_10ScanParamsE:
	s_load_dwordx2 s[0:1], s[4:5], 0x0
	;;#ASMSTART
	{nop}
	global_load_dwordx4 v[2:5], v1, s[0:1] offset:0 nt
	;;#ASMEND
.LBB0_1:
	;;#ASMSTART
	s_nop 4
	global_load_dwordx4 v[6:9], v1, s[2:3] offset:0 nt
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_xor_b32_e32 v10, v2, v20
	{extra}
	s_cmp_lt_u32 s8, s9
	s_cbranch_scc0 .LBB0_3
	;;#ASMSTART
	s_nop 4
	global_load_dwordx4 v[2:5], v1, s[0:1] offset:0 nt
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_xor_b32_e32 v11, v6, v20
	s_cmp_lt_u32 s8, s9
	s_cbranch_scc1 .LBB0_1
.LBB0_3:
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_endpgm
.Lfunc_end0:
"""


def run(tmp_path, nop="s_nop 4", extra="s_nop 0", scratch=0, spill=0):
    a, r = tmp_path / "k.s", tmp_path / "k.res"
    a.write_text(ASM.format(nop=nop, extra=extra))
    r.write_text(RES.format(scratch=scratch, spill=spill))
    p = subprocess.run([sys.executable, TOOL, str(a), str(r)], capture_output=True, text=True)
    return p.returncode, p.stdout


def test_accepts_the_pipelined_structure(tmp_path):
    rc, out = run(tmp_path)
    assert rc == 0 and "3 asm load groups in 1 kernels" in out, out


def test_rejects_a_copy_of_an_in_flight_destination(tmp_path):
    rc, out = run(tmp_path, extra="v_mov_b32_e32 v30, v7")          # v7 is being loaded while the first tile is scored
    assert rc == 1 and "touches a destination of the load group" in out, out
    rc, out = run(tmp_path, extra="v_mov_b32_e32 v30, v3")          # v3 was retired by the counted wait: fine
    assert rc == 0, out


def test_rejects_missing_wait_state_pad_scratch_and_spills(tmp_path):
    rc, out = run(tmp_path, nop="s_nop 0")
    assert rc == 1 and "does not open with s_nop 4" in out
    rc, out = run(tmp_path, scratch=16)
    assert rc == 1 and "scratch" in out
    rc, out = run(tmp_path, spill=2)
    assert rc == 1 and "VGPR spills" in out


# hipcc merges the loop's `break` exits and its back edge into ONE block steered by an SGPR mask; the address arithmetic of
# the loop-head group may then sit in registers that group's own loads are about to overwrite
ASM_MERGED = """_ZN3isk11scan_kernelILi1ELb0ELi8ELi0ELb1EEEvNS_10ScanParamsE:
	;;#ASMSTART
	s_nop 4
	global_load_dwordx4 v[2:5], v1, s[0:1] offset:0 nt
	;;#ASMEND
	s_branch .LBB0_5
.LBB0_4:
	{in_exit_block}
	s_and_b64 vcc, exec, s[0:1]
	s_cbranch_vccnz .LBB0_9
.LBB0_5:
	v_lshlrev_b64 v[6:7], 12, v[40:41]
	;;#ASMSTART
	s_nop 4
	global_load_dwordx4 v[6:9], v1, s[2:3] offset:0 nt
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_xor_b32_e32 v10, v2, v20
	s_cbranch_vccnz .LBB0_4
	;;#ASMSTART
	s_nop 4
	global_load_dwordx4 v[2:5], v1, s[0:1] offset:0 nt
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_xor_b32_e32 v11, v6, v20
	s_branch .LBB0_4
.LBB0_9:
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_endpgm
.Lfunc_end0:
"""


def run_merged(tmp_path, in_exit_block="s_nop 0"):
    a, r = tmp_path / "m.s", tmp_path / "m.res"
    a.write_text(ASM_MERGED.format(in_exit_block=in_exit_block))
    r.write_text(RES.format(scratch=0, spill=0))
    p = subprocess.run([sys.executable, TOOL, str(a), str(r)], capture_output=True, text=True)
    return p.returncode, p.stdout


def test_accepts_address_arithmetic_in_the_loop_head_groups_own_registers(tmp_path):
    rc, out = run_merged(tmp_path)
    assert rc == 0 and "3 asm load groups in 1 kernels" in out, out


def test_still_rejects_a_touch_in_the_merged_exit_block(tmp_path):
    # on the `break` path the loop-head group is in flight until .LBB0_9: reading v7 in the shared exit block is a real hazard
    rc, out = run_merged(tmp_path, in_exit_block="v_mov_b32_e32 v30, v7")
    assert rc == 1 and "touches a destination of the load group" in out, out


# ---- the packed matrix-core kernel: distances from an MFMA to the users of its result (invariants 4 and 5) -----------------

PACK_RES = RES + """remark: Function Name: _ZN3isk16mfma_pack_kernelILi0EEEvNS_10ScanParamsEj
remark:     TotalSGPRs: 60
remark:     VGPRs: 150
remark:     AGPRs: 0
remark:     ScratchSize [bytes/lane]: 0
remark:     Occupancy [waves/SIMD]: 3
remark:     SGPRs Spill: 0
remark:     VGPRs Spill: 0
"""

PACK_ASM = """_ZN3isk16mfma_pack_kernelILi0EEEvNS_10ScanParamsEj:
	v_mov_b32_e32 v100, 0
.LBB1_1:
	;;#ASMSTART
	v_mfma_f32_32x32x64_f8f6f4 v[18:33], v[90:93], v[98:101], v[2:17] cbsz:4 blgp:4
	v_pk_minimum3_f16 v130, v131, v50, v51
	v_pk_minimum3_f16 v132, v58, v59, v60
	{second}
	v_mfma_scale_f32_32x32x64_f8f6f4 v[18:33], v[82:85], v[98:101], v[18:33], v145, v146 op_sel_hi:[0,0,0] cbsz:4 blgp:4
	;;#ASMEND
	{pad}
	s_cmp_lt_u32 s8, s9
	s_cbranch_scc1 .LBB1_2
	v_add_u32_e32 v140, 1, v140
.LBB1_2:
	{reader}
	s_cmp_lt_u32 s10, s11
	s_cbranch_scc1 .LBB1_1
	s_endpgm
.Lfunc_end1:
"""


def run_pack(tmp_path, second="v_pk_minimum3_f16 v130, v130, v52, v53", pad="s_nop 7\n\ts_nop 3", reader="v_mov_b32_e32 v141, v18"):
    a, r = tmp_path / "k.s", tmp_path / "k.res"
    a.write_text(ASM.format(nop="s_nop 4", extra="s_nop 0") + PACK_ASM.format(second=second, pad=pad, reader=reader))
    r.write_text(PACK_RES.format(scratch=0, spill=0))
    p = subprocess.run([sys.executable, TOOL, str(a), str(r)], capture_output=True, text=True)
    return p.returncode, p.stdout


def test_accepts_results_read_far_enough_from_their_mfma(tmp_path):
    rc, out = run_pack(tmp_path)
    assert rc == 0 and "2 MFMAs in 1 matrix-core kernels" in out, out


def test_rejects_a_result_read_too_early_on_any_path(tmp_path):
    rc, out = run_pack(tmp_path, pad="s_nop 3")                       # 4 + 2 (+ 1 on the longer path) wait states: too few on both
    assert rc == 1 and "wait states after the MFMA that writes it" in out, out
    # the SHORT path (branch taken) decides: s_nop 7 + s_nop 0 + s_cmp + s_cbranch = 11 < 12, the fall-through has one more
    rc, out = run_pack(tmp_path, pad="s_nop 7\n\ts_nop 0")
    assert rc == 1 and "only 11 wait states" in out, out
    rc, out = run_pack(tmp_path, pad="s_nop 7\n\ts_nop 1")
    assert rc == 0, out
    # a reader of another register is no business of the MFMA
    rc, out = run_pack(tmp_path, pad="s_nop 7\n\ts_nop 1", reader="v_mov_b32_e32 v141, v50")
    assert rc == 0, out
    # ... but with a short pad the next trip's first MFMA overwrites v[18:33] while the scaled one is still in the pipe
    rc, out = run_pack(tmp_path, pad="s_nop 0", reader="v_mov_b32_e32 v141, v50")
    assert rc == 1 and "v_mfma_f32_32x32x64_f8f6f4 v[18:33]" in out and "uses v18" in out, out
    # across the loop's back edge: the first fold instruction of the next trip names v18 (s_cmp + s_cbranch + MFMA in between)
    rc, out = run_pack(tmp_path, second="v_pk_minimum3_f16 v130, v130, v18, v53", pad="s_nop 7\n\ts_nop 7", reader="s_nop 0")
    assert rc == 1 and "wait states after the MFMA that writes it" in out, out


def test_rejects_a_fold_instruction_that_directly_follows_its_producer(tmp_path):
    rc, out = run_pack(tmp_path, second="v_pk_minimum3_f16 v132, v132, v52, v53")
    assert rc == 1 and "follows the v_pk_minimum3_f16 that writes its operand directly" in out, out
