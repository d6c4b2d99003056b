#!/usr/bin/env python3
"""
Build-time audit of the scan kernels (run by ``make -C iscc_search_amd/csrc audit`` and by ``__graft_entry__.build()``).

The XOR + popcount scan issues its streaming loads from inline asm (``load_tile_asm`` in csrc/kernels.hip.h) so that the
prefetch of the next tile stays in flight while the current one is scored.  hipcc neither counts nor orders what is inside
an asm statement (cdna_hip_programming.md section 5.7), so three invariants are the kernel author's and are checked here:

  1. no scratch and no VGPR spill in any kernel that uses those loads (a spill or copy of a load destination between the
     load and its counted wait would move garbage) -- from ``-Rpass-analysis=kernel-resource-usage``;
     the MFMA scan kernels are held to the same (their accumulators must stay in registers);
  2. between a tile's ``global_load_dwordx4`` group and the counted ``s_waitcnt vmcnt(N)`` that retires it -- the first asm
     wait after the NEXT load group, or any ``s_waitcnt vmcnt(0)`` -- no instruction on ANY control-flow path reads, writes or
     moves one of its destination VGPRs (checked on the gfx950 assembly, following branches and loop back-edges);
  3. every such load group opens with ``s_nop 4`` (the scalar bases may come straight from v_readfirstlane / v_readlane: a
     VALU-written SGPR needs 5 wait states before a VMEM instruction reads it).

The packed matrix-core scan (``mfma_pack_kernel`` in csrc/mfma_scan.hip) issues its MFMAs and the fold of their results
from inline asm in a fixed order; hipcc places no hazard nops for asm, so two more invariants are checked on the assembly:

  4. along EVERY control-flow path, an instruction that reads or writes a VGPR written by a ``v_mfma*`` comes at least
     MFMA_WAIT_STATES wait states after it (one per instruction, N + 1 for ``s_nop N`` -- the rule hipcc applies to its
     own code; an accumulating MFMA whose SrcC is its own destination is exempt);
  5. a ``v_pk_minimum3_f16`` is never followed directly by an instruction that names its destination.

usage: audit_kernels.py <device assembly .s> <kernel-resource-usage remarks .txt>
"""
import re
import sys

KERNEL_RE = re.compile(r"^(_ZN3isk\w+):\s*(;.*)?$")
LABEL_RE = re.compile(r"^(\.LBB\d+_\d+):")
VREG_RE = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
AUDITED = ("scan_kernel", "scan_adapt_kernel", "mfma_scan_kernel", "mfma_pack_kernel")
# an 8-pass MFMA (v_mfma_f32_32x32x64_f8f6f4 with FP4 operands) may be read by the VALU 11 wait states after it was issued
# (LLVM: passes + 3); one more for margin
MFMA_WAIT_STATES = 12


def vregs(text):
    out = set()
    for m in VREG_RE.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse_kernels(path):
    """{kernel name: [(text, in_asm)]} with labels kept as pseudo-instructions."""
    kernels, cur, in_asm = {}, None, False
    with open(path) as f:
        for raw in f:
            line = raw.rstrip("\n")
            m = KERNEL_RE.match(line)
            if m and not line.startswith("."):
                cur = kernels.setdefault(m.group(1), [])
                in_asm = False
                continue
            if cur is None:
                continue
            s = line.strip()
            if s.startswith(".Lfunc_end"):
                cur = None
                continue
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            lm = LABEL_RE.match(s)
            if lm:
                cur.append((lm.group(1) + ":", False))
                continue
            if not s or s.startswith(";") or s.startswith("."):
                continue
            cur.append((s.split(";")[0].strip(), in_asm))
    return kernels


def audit_loads(name, ins):
    """Invariants 2 and 3 for one kernel; returns a list of violation strings."""
    labels = {t[:-1]: i for i, (t, _) in enumerate(ins) if t.endswith(":")}
    # asm load groups: maximal runs of asm instructions containing global_load_dwordx4
    groups, i = [], 0
    while i < len(ins):
        if ins[i][1] and (ins[i][0].startswith("global_load_dwordx4") or ins[i][0].startswith("s_nop")):
            j, dests, has_load = i, set(), False
            while j < len(ins) and ins[j][1] and (ins[j][0].startswith("global_load_dwordx4") or ins[j][0].startswith("s_nop")):
                if ins[j][0].startswith("global_load_dwordx4"):
                    has_load = True
                    dests |= vregs(ins[j][0].split(",")[0])
                j += 1
            if has_load:
                groups.append((i, j, dests))
            i = max(j, i + 1)
        else:
            i += 1
    bad = []
    group_start = {g[0]: g for g in groups}
    for start, end, dests in groups:
        if not ins[start][0].startswith("s_nop 4"):
            bad.append(f"{name}: load group at #{start} does not open with s_nop 4")
        # Explore every path from the end of the group until the group is retired.  State: position, whether the next
        # tile's loads have been issued, and whether EXEC is known to be zero (the taken side of s_cbranch_execz, until
        # something writes EXEC: hipcc structurises the loop's `break` tests that way, and vector instructions executed
        # with EXEC = 0 neither read nor write anything -- v_readlane / v_readfirstlane excepted).
        # `younger`: asm loads issued since the group.  Loads return in order, so an asm `s_waitcnt vmcnt(N)` retires the group
        # once at least N younger loads have been issued (two buffers: the other tile's U*W loads and vmcnt(U*W); a ring of
        # DEPTH one-load slots: DEPTH - 1 younger loads and vmcnt(DEPTH - 1)).
        todo, seen = [(end, 0, False)], set()
        while todo:
            i, younger, exec0 = todo.pop()
            while i < len(ins):
                if (i, younger, exec0) in seen:
                    break
                seen.add((i, younger, exec0))
                text, in_asm = ins[i]
                if text.endswith(":"):
                    i += 1
                    continue
                if exec0 and re.match(r"s_\w+\s+exec\b|s_\w+saveexec", text):
                    exec0 = False
                if i in group_start:                       # the next tile's loads: must not touch ours either
                    g = group_start[i]
                    if g[0] == start and not exec0:
                        # Back at the group we started from without having met its retiring wait: every real path around the
                        # loop issues the OTHER buffer's loads and their counted wait first.  What gets here are paths that
                        # are impossible for their data (hipcc merges the `break` exits into one block steered by an SGPR
                        # mask); everything reachable within one iteration has been checked by now.
                        break
                    if not exec0:
                        for k in range(g[0], g[1]):
                            if vregs(ins[k][0]) & dests:
                                bad.append(f"{name}: load group at #{g[0]} touches in-flight destinations of the group at #{start}: {ins[k][0]}")
                    # (counted on the EXEC = 0 paths too: what they skip is the vector work, not the wave's instruction stream)
                    younger = min(younger + sum(1 for k in range(g[0], g[1]) if ins[k][0].startswith("global_load_dword")), 64)
                    i = g[1]
                    continue
                m = re.match(r"s_waitcnt\b.*vmcnt\((\d+)\)", text)
                if m and (int(m.group(1)) == 0 or (in_asm and younger and int(m.group(1)) <= younger)):
                    break                                   # retired on this path
                if text.startswith("s_endpgm"):
                    break
                vector = not text.startswith("s_")
                if vregs(text) & dests and (not exec0 or text.startswith(("v_readlane", "v_readfirstlane"))) and vector:
                    if i < start and not any(ins[j][0].endswith(":") or ins[j][0].startswith(("s_branch", "s_cbranch", "s_endpgm")) for j in range(i, start)):
                        # straight-line code that runs INTO the group we started from (its address arithmetic, in registers the
                        # group's own loads are about to overwrite): the same impossible return as in the case above
                        break
                    bad.append(f"{name}: `{text}` (#{i}) touches a destination of the load group at #{start} before its counted wait")
                    break
                bm = re.match(r"(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)", text)
                if bm:
                    target = labels.get(bm.group(2))
                    if target is None:
                        bad.append(f"{name}: branch to unknown label {bm.group(2)}")
                        break
                    if bm.group(1) == "s_branch":
                        i = target
                        continue
                    todo.append((target, younger, True if bm.group(1) == "s_cbranch_execz" else exec0))
                    if bm.group(1) == "s_cbranch_execnz":
                        exec0 = True                        # falling through an execnz branch: EXEC is zero
                i += 1
    return bad, len(groups)


def audit_mfma_distances(name, ins):
    """Invariants 4 and 5 for one kernel; returns (violations, number of MFMAs seen)."""
    nodes = [(i, t) for i, (t, _) in enumerate(ins) if not t.endswith(":")]
    index_of = {}                                   # position in `ins` -> node number of the next real instruction
    nxt = len(nodes)
    for n in range(len(nodes) - 1, -1, -1):
        index_of[nodes[n][0]] = n
    k = len(nodes)
    for i in range(len(ins) - 1, -1, -1):
        if i in index_of:
            k = index_of[i]
        else:
            index_of[i] = k
    labels = {t[:-1]: index_of[i] for i, (t, _) in enumerate(ins) if t.endswith(":")}
    bad, n_mfma = [], 0
    succ = []
    for n, (_, text) in enumerate(nodes):
        out = []
        bm = re.match(r"(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)", text)
        if bm:
            if bm.group(2) in labels:
                out.append(labels[bm.group(2)])
            if bm.group(1) != "s_branch" and n + 1 < len(nodes):
                out.append(n + 1)
        elif not text.startswith("s_endpgm") and n + 1 < len(nodes):
            out.append(n + 1)
        succ.append(out)

    def operands(text):
        parts = text.split(None, 1)
        return [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []

    # forward dataflow: state = {vgpr: wait states since the MFMA that wrote it}, entries dropped once safe; meet = minimum
    state = [None] * len(nodes)
    state[0] = {}
    work = [0]
    reported = set()
    while work:
        n = work.pop()
        cur = dict(state[n])
        text = nodes[n][1]
        op = text.split()[0]
        if op.startswith("v_mfma"):
            ops = operands(text)
            dst = vregs(ops[0])
            srcc = vregs(ops[3]) if len(ops) > 3 else set()
            touched = set()
            for o in ops[1:3] + ops[4:]:
                touched |= vregs(o.split()[0]) if o else set()
            if srcc != dst:
                touched |= srcc
            hot = sorted(r for r in touched | (dst if srcc != dst else set()) if r in cur)
            if hot and n not in reported:
                reported.add(n)
                bad.append(f"{name}: `{text}` (#{nodes[n][0]}) uses v{hot[0]} {cur[hot[0]]} wait states after the MFMA that writes it")
            n_mfma += 1 if n not in reported else 0
            step = 1
            cur = {r: w + step for r, w in cur.items() if w + step < MFMA_WAIT_STATES}
            for r in dst:
                cur[r] = 0
        else:
            hot = sorted(r for r in vregs(text) if r in cur) if not op.startswith("s_") else []
            if hot and n not in reported:
                reported.add(n)
                bad.append(f"{name}: `{text}` (#{nodes[n][0]}) touches v{hot[0]} only {cur[hot[0]]} wait states after the MFMA that writes it (need {MFMA_WAIT_STATES})")
            m = re.match(r"s_nop\s+(\d+)", text)
            step = int(m.group(1)) + 1 if m else 1
            cur = {r: w + step for r, w in cur.items() if w + step < MFMA_WAIT_STATES}
        for t in succ[n]:
            old = state[t]
            if old is None:
                state[t] = dict(cur)
                work.append(t)
            else:
                merged = dict(old)
                changed = False
                for r, w in cur.items():
                    if r not in merged or w < merged[r]:
                        merged[r] = w
                        changed = True
                if changed:
                    state[t] = merged
                    work.append(t)
    n_mfma = sum(1 for _, t in nodes if t.startswith("v_mfma"))
    for n, (_, text) in enumerate(nodes[:-1]):
        if text.startswith("v_pk_minimum3_f16"):
            dst = vregs(operands(text)[0])
            if dst & vregs(nodes[n + 1][1]) and n + 1 in succ[n]:
                bad.append(f"{name}: `{nodes[n + 1][1]}` (#{nodes[n + 1][0]}) follows the v_pk_minimum3_f16 that writes its operand directly")
    return bad, n_mfma


def audit_resources(path):
    txt = open(path).read()
    rows, bad = [], []
    for block in re.split(r"Function Name: ", txt)[1:]:
        name = block.split()[0]
        if not any(k in name for k in AUDITED):
            continue

        def g(key):
            return int(re.search(key + r": (\d+)", block).group(1))

        row = dict(name=name, sgpr=g("TotalSGPRs"), vgpr=g("VGPRs"), agpr=g("AGPRs"), occ=g(r"Occupancy \[waves/SIMD\]"),
                   sgpr_spill=g("SGPRs Spill"), vgpr_spill=g("VGPRs Spill"), scratch=g(r"ScratchSize \[bytes/lane\]"))
        rows.append(row)
        if row["vgpr_spill"] or row["scratch"]:
            bad.append(f"{name}: {row['vgpr_spill']} VGPR spills, {row['scratch']} bytes/lane of scratch -- forbidden in the scan kernels")
    return rows, bad


def main():
    asm, res = sys.argv[1], sys.argv[2]
    rows, bad = audit_resources(res)
    kernels = parse_kernels(asm)
    n_groups = 0
    n_kernels = 0
    for name, ins in kernels.items():
        if "scan_kernel" in name or "scan_adapt_kernel" in name or "mfma_pack_kernel" in name:
            b, n = audit_loads(name, ins)
            bad += b
            if n:
                n_kernels += 1
                n_groups += n
    n_pack = n_mfma = 0
    for name, ins in kernels.items():
        if "mfma_pack_kernel" in name or "mfma_scan_kernel" in name:     # every matrix-core kernel: their stages are inline asm
            b, n = audit_mfma_distances(name, ins)
            bad += b
            n_pack += 1
            n_mfma += n
    if "-v" in sys.argv:
        for r in rows:
            print("%-70s SGPR=%-3d VGPR=%-3d AGPR=%-3d waves/SIMD=%d sgpr_spill=%-3d vgpr_spill=%d scratch=%d" % (
                r["name"][:70], r["sgpr"], r["vgpr"], r["agpr"], r["occ"], r["sgpr_spill"], r["vgpr_spill"], r["scratch"]))
    spills = [r["sgpr_spill"] for r in rows]
    print(f"audit: {len(rows)} scan kernels, 0 scratch / 0 VGPR spills required; SGPR spills {min(spills) if spills else 0}..{max(spills) if spills else 0} (to VGPR lanes, allowed); "
          f"{n_groups} asm load groups in {n_kernels} kernels checked against their counted waits; "
          f"{n_mfma} MFMAs in {n_pack} matrix-core kernels checked for {MFMA_WAIT_STATES} wait states to every use of their results")
    if any("mfma_scan_kernel" in name for name in kernels) and not any("mfma_pack_kernel" in name for name in kernels):
        bad.append("mfma_scan_kernel is in the build but mfma_pack_kernel is not: kernel names changed; update tools/audit_kernels.py")
    if n_kernels == 0 or not rows:
        bad.append("nothing was audited: kernel names or the asm-load pattern changed; update tools/audit_kernels.py")
    for b in bad[:40]:
        print("AUDIT FAILURE:", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
