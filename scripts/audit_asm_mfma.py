#!/usr/bin/env python3
"""Build-time audit of the kernels whose MFMAs are written as inline asm (fa_device.h MfmaAsm: O^T / Q^T pinned in AGPRs).

hipcc pads no hazard for an asm statement, so for every `v_mfma_*` between ;;#ASMSTART / ;;#ASMEND of the generated ISA this
script checks, on the straight-line code of its basic block:
  (1) no VALU instruction writes one of its source registers within the two wait states in front of it
      (VALU write -> MFMA operand: cdna_hip_programming.md 5.6 item 2; the statement's own leading `s_nop 1` counts);
  (2) nothing but an MFMA that takes the result whole as its C operand reads or writes its destination registers within the
      12 wait states behind an 8-pass result (4-pass: 8) - instructions count one state each, `s_nop N` counts N + 1;
  (3) scratch accesses anywhere are findings; compiler v_accvgpr_* copies inside a block of >= 30 MFMAs are REPORTED per block
      (a ragged step may carry a few; the four steady steps of the loop must carry none - read the report).
  (4) M0: the LDS-DMA statements of these kernels write M0 without restoring it (fa_device.h lds_dma_b128_m0) - legal only while nothing
      else in the kernel reads M0, so any `s_movrel*` / `v_movrel*` / `s_set_gpr_idx*` / `s_sendmsg` / `ds_*_gs*` in it is a finding.
usage:  python scripts/audit_asm_mfma.py flash-attention-ipu_amd/build/asm/fa_fwd_pipe-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-name-substring]
exit code 1 on a finding.  (`make -C flash-attention-ipu_amd asm SRC=fa_fwd_pipe` writes the .s)
"""
import re
import sys


def regs(tok):
    """'v[4:19]' -> {('v', 4..19)}, 'a12' -> {('a', 12)}, anything else -> empty"""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


def operands(line):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else "", []
    return parts[0], [t.strip() for t in parts[1].split(",")]


def states(op, ops):
    if op == "s_nop":
        return int(ops[0]) + 1
    return 1


def is_valu(op):
    return op.startswith("v_") and not op.startswith("v_mfma") and not op.startswith("v_smfmac")


def audit(path, want=""):
    text = open(path).read()
    findings = 0
    for km in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = km.group(1), km.group(2)
        if want not in name or "v_mfma" not in body:
            continue
        lines = body.split("\n")
        # basic blocks of (op, operands, in_asm)
        blocks, cur, in_asm = [], [], False
        for ln in lines:
            if re.match(r"^\.LBB\d+_\d+:", ln) or re.match(r"^; %bb\.", ln):
                if cur:
                    blocks.append(cur)
                cur = []
                continue
            s = ln.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith((".", ";")):
                continue
            op, ops = operands(s)
            cur.append((op, ops, in_asm))
            if op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                blocks.append(cur)
                cur = []
        if cur:
            blocks.append(cur)
        n_asm = 0
        for blk in blocks:
            for op, ops, ia in blk:
                if op.startswith(("s_movrel", "v_movrel", "s_set_gpr_idx", "s_sendmsg")) or "_gs_" in op:
                    print(f"{name}: `{op} {', '.join(ops)}` reads M0, which the kernel's LDS-DMA statements clobber")
                    findings += 1
        for bi, blk in enumerate(blocks):
            n_mfma = sum(1 for op, _, _ in blk if op.startswith("v_mfma"))
            for op, ops, ia in blk:
                if op.startswith("scratch_"):
                    print(f"{name}: scratch access `{op} {', '.join(ops)}`")
                    findings += 1
            if n_mfma >= 30:
                n_acc = sum(1 for op, _, ia in blk if op.startswith("v_accvgpr") and not ia)
                if n_acc:
                    print(f"{name}: note: a block of {n_mfma} MFMAs / {len(blk)} instructions carries {n_acc} compiler v_accvgpr_* copies")
            for i, (op, ops, ia) in enumerate(blk):
                if not (ia and op.startswith("v_mfma")):
                    continue
                n_asm += 1
                dst = regs(ops[0])
                srcs = set().union(*(regs(t) for t in ops[1:4]))
                passes = 8 if "32x32" in op else 4
                # (1) VALU write to a source within 2 states in front
                st, j = 0, i - 1
                while j >= 0 and st < 2:
                    pop, pops, _ = blk[j]
                    if is_valu(pop) and pops and (regs(pops[0]) & srcs):
                        print(f"{name}: VALU write `{pop} {', '.join(pops)}` {st} states in front of asm `{op} {', '.join(ops)}`")
                        findings += 1
                    st += states(pop, pops)
                    j -= 1
                # (2) result touched within the settle window.  Time is counted in wait states since the MFMA's issue: one per instruction,
                # N + 1 for `s_nop N`, and a following MFMA cannot issue before the matrix pipe has taken the previous one's passes
                # (back-to-back MFMAs issue `passes` states apart).  The scan runs on into the blocks that follow in program order
                # (fall-through; a loop's back edge re-enters a block already covered from its own start).
                need = 12 if passes == 8 else 8
                t, j = 0, i + 1
                last_mfma_t, last_mfma_passes = 0, passes
                stream = blk[i + 1:] + [x for nb in blocks[bi + 1:bi + 4] for x in nb]
                for nop, nops, _ in stream:
                    if t >= need:
                        break
                    if nop.startswith(("v_mfma", "v_smfmac")):
                        t = max(t + 1, last_mfma_t + last_mfma_passes)
                        last_mfma_t, last_mfma_passes = t, (8 if "32x32" in nop else 4)
                        if t >= need:
                            break
                        touched = set().union(*(regs(x) for x in nops[:4])) & dst
                        whole_c = len(nops) >= 4 and regs(nops[3]) == dst and regs(nops[0]) == dst
                        a_b = set().union(*(regs(x) for x in nops[1:3])) & dst
                        if touched and (not whole_c or a_b):
                            print(f"{name}: MFMA `{nop} {', '.join(nops)}` touches the result of asm `{op} {', '.join(ops)}` after {t} states")
                            findings += 1
                        continue
                    t += states(nop, nops)
                    if t >= need:
                        break
                    if not nop.startswith("s_"):
                        touched = set().union(*(regs(x) for x in nops)) & dst if nops else set()
                        if touched:
                            print(f"{name}: `{nop} {', '.join(nops)}` touches the result of asm `{op} {', '.join(ops)}` after {t} states (need {need})")
                            findings += 1
        if n_asm:
            print(f"{name}: {n_asm} asm MFMAs audited")
    return findings


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    f = audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    print(f"audit: {f} finding(s)")
    sys.exit(1 if f else 0)
