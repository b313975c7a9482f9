#!/usr/bin/env python3
"""Build check: the hazards of the hand-written v_fmac_f64_dpp instructions (fb_row16.h, fmac_bc), which
the compiler's hazard recognizer does not see inside inline assembly, verified on the DISASSEMBLY OF THE
BUILT OBJECTS (the code objects inside fbstab_amd/csrc/build/<lib>/rec_*.o or a shared library) - every
path into every such instruction, through labels, loop back-edges and fall-throughs:
  1. no VALU instruction writes the DPP source register pair within the TWO wait states before it
     (s_nop n counts n + 1, any other instruction 1; v_swap / v_permlane*_swap write both operands);
  2. no VALU write of EXEC (v_cmpx*) within the FIVE wait states before it;
  3. no transcendental instruction (v_rsq / v_rcp / v_sqrt / v_exp / v_log / v_sin / v_cos) writes ANY of
     its register operands in the ONE wait state before it (result forwarding of the trans unit);
  4. a path that leaves the function (entry, a call's return point, code that only an indirect jump
     can reach) before the wait states are accounted for is reported as unverifiable.
The fused instructions carry NO s_nop of their own (fb_row16.h: FB_FMAC_GUARD_NOP=0) - this check is what
stands between the build and a stale operand, so `make` runs it and fails on a finding.
An object in which no amdgcn code object is found is a finding (nothing was checked); with --expect-nonzero
(the build) so is an object without a single fused instruction.
usage: tools/check_dpp_hazards.py [--expect-nonzero] [objects or libraries ...]   (default: the product build's rec_*.o)"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = ("v_rsq_", "v_rcp_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
NOT_VGPR_WRITERS = ("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")
BOTH = ("v_swap_", "v_permlane16_swap", "v_permlane32_swap")


def regs_of(tok):
    m = re.fullmatch(r"[-|]*v\[(\d+):(\d+)\][|]*", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"[-|]*v(\d+)[|]*", tok)
    return {int(m.group(1))} if m else set()


def operands(text):
    parts = text.split(None, 1)
    return [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []


def vgpr_writes(text):
    op = text.split()[0]
    if not op.startswith("v_") or op.startswith(NOT_VGPR_WRITERS):
        return set()
    ops = operands(text)
    if not ops:
        return set()
    w = regs_of(ops[0].split()[0])
    if op.startswith(BOTH) and len(ops) > 1:
        w |= regs_of(ops[1].split()[0])
    return w


def disassemble(path, tmp):
    """Code objects of `path` (an offload bundle: .o / .so) -> list of disassembly texts."""
    local = os.path.join(tmp, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.check_call([OBJDUMP, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = []
    for co in sorted(glob.glob(local + ".*amdgcn*")):
        out.append(subprocess.check_output([OBJDUMP, "-d", "--symbolize-operands", "--no-show-raw-insn", co], text=True))
    return out


def functions(dis):
    """-> [(name, [(label or None, text)])], one entry per function symbol."""
    fns, cur, pending = [], None, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if re.fullmatch(r"L\d+", m.group(1)):
                pending = m.group(1)
            else:
                cur = []
                fns.append((m.group(1), cur))
                pending = None
            continue
        if cur is None or not line.startswith("\t"):
            continue
        text = line.split("//")[0].strip()
        if not text:
            continue
        cur.append((pending, text))
        pending = None
    return fns


def check_function(name, ins, unit):
    label_at = {lab: i for i, (lab, _) in enumerate(ins) if lab}
    jumps = {}
    for i, (_, t) in enumerate(ins):
        if t.startswith(("s_cbranch", "s_branch")):
            jumps.setdefault(t.split()[-1], []).append(i)
    findings, n = [], 0
    for i, (_, t) in enumerate(ins):
        if not t.startswith("v_fmac_f64_dpp"):
            continue
        n += 1
        ops = operands(t)
        src = regs_of(ops[1].split()[0])
        allr = set().union(*(regs_of(o.split()[0]) for o in ops[:3]))
        seen = set()

        def walk(j, waits, via_jump=False):
            """instruction j is the next one BEFORE the point reached with `waits` wait states behind it;
            via_jump: j is the branch instruction this path was TAKEN from (an unconditional s_branch is
            then one more instruction on the path - one wait state - not the end of it)"""
            while waits < 5:
                if j < 0:
                    if waits < 2:
                        findings.append(f"{unit}: {name}: function entry {waits} wait states before `{t}` (unverifiable)")
                    return
                if (j, waits, via_jump) in seen:
                    return
                seen.add((j, waits, via_jump))
                lab, p = ins[j]
                op = p.split()[0]
                taken, via_jump = via_jump, False
                if op in ("s_endpgm",) or op.startswith("s_setpc") or (op == "s_branch" and not taken):
                    # nothing falls through these.  What follows them is reached through its label (followed
                    # above) - or, WITHOUT a label, only by an indirect jump this tool cannot follow
                    if j + 1 < len(ins) and not ins[j + 1][0] and waits < 2:
                        findings.append(f"{unit}: {name}: code behind `{p}` without a label {waits} wait states before `{t}` (unverifiable)")
                    return
                if op.startswith(("s_swappc", "s_call")):
                    if waits < 2:
                        findings.append(f"{unit}: {name}: a call returns {waits} wait states before `{t}` (unverifiable)")
                    return
                if op == "s_nop":
                    w = int(p.split()[1]) + 1
                else:
                    w = 1
                    if op.startswith("v_cmpx"):
                        findings.append(f"{unit}: {name}: EXEC written by `{p}` {waits} wait states before `{t}`")
                    wr = vgpr_writes(p)
                    if waits < 2 and wr & src:
                        findings.append(f"{unit}: {name}: DPP source written by `{p}` {waits} wait states before `{t}`")
                    if waits < 1 and op.startswith(TRANS) and wr & allr:
                        findings.append(f"{unit}: {name}: operand written by `{p}` (trans) directly before `{t}`")
                waits += w
                if lab:  # other ways into this instruction
                    for b in jumps.get(lab, []):
                        walk(b, waits, True)
                j -= 1

        # the instruction's own label: paths that jump straight to it
        if ins[i][0]:
            for b in jumps.get(ins[i][0], []):
                walk(b, 0, True)
        walk(i - 1, 0)
    return n, findings


def main():
    args = sys.argv[1:]
    expect_nonzero = "--expect-nonzero" in args  # the build: an object without a fused instruction is a finding
    args = [a for a in args if a != "--expect-nonzero"]
    paths = args or sorted(glob.glob(os.path.join(ROOT, "fbstab_amd", "csrc", "build", "libfbstab_hip", "rec_*.o")))
    if not paths:
        print("no objects to check")
        return 1
    total, bad = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for pth in paths:
            unit = os.path.basename(pth)
            n_unit = 0
            texts = disassemble(pth, tmp)
            if not texts:
                bad.append(f"{unit}: no amdgcn code object found in it (nothing was checked)")
            for dis in texts:
                for name, ins in functions(dis):
                    n, f = check_function(name[:60], ins, unit)
                    n_unit += n
                    bad += f
            total += n_unit
            print(f"{unit}: {n_unit} v_fmac_f64_dpp instructions checked")
            if expect_nonzero and n_unit == 0:
                bad.append(f"{unit}: no v_fmac_f64_dpp instruction found (--expect-nonzero)")
    for f in sorted(set(bad)):
        print(f)
    print(f"{total} instructions; " + (f"{len(set(bad))} findings" if bad else "no hazard found"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
