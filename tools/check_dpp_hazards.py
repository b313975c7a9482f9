#!/usr/bin/env python3
"""Developer tool: the two hazards of the hand-written v_fmac_f64_dpp instructions (fb_row16.h,
fmac_bc), which the compiler's hazard recognizer does not see inside inline assembly, checked on
the assembly of the record instances that use them (one row per QP):
  1. a VALU instruction that writes the DPP source register pair within the TWO wait states before
     the instruction (s_nop n counts n + 1; any other instruction counts 1);
  2. a VALU write of EXEC (v_cmpx*) within the FIVE wait states before it.
Exit code 1 if either is found.  usage: tools/check_dpp_hazards.py [rec_12_4_20 rec_12_4_32 ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
units = sys.argv[1:] or ["rec_12_4_20", "rec_12_4_32"]
bad = 0
for u in units:
    asm = f"/tmp/_hazard_{u}.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
                           "--cuda-device-only", "-S", "-o", asm, u + ".hip"], cwd=os.path.join(ROOT, "fbstab_amd", "csrc"),
                          stderr=subprocess.DEVNULL)
    body = []
    for l in open(asm):
        t = l.split(";")[0].strip()
        if not t or t.startswith((".", "//")) or t.endswith(":"):
            continue
        body.append(t)
    n = 0
    for i, t in enumerate(body):
        if not t.startswith("v_fmac_f64_dpp"):
            continue
        n += 1
        regs = re.findall(r"v\[(\d+):(\d+)\]", t)
        src = set(range(int(regs[1][0]), int(regs[1][1]) + 1))
        waits, j = 0, i - 1
        while j >= 0 and waits < 5:
            p = body[j]
            if p.startswith("s_nop"):
                waits += int(p.split()[1]) + 1
                j -= 1
                continue
            if re.match(r"v_cmpx", p):
                print(f"{u}: EXEC written by `{p}` {waits} wait states before `{t}`")
                bad += 1
            if waits < 2 and p.startswith("v_") and not p.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                m = re.match(r"v_\w+\s+(v\[(\d+):(\d+)\]|v(\d+))", p)
                if m:
                    dst = set(range(int(m.group(2)), int(m.group(3)) + 1)) if m.group(2) else {int(m.group(4))}
                    if dst & src:
                        print(f"{u}: DPP source written by `{p}` {waits} wait states before `{t}`")
                        bad += 1
            if p.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
                break  # (a block boundary: what precedes on other paths is not visible here)
            waits += 1
            j -= 1
    print(f"{u}: {n} v_fmac_f64_dpp instructions checked")
print("hazards found:" if bad else "no hazard found", bad if bad else "")
sys.exit(1 if bad else 0)
