#!/usr/bin/env python3
"""Developer tool: the record kernel's Newton step as the ISA has it - instructions per
phase and per class.  Compiles fbstab_amd/csrc/rec_12_4_20.hip to
assembly with -DFB_PHASE_MARKERS (fb_common.h: the FB_STAMP_LAP / FB_PHASE points become
comments in the instruction stream; no instruction is added), takes the batch kernel
(EXACT, not KEEP, not the probe) and counts, between consecutive markers, the instructions
of each class.  The kernel holds the Newton step twice (newton_step_t<ROW>): the copy
WITHOUT the inv(Pi) image is the row form of the costate step, the one the BASELINE
workload runs.  argv: [--asm file.s]  (reuse an assembly file instead of compiling)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN12_GLOBAL__N_121fbstab_mpc_r16_kernelILi12ELi4ELi20ELb0ELb1ELb0ELi1EEE"

CLASSES = [
    ("fma_f64", r"^v_(fma|fmac|mul|add)_f64"),
    ("dpp_move", r"^v_mov_b(64|32)_dpp"),
    ("dpp_other", r"_dpp$|_dpp "),
    ("lane_xchg", r"^v_(readlane|writelane|readfirstlane|permlane)"),
    ("sqrt_div_seed", r"^v_(rsq|rcp|sqrt|div_|frexp|ldexp|trig)"),
    ("cmp_select", r"^v_(cmp|cmpx|cndmask)"),
    ("minmax_f64", r"^v_(max|min)_f64"),
    ("int_addr", r"^v_(add|sub|mul|mad|lshl|lshr|ashr|and|or|xor|bfe|bfi|not|lshlrev|lshrrev|ashrrev|add3|lshl_add|mul_lo|mul_hi|mad_u64|subrev|addc|subb)_(u|i|co|nc|b)\w*"),
    ("move_copy", r"^v_(mov_b32|mov_b64|accvgpr|swap|pk_mov)"),
    ("cvt_misc_valu", r"^v_"),
    ("lds", r"^ds_"),
    ("vmem", r"^(global|flat|buffer)_"),
    ("scratch", r"^scratch_"),
    ("waitcnt_nop", r"^s_(waitcnt|nop|sleep|barrier)"),
    ("salu", r"^s_"),
]
VALU = {"fma_f64", "dpp_move", "dpp_other", "lane_xchg", "sqrt_div_seed", "cmp_select", "minmax_f64", "int_addr",
        "move_copy", "cvt_misc_valu"}


def classify(mn, line):
    for name, rx in CLASSES:
        if re.search(rx, mn if name != "dpp_other" else line):
            return name
    return "other"


def main():
    asm = None
    if "--asm" in sys.argv:
        asm = sys.argv[sys.argv.index("--asm") + 1]
    else:
        asm = "/tmp/_rec_12_4_20_markers.s"
        subprocess.check_call(
            ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
             "-DFB_PHASE_MARKERS", "--cuda-device-only", "-S", "-o", asm, "rec_12_4_20.hip"],
            cwd=os.path.join(ROOT, "fbstab_amd", "csrc"), stderr=subprocess.DEVNULL)
    lines = open(asm).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and l.split(";")[0].rstrip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    phase, copy = "prologue", 0
    table = collections.OrderedDict()
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"; FBPHASE (\w+)", t)
        if m:
            if m.group(1) == "fwd_top":
                copy += 1
            phase = f"{copy}:{m.group(1)}"
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        mn = t.split()[0]
        table.setdefault(phase, collections.Counter())[classify(mn, t.split(";")[0])] += 1
    # which copy carries the inv(Pi) image (LDS writes between markers 1 and 2)?
    lds12 = {c: table.get(f"{c}:1", {}).get("lds", 0) for c in (1, 2)}
    row_copy = min(lds12, key=lds12.get)
    names = [n for n, _ in CLASSES]
    print(f"copy {row_copy} = newton_step_t<ROW = true> (LDS instructions between markers 1 and 2: {lds12})")
    hdr = f"{'phase':22s}" + "".join(f"{n[:9]:>10s}" for n in names) + f"{'VALU':>8s}{'all':>8s}"
    print(hdr)
    tot = collections.Counter()
    for ph, cnt in table.items():
        if not ph.startswith(f"{row_copy}:"):
            continue
        valu = sum(v for k, v in cnt.items() if k in VALU)
        print(f"{ph:22s}" + "".join(f"{cnt.get(n, 0):10d}" for n in names) + f"{valu:8d}{sum(cnt.values()):8d}")
        tot.update(cnt)
    valu = sum(v for k, v in tot.items() if k in VALU)
    print(f"{'sum_static':22s}" + "".join(f"{tot.get(n, 0):10d}" for n in names) + f"{valu:8d}{sum(tot.values()):8d}")

    # ---- the cooperative passes (real calls: functions of their own in the assembly) ----
    def scan(fn_substr):
        # (the trip loop of a pass is one natural loop: the assembler's block comments say which
        # blocks belong to it - the phase markers do not bracket it, the loop is laid out rotated)
        st = next((i for i, l in enumerate(lines) if fn_substr in l.split(";")[0] and l.split(";")[0].rstrip().endswith(":")
                   and "ILi12ELi4ELi20ELb1ELb0ELi1EE" in l), None)
        if st is None:
            return None
        en = next(i for i in range(st, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
        inside, body, rest = False, collections.Counter(), collections.Counter()
        for l in lines[st + 1:en]:
            t = l.strip()
            if re.match(r"\.LBB\d+_\d+:", t):
                inside = "Loop" in t
                continue
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            (body if inside else rest)[classify(t.split()[0], t.split(";")[0])] += 1
        return body, rest
    v = lambda c: sum(n for k, n in c.items() if k in VALU)
    passes = {}
    print("\ncooperative passes (EXACT instance): VALU / all instructions per trip (the blocks of the trip loop) and outside the loop")
    for name in ("trial_pass_coop", "open_pass_coop", "close_pass_coop", "load_pass_coop"):
        r = scan(name)
        if r is None:
            print(f"  {name}: not found")
            continue
        body, rest = r
        passes[name] = (v(body), v(rest))
        print(f"  {name:18s} per trip {v(body):5d} VALU / {sum(body.values()):5d} all ({body.get('fma_f64', 0)} fma, {body.get('dpp_move', 0)} dpp moves, "
              f"{body.get('vmem', 0)} vmem); outside the loop {v(rest):5d} VALU / {sum(rest.values()):5d} all")

    # ---- dynamic estimate for the 8192-QP BASELINE batch (call counts: profiles/r04_z_r16_wave_time_shares.txt) ----
    calls = {"newton_step": 47634, "trial_pass_coop": 184539, "open_pass_coop": 26813, "close_pass_coop": 18621,
             "load_pass_coop": 8192}
    N1, trips = 31, 8
    fwd = [ph for ph in table if ph.startswith(f"{row_copy}:") and ph.split(":")[1] not in
           ("k_general", "bwd_top", "9", "10", "bwd_end", "fwd_end", "write_top")]
    bwd = [f"{row_copy}:bwd_top", f"{row_copy}:9", f"{row_copy}:10"]
    vf = sum(v(table[ph]) for ph in fwd)
    vb = sum(v(table[ph]) for ph in bwd if ph in table)
    last = sum(v(table[ph]) for ph in fwd if ph.split(":")[1] in ("wwt", "7", "tinv12", "ttt", "8"))  # skipped at the last stage
    est = {"newton_step (forward + backward stage, bounds path, row form)": calls["newton_step"] * (N1 * (vf + vb) - last)}
    for name, (vbody, vrest) in passes.items():
        est[name] = calls[name] * (trips * vbody + vrest)
    total = sum(est.values())
    print(f"\nforward stage {vf} VALU, backward stage {vb} VALU per wavefront (four QPs)")
    for k, n in est.items():
        print(f"  {k:70s} {n / 1e9:6.3f} G")
    import glob
    import json
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_r16_sq_counters.json")))[-1]
    measured = json.load(open(newest))["SQ_INSTS_VALU"]
    print(f"  {'sum':70s} {total / 1e9:6.3f} G   (SQ_INSTS_VALU of the same batch, profiles/{os.path.basename(newest)}: "
          f"{measured / 1e9:.3f} G; the call counts above are the DIAGNOSTIC build's - how many wavefront-level Newton steps "
          f"the rows' 155,251 make depends on the build's timing)")


if __name__ == "__main__":
    main()
