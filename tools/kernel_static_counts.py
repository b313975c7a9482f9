#!/usr/bin/env python3
"""Developer tool: static instruction counts by class of one kernel (default: the headline's batch kernel) in
built objects / libraries - for telling two builds apart when a source change moved the register allocation.
usage: tools/kernel_static_counts.py <file.o|.so> ...   [KERNEL=<mangled prefix>]"""
import collections, os, re, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_dpp_hazards as H
KERNEL = os.environ.get("KERNEL", "_ZN12_GLOBAL__N_121fbstab_mpc_r16_kernelILi12ELi4ELi20ELb0ELb1ELb0ELi1EEE")
CLASSES = [("fmac_dpp", r"^v_fmac_f64_dpp"), ("fma_f64", r"^v_(fma|fmac|mul|add)_f64"), ("dpp_mov", r"^v_mov_b(64|32)_dpp"),
           ("accvgpr", r"^v_accvgpr"), ("mov", r"^v_mov_b(32|64)"), ("cndmask", r"^v_cndmask"), ("cmp", r"^v_cmp"),
           ("other_valu", r"^v_"), ("scratch", r"^scratch_"), ("global", r"^global_"), ("ds", r"^ds_"),
           ("waitcnt", r"^s_waitcnt"), ("nop", r"^s_nop"), ("salu", r"^s_")]
for path in sys.argv[1:]:
    with tempfile.TemporaryDirectory() as tmp:
        dis = H.disassemble(path, tmp)
    cnt = collections.Counter()
    for text in dis:
        on = False
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                if not re.match(r"^L\d+$", m.group(1)):   # (--symbolize-operands prints branch targets as <L123>)
                    on = (KERNEL in m.group(1)) if os.environ.get("SUBSTR") else m.group(1).startswith(KERNEL)
                continue
            if not on:
                continue
            t = line.strip()
            if not t or t.startswith(("<", ";")):
                continue
            mn = t.split()[0]
            for name, rx in CLASSES:
                if re.match(rx, mn):
                    cnt[name] += 1
                    break
            else:
                cnt["other"] += 1
    print(f"{os.path.basename(path):28s} " + " ".join(f"{k}={cnt[k]}" for k, _ in CLASSES) + f" total={sum(cnt.values())}")
