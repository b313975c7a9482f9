#!/usr/bin/env python3
"""Developer tool: registers / spills / LDS per kernel of the HIP library, from
`hipcc -Rpass-analysis=kernel-resource-usage` (compiles to a scratch output).
usage: tools/resource_usage.py [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "fbstab_amd", "csrc", "fbstab_hip.hip")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
       "-ffp-contract=fast", "-Rpass-analysis=kernel-resource-usage", "-DFB_SINGLE_TU", "-o", "/tmp/_ru.so", src] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark: +(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    print(f"{name:60s} VGPR {r.get('VGPRs', '?'):>4s} AGPR {r.get('AGPRs', '?'):>4s} spill {r.get('VGPRs Spill', '?'):>4s} "
          f"SGPR {r.get('SGPRs', '?'):>4s} sspill {r.get('SGPRs Spill', '?'):>4s} scratch {r.get('ScratchSize [bytes/lane]', '?'):>5s} "
          f"occ {r.get('Occupancy [waves/SIMD]', '?')} LDS {r.get('LDS Size [bytes/block]', '?')}")
