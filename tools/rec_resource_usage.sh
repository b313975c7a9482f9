#!/bin/bash
# Developer tool: registers / spills / scratch of every kernel and noinline pass of ONE record translation unit
# (default rec_12_4_20) from -Rpass-analysis=kernel-resource-usage.  usage: tools/rec_resource_usage.sh [-DFLAG ...]
cd "$(dirname "$0")/../fbstab_amd/csrc"
rec=${REC:-rec_12_4_20}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast --cuda-device-only -c -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" $rec.hip 2>&1 | python3 -c "
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r'remark: +(.*?) \[-Rpass', line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
for r in rows:
    n = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    n = re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0][-70:]
    print('%-70s VGPR %4s AGPR %4s spill %4s sspill %4s scratch %5s' % (n, r.get('VGPRs','?'), r.get('AGPRs','?'), r.get('VGPRs Spill','?'), r.get('SGPRs Spill','?'), r.get('ScratchSize [bytes/lane]','?')))
"
