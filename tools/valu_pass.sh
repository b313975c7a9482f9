#!/bin/bash
# Developer tool (GPU box): ONE counter pass (the SQ instruction counts) over one 8192-QP launch of a library's
# MPC kernel - the quick form of tools/pmc_lib.sh for an instruction-count A/B.  usage: tools/valu_pass.sh <lib.so> ...
R=$PWD
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  export FBSTAB_HIP_LIB=$R/$L
  D=/tmp/valu_$(basename $L .so); rm -rf $D
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $D -o p -- python3 $R/tools/variant_bench.py 8192 1 > $D.log 2>&1
  python3 $R/tools/rocpd_summary.py pmc $D/p_results.db fbstab_ 2>/dev/null | python3 -c "
import json, sys
rows = json.load(sys.stdin)
last = max(r['dispatch_id'] for r in rows)
print('$(basename $L .so)', {r['counter']: r['value'] for r in rows if r['dispatch_id'] == last}, 'kernel_ms', [round(r['duration_ns'] / 1e6, 2) for r in rows if r['dispatch_id'] == last][:1])
"
done
