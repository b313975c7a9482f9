#!/bin/bash
# Developer tool (GPU box): the headline number (bench.py --extras 0 --cpu-sample 0) against the size of
# the persistent grid (FBSTAB_HIP_WGS_PER_CU: wavefronts per CU a launch may occupy) and the number of
# launches in flight.  usage: tools/wgs_sweep.sh <rounds> "<wgs list>" "<pipeline list>" [batch]
R=$1; WL=$2; PL=$3; B=${4:-8192}
for rep in $(seq 1 $R); do
  for W in $WL; do
    for P in $PL; do
      FBSTAB_HIP_WGS_PER_CU=$W timeout 300 python bench.py --extras 0 --cpu-sample 0 --pipeline $P --batch $B 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('wgs/CU $W pipeline %2d batch $B rep $rep: %7.0f QP/s  (%.2f ms per step)' % ($P, d['value'], d['ms_per_step']), flush=True)
"
    done
  done
done
