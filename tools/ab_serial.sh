#!/bin/bash
# Developer tool (GPU box): one batch at a time (bench.py --pipeline 1: the handle is created for a device of its
# own) for several builds of the library, interleaved, N rounds.  usage: tools/ab_serial.sh <rounds> <lib.so> ...
R=$1; shift
for rep in $(seq 1 $R); do
  for L in "$@"; do
    FBSTAB_HIP_LIB=$PWD/$L timeout 300 python bench.py --pipeline 1 --steps 10 --warmup 2 --extras 0 --cpu-sample 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-28s rep $rep: %7.0f QP/s  (%.2f ms per step, newton %.4f)' % ('$(basename $L .so)', d['value'], d['ms_per_step'], d['fp64']['mean_newton_iters']))
"
  done
done
