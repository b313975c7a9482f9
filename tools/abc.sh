#!/bin/bash
# Developer tool (GPU box): the pipelined headline of several build variants, interleaved.
#   tools/abc.sh <tag> <rounds> <lib>...
T=$1; N=$2; shift 2
O=gpurun_out/$T; mkdir -p $O
for i in $(seq 1 $N); do
  for L in "$@"; do
    FBSTAB_HIP_LIB=$L timeout 300 python bench.py --cpu-sample 0 --extras 0 2>> $O/bench.err |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $L .so)', round(d['value']), 'QP/s', round(d['ms_per_step'],2), 'ms')" | tee -a $O/abc.txt
  done
done
