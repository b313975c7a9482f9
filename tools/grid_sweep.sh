#!/bin/bash
# Developer tool (GPU box): the pipelined headline against the number of workgroups per CU of a
# launch (FBSTAB_HIP_WGS_PER_CU) and the number of hardware queues the runtime spreads streams
# over (GPU_MAX_HW_QUEUES).  usage: tools/grid_sweep.sh <tag>
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do
for cfg in "4 4" "3 4" "2 4" "1 4" "4 8" "2 8" "1 8"; do
  set -- $cfg
  for pl in 8 16; do
    FBSTAB_HIP_WGS_PER_CU=$1 GPU_MAX_HW_QUEUES=$2 timeout 300 python bench.py --cpu-sample 0 --extras 0 --pipeline $pl 2>> $O/err.log |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgs/cu $1 hwq $2 pipeline $pl:', round(d['value']), 'QP/s', round(d['ms_per_step'],2), 'ms')" | tee -a $O/sweep.txt
  done
done
done
