#!/bin/bash
# Developer tool (GPU box): the headline as the driver launches it for N > 1, with ONE rank - the gather
# path (RCCL process group, one gather per batch) on one device - beside the plain single-process run.
# Prints one summary line per run (the bench lines themselves: pipe bench.py's stdout through
# `grep '^{'` - RCCL writes its banner to the same stream, and a file that starts with it is not JSON).
# usage: tools/torchrun_1rank.sh <rounds>
R=${1:-2}
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-22s %7.0f QP/s  (%.2f ms per step)' % ('$1', d['value'], d['ms_per_step']), flush=True)
"; }
for rep in $(seq 1 $R); do
  timeout 300 python bench.py --extras 0 --cpu-sample 0 2>/dev/null | show "plain rep $rep"
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --extras 0 --cpu-sample 0 2>/dev/null | show "torchrun 1 rank rep $rep"
done
