#!/bin/bash
# Developer tool (GPU box): the pipelined headline of two build variants, interleaved
# A B A B ... on one box (order and warm-up effects cancel).  usage: tools/abab.sh <tag> <libA> <libB> [rounds]
T=$1; A=$2; B=$3; N=${4:-3}
O=gpurun_out/$T; mkdir -p $O
for i in $(seq 1 $N); do
  for L in $A $B; do
    FBSTAB_HIP_LIB=$L timeout 300 python bench.py --cpu-sample 0 --extras 0 >> $O/bench_$(basename $L .so).json 2>> $O/bench.err
  done
done
python - "$O" "$A" "$B" <<'PY'
import json, sys, os
O = sys.argv[1]
for L in sys.argv[2:]:
    N = os.path.basename(L)[:-3]
    vals = [json.loads(l) for l in open(f"{O}/bench_{N}.json") if l.startswith("{")]
    print(N, [round(d["value"]) for d in vals], "QP/s", [round(d["ms_per_step"], 2) for d in vals])
PY
