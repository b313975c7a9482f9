#!/bin/bash
# Developer tool (GPU box): the whole bench line for two or more builds of the library, interleaved.
# usage: tools/ab_bench.sh <outdir under gpurun_out> <lib.so> <lib.so> ...
O=gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
  for L in "$@"; do
    N=$(basename $L .so)
    FBSTAB_HIP_LIB=$PWD/$L timeout 600 python bench.py --cpu-sample 0 > $O/bench_${N}_$rep.json 2>> $O/bench.err
    python3 - <<PY
import json
d=json.loads(open("$O/bench_${N}_$rep.json").read().strip().splitlines()[-1])
print("$N rep $rep: headline %.0f serial %.0f ltv %.0f (newton %.2f, %.3g it/s) dense %.0f receding %.0f" % (d["value"], d["serial"]["value"], d["ltv_dense_rows"]["value"], d["ltv_dense_rows"]["mean_newton_iters"], d["ltv_dense_rows"]["newton_iters_per_sec"], d["dense"]["value"], d["receding"]["value"]))
PY
  done
done
