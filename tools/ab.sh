#!/bin/bash
# Developer tool (GPU box): A/B of build variants of the library on ONE box.
#   tools/ab.sh <tag> <lib.so> [<lib.so> ...]        results under gpurun_out/<tag>/
# Per variant: the Newton-step / synthetic-batch parity tests, one launch at a time
# (tools/variant_bench.py) and the pipelined headline (bench.py, no extras, twice).
# AB_TESTS=0 skips the tests, AB_K="..." overrides their selection.
T=$1; shift
O=gpurun_out/$T
mkdir -p $O
K=${AB_K:-"newton_step or paths_agree or synthetic_batch or time_varying or smaller_shapes or random_time"}
for L in "$@"; do
  N=$(basename $L .so)
  if [ "${AB_TESTS:-1}" != "0" ]; then
    FBSTAB_HIP_LIB=$L timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$K" > $O/pytest_$N.txt 2>&1
    tail -n 3 $O/pytest_$N.txt
  fi
  FBSTAB_HIP_LIB=$L timeout 300 python tools/variant_bench.py 8192 3 > $O/serial_$N.txt 2>&1
  FBSTAB_HIP_LIB=$L timeout 300 python bench.py --cpu-sample 0 --extras 0 > $O/bench_$N.json 2> $O/bench_$N.err
  FBSTAB_HIP_LIB=$L timeout 300 python bench.py --cpu-sample 0 --extras 0 >> $O/bench_$N.json 2>> $O/bench_$N.err
done
python - "$O" "$@" <<'PY'
import json, sys, os
O = sys.argv[1]
for L in sys.argv[2:]:
    N = os.path.basename(L)[:-3]
    try:
        vals = [json.loads(l) for l in open(f"{O}/bench_{N}.json") if l.startswith("{")]
        print(N, [round(d["value"]) for d in vals], "QP/s", [round(d["ms_per_step"], 2) for d in vals], vals[0]["all_converged"])
    except Exception as e:
        print(N, "bench failed", e)
    try:
        print(open(f"{O}/serial_{N}.txt").read().strip().splitlines()[0][:220])
    except Exception as e:
        print(N, "serial failed", e)
PY
