#!/bin/bash
# Round 4: calibration of the per-step elimination-order choice of the one-wavefront dense
# kernel (fbstab_hip_dense_set_factorisation): config 2 timing per mode, then the degenerate
# fuzz family on every mode against one oracle solve.  $1 = output directory under gpurun_out,
# $2 = comma-separated modes (tools/fuzz_dense.py).
set -e
O=gpurun_out/${1:-r04_a}
MODES=${2:-pivoted,natural,16,24,32,40,16n,24n,32n}
mkdir -p $O
for m in ${MODES//,/ }; do
  python tools/dense_bench.py 4096 $m > $O/dense_bench_$m.txt 2>&1
  head -2 $O/dense_bench_$m.txt
done
python tools/fuzz_dense.py 200 11 64 $MODES > $O/fuzz_dense_modes.txt 2>&1
tail -12 $O/fuzz_dense_modes.txt
