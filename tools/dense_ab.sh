#!/bin/bash
# Developer tool (GPU box): dense configs[1] of two build variants, interleaved.
#   tools/dense_ab.sh <tag> <libA> <libB> [rounds]
T=$1; A=$2; B=$3; N=${4:-3}
O=gpurun_out/$T; mkdir -p $O
for i in $(seq 1 $N); do
  for L in $A $B; do
    FBSTAB_HIP_LIB=$L timeout 300 python tools/dense_bench.py 2>&1 | grep -v amdgpu | head -1 | sed "s|^|$(basename $L .so) |" | tee -a $O/dense.txt
  done
done
