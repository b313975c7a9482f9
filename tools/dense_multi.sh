#!/bin/bash
# Developer tool (GPU box): dense configs[1] of several build variants, interleaved.
#   tools/dense_multi.sh <tag> <rounds> <lib>...
T=$1; N=$2; shift 2
O=gpurun_out/$T; mkdir -p $O
for i in $(seq 1 $N); do
  for L in "$@"; do
    FBSTAB_HIP_LIB=$L timeout 300 python tools/dense_bench.py 2>&1 | grep -v amdgpu | head -1 | cut -c1-100 | sed "s|^|$(basename $L .so) |" | tee -a $O/dense.txt
  done
done
