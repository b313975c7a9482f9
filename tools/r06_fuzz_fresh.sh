#!/bin/bash
# Developer tool (GPU box): the strict shape fuzz on seeds NO build of this repository was ever run on before the
# final library existed (the 46 seeds of tools/r06_fuzz_all.sh were the development set: a deviation found there was
# closed and the seed re-run).  Same families, same strict comparison (tools/fuzz_shapes.py), fresh seed ranges.
# usage: tools/r06_fuzz_fresh.sh <out dir under gpurun_out>
D=gpurun_out/$1
mkdir -p $D
run() {  # <file tag> <label> <seed> <args...>
  local tag=$1 label=$2 s=$3; shift 3
  timeout 900 python tools/fuzz_shapes.py 150 $s "$@" > $D/fuzz_${tag}_$s.txt 2>&1
  echo "seed $s ($label): $(tail -n 1 $D/fuzz_${tag}_$s.txt)"
}
for s in 1501 1502 1503 1504 1505 1506; do run dense "dense rows, all instances" $s all; done
for s in 1511 1512 1513 1514; do run r16 "dense rows, <12,4,20> only" $s r16; done
for s in 1601 1602 1603; do run bounds "bounds" $s all bounds; done
for s in 1701 1702 1703; do run sparse "sparse rows" $s all sparse; done
for s in 1801 1802; do run warm_dense "warm, dense rows, all instances" $s all warm; done
for s in 1811 1812; do run warm_r16 "warm, dense rows, <12,4,20> only" $s r16 warm; done
run warm_bounds "warm, bounds" 1821 all bounds warm
run warm_sparse "warm, sparse rows" 1831 all sparse warm
grep -h "CHECK" -B3 $D/fuzz_*.txt | cut -c1-330 | head -120
sha256sum fbstab_amd/libfbstab_hip.so
