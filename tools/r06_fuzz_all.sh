#!/bin/bash
# Developer tool (GPU box): the round's whole strict shape fuzz - round 5's 34 seeds (tools/r05_fuzz_all.sh: dense
# rows over every instance and over <12,4,20>, bounds, sparse rows) plus the WARM-START family:
# twelve seeds, every shape solved a second time from the device's first solution with a perturbed x0, device and
# oracle given the same guess, every count compared strictly.  usage: tools/r06_fuzz_all.sh <out dir under gpurun_out> [warm-only]
D=gpurun_out/$1
mkdir -p $D
if [ "$2" != "warm-only" ]; then bash tools/r05_fuzz_all.sh $1; fi
for s in 401 402 403 404; do
  timeout 900 python tools/fuzz_shapes.py 150 $s all warm > $D/fuzz_warm_dense_$s.txt 2>&1
  echo "seed $s (warm, dense rows, all instances): $(tail -n 1 $D/fuzz_warm_dense_$s.txt)"
done
for s in 411 412 413 414; do
  timeout 900 python tools/fuzz_shapes.py 150 $s r16 warm > $D/fuzz_warm_r16_$s.txt 2>&1
  echo "seed $s (warm, dense rows, <12,4,20> only): $(tail -n 1 $D/fuzz_warm_r16_$s.txt)"
done
for s in 421 422; do
  timeout 900 python tools/fuzz_shapes.py 150 $s all bounds warm > $D/fuzz_warm_bounds_$s.txt 2>&1
  echo "seed $s (warm, bounds): $(tail -n 1 $D/fuzz_warm_bounds_$s.txt)"
done
for s in 431 432; do
  timeout 900 python tools/fuzz_shapes.py 150 $s all sparse warm > $D/fuzz_warm_sparse_$s.txt 2>&1
  echo "seed $s (warm, sparse rows): $(tail -n 1 $D/fuzz_warm_sparse_$s.txt)"
done
grep -h "CHECK" -B3 $D/fuzz_warm_*.txt | cut -c1-330 | head -80
sha256sum fbstab_amd/libfbstab_hip.so
