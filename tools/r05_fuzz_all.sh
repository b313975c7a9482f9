#!/bin/bash
# Developer tool (GPU box): the round's whole strict shape fuzz - tools/r05_fuzz.sh (20 seeds, dense constraint rows)
# plus six seeds of bound constraints and eight of sparse rows.  usage: tools/r05_fuzz_all.sh <out dir under gpurun_out>
D=gpurun_out/$1
bash tools/r05_fuzz.sh $1
for s in 201 202 203 204 205 206; do
  timeout 600 python tools/fuzz_shapes.py 150 $s all bounds > $D/fuzz_bounds_$s.txt 2>&1
  echo "seed $s (bounds): $(tail -n 1 $D/fuzz_bounds_$s.txt)"
done
for s in 301 302 303 304 305 306 307 308; do
  timeout 600 python tools/fuzz_shapes.py 150 $s all sparse > $D/fuzz_sparse_$s.txt 2>&1
  echo "seed $s (sparse rows): $(tail -n 1 $D/fuzz_sparse_$s.txt)"
done
grep -h "CHECK" -B2 $D/fuzz_*.txt | cut -c1-330
sha256sum fbstab_amd/libfbstab_hip.so
