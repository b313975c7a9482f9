#!/bin/bash
# Developer tool (GPU box): VERDICT r4 item 1's fuzz criterion - tools/fuzz_shapes.py 150 on seeds 41..50 (every
# instance) and on ten more seeds restricted to shapes of the headline instance <12,4,20>, strict comparison
# of every count with the oracle's.  usage: tools/r05_fuzz.sh <out dir under gpurun_out>
D=gpurun_out/$1
mkdir -p $D
for s in 41 42 43 44 45 46 47 48 49 50; do
  timeout 600 python tools/fuzz_shapes.py 150 $s > $D/fuzz_all_$s.txt 2>&1
  echo "seed $s (all instances): $(tail -1 $D/fuzz_all_$s.txt)"
done
for s in 141 142 143 144 145 146 147 148 149 150; do
  timeout 600 python tools/fuzz_shapes.py 150 $s r16 > $D/fuzz_r16_$s.txt 2>&1
  echo "seed $s (<12,4,20> only): $(tail -1 $D/fuzz_r16_$s.txt)"
done
grep -h "CHECK" -B2 $D/fuzz_*.txt | head -40
