#!/bin/bash
# Developer tool (GPU box): every seed of the round's strict fuzz that draws over ALL instances - the development set's
# 24 cold + 8 warm (tools/r06_fuzz_all.sh) and the fresh set's 12 cold + 4 warm (tools/r06_fuzz_fresh.sh) - restricted to the
# shapes ONE record instance takes (fuzz_shapes.py only=<name>: the streams are untouched, the other shapes are skipped):
# for a change that leaves every other instance's object file byte-identical.
# usage: tools/r06_fuzz_one_instance.sh <out dir under gpurun_out> <part of the kernel name, e.g. "<24,8,16>">
D=gpurun_out/$1
K=$2
mkdir -p $D
run() {  # <file tag> <seed> <args...>
  local tag=$1 s=$2; shift 2
  timeout 600 python tools/fuzz_shapes.py 150 $s "$@" "only=$K" > $D/fuzz_${tag}_$s.txt 2>&1
  echo "seed $s ($tag): $(tail -n 1 $D/fuzz_${tag}_$s.txt)"
}
for s in 41 42 43 44 45 46 47 48 49 50 1501 1502 1503 1504 1505 1506; do run dense $s all; done
for s in 201 202 203 204 205 206 1601 1602 1603; do run bounds $s all bounds; done
for s in 301 302 303 304 305 306 307 308 1701 1702 1703; do run sparse $s all sparse; done
for s in 401 402 403 404 1801 1802; do run warm_dense $s all warm; done
for s in 421 422 1821; do run warm_bounds $s all bounds warm; done
for s in 431 432 1831; do run warm_sparse $s all sparse warm; done
grep -h "CHECK" -B3 $D/fuzz_*.txt | cut -c1-330 | head -80
sha256sum fbstab_amd/libfbstab_hip.so
