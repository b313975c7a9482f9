#!/bin/bash
# Developer tool (here): a variant of the library that differs in fbstab_hip.hip's flags only
# (the record-kernel objects are taken from the default build).  usage: tools/dense_variant.sh <name> <flags...>
N=$1; shift
cd "$(dirname "$0")/../fbstab_amd/csrc" && mkdir -p build/$N ../../tools/_build &&
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function "$@" -c -o build/$N/fbstab_hip.o fbstab_hip.hip &&
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/_build/$N.so build/$N/fbstab_hip.o build/libfbstab_hip/rec_*.o && echo built tools/_build/$N.so
