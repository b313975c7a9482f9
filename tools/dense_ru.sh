#!/bin/bash
# Developer tool: registers / spills / scratch of the one-wavefront dense kernel as the
# library build compiles it (not the single-TU diagnostic build).  usage: tools/dense_ru.sh [extra flags]
cd "$(dirname "$0")/../fbstab_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Rpass-analysis=kernel-resource-usage "$@" \
  -c -o /tmp/_dense_ru.o fbstab_hip.hip 2>&1 | grep -A12 "Function Name: _ZN12_GLOBAL__N_124fbstab_dense_wave_kernelILb0" | grep -i "spill\|VGPRs:\|Scratch" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ';'; echo
