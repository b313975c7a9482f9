#!/bin/bash
# Developer tool: a variant of ONE record instance (default rec_12_4_20, the headline's) compiled with extra
# flags and linked with the product build's other objects into tools/_build/<name>.so - a minute instead of
# the three of a whole library; for same-box A/B runs (tools/ab_headline.sh).
# usage: tools/variant_rec.sh <name> [-DFLAG ...]        (REC=rec_24_8_16 tools/variant_rec.sh ... for another instance)
set -e
set -o pipefail
cd "$(dirname "$0")/../fbstab_amd/csrc"
name=$1; shift
rec=${REC:-rec_12_4_20}
out=../../tools/_build/$name
mkdir -p $out
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function "$@" -c -o $out/$rec.o $rec.hip
# (a variant object with a DPP hazard - or with no fused instruction at all - is not linked: the product
# Makefile gates on the same check)
case " $* " in *FB_FMAC_DPP=0*) expect="" ;; *) expect="--expect-nonzero" ;; esac
python3 ../../tools/check_dpp_hazards.py $expect $out/$rec.o | tail -1
python3 ../../tools/check_vgpr_budget.py --max 496 --only fbstab_mpc_r16_kernel $out/$rec.o | grep -v probe | grep "over the\|kernel(s) over" || true
objs=""
for o in build/libfbstab_hip/*.o; do
  b=$(basename $o)
  if [ "$b" = "$rec.o" ]; then objs="$objs $out/$rec.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/_build/$name.so $objs
echo "built tools/_build/$name.so"
