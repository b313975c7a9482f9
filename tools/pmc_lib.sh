#!/bin/bash
# Developer tool (GPU box): traffic and SQ counter passes over ONE launch of the MPC
# kernel of a build variant.  usage: tools/pmc_lib.sh <outdir> <lib.so> [batch]
# Each pass is its own process, --pmc with --kernel-trace only, wrapped in timeout.
# PMC_PROG="tools/dense_bench.py" takes the passes over the dense kernel instead (the last
# dispatch of the program's solver kernel is the one reported either way).
R=$PWD; OUT=$R/$1; export FBSTAB_HIP_LIB=$R/$2; B=${3:-8192}; mkdir -p $OUT
PROG=${PMC_PROG:-tools/variant_bench.py}; ARGS="$B 1"; [ -n "$PMC_PROG" ] && ARGS="${PMC_ARGS:-}"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
 "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" \
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" \
 "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCC_HIT TCC_MISS TCC_REQ" \
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p -- python3 $R/$PROG $ARGS > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
for j in $(seq 1 $i); do python3 tools/rocpd_summary.py pmc $OUT/p$j/p_results.db fbstab_ > $OUT/p$j.json 2>/dev/null; rm -rf $OUT/p$j; done
python3 - <<PY
import json,glob
# the solver kernel's dispatches (a warm-up and the timed ones): counters of the LAST one
tot={}
for f in sorted(glob.glob("$OUT/p*.json")):
    try: rows=json.load(open(f))
    except Exception as e: print(f,"unreadable"); continue
    if not rows: continue
    last=max(r["dispatch_id"] for r in rows)
    for r in rows:
        if r["dispatch_id"]==last:
            tot[r["counter"]]=r["value"]; tot.setdefault("kernel_ms_under_pmc",[]).append(round(r["duration_ns"]/1e6,3))
tot["kernel_ms_under_pmc"]=sorted(set(tot.get("kernel_ms_under_pmc",[])))
import hashlib, os
tot["library_sha256"]=hashlib.sha256(open(os.environ["FBSTAB_HIP_LIB"],"rb").read()).hexdigest()
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    tot["hbm_bytes_per_launch_raw"]=(tot["FETCH_SIZE"]+tot["WRITE_SIZE"])*1024
    tot["hbm_bytes_per_launch_fetch_doubled"]=(2*tot["FETCH_SIZE"]+tot["WRITE_SIZE"])*1024
json.dump(tot, open("$OUT/summary.json","w"), indent=1)
print(json.dumps(tot))
PY
