#!/bin/bash
# Developer tool (GPU box): SQ counter passes over tools/variant_bench.py.
# usage: tools/pmc_passes.sh <outdir> <batch> ; each pass is its own process,
# --pmc with --kernel-trace only, wrapped in timeout.
R=$PWD; OUT=$R/$1; B=$2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
 "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" \
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" \
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
 "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCC_HIT TCC_MISS TCC_REQ" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p -- python3 $R/tools/variant_bench.py $B 1 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
for i in 1 2 3 4; do python3 tools/rocpd_summary.py pmc $OUT/p$i/p_results.db _kernel > $OUT/p$i.json 2>/dev/null; done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/p*.json")):
    try: rows=json.load(open(f))
    except Exception as e: print(f,"unreadable"); continue
    last=max(r["dispatch_id"] for r in rows) if rows else None
    print(f, {r["counter"]: r["value"] for r in rows if r["dispatch_id"]==last})
PY
