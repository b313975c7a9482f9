#!/bin/bash
# Developer tool (GPU box): SQ / cache counter passes over tools/dense_bench.py (one
# dense launch of config 2).  usage: tools/pmc_dense.sh <outdir> ; each pass is its
# own process, --pmc with --kernel-trace only, wrapped in timeout.
R=$PWD; OUT=$R/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
 "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" \
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" \
 "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
 "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCC_HIT TCC_MISS TCC_REQ" \
 "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p -- python3 $R/tools/dense_bench.py > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? $set"
done
cd $R
for j in $(seq 1 $i); do python3 tools/rocpd_summary.py pmc $OUT/p$j/p_results.db dense > $OUT/p$j.json 2>/dev/null; done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/p*.json")):
    try: rows=json.load(open(f))
    except Exception as e: print(f,"unreadable"); continue
    last=max(r["dispatch_id"] for r in rows) if rows else None
    print(f.split("/")[-1], {r["counter"]: r["value"] for r in rows if r["dispatch_id"]==last}, [r["duration_ns"] for r in rows if r["dispatch_id"]==last][:1])
PY
