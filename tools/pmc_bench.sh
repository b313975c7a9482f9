#!/bin/bash
# Developer tool (GPU box): counter passes over the pipelined bench (8 launches in
# flight).  usage: tools/pmc_bench.sh <outdir> ; each pass its own process, --pmc
# with --kernel-trace only, wrapped in timeout.
R=$PWD; OUT=$R/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
 "FETCH_SIZE" "WRITE_SIZE" \
 "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU" \
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" \
 "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
 "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCC_HIT TCC_MISS TCC_REQ" \
 "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ_64B" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p -- python3 $R/bench.py --cpu-sample 0 --steps 16 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? $set"
done
cd $R
for j in $(seq 1 $i); do python3 tools/rocpd_summary.py pmc $OUT/p$j/p_results.db r16_kernel > $OUT/p$j.json 2>/dev/null; done
python3 - <<PY
import json,glob,collections
for f in sorted(glob.glob("$OUT/p*.json")):
    try: rows=json.load(open(f))
    except Exception as e: print(f,"unreadable"); continue
    agg=collections.defaultdict(list)
    for r in rows: agg[r["counter"]].append(r["value"])
    print(f.split("/")[-1], {k: (sum(v)/len(v), len(v)) for k,v in agg.items()})
PY
