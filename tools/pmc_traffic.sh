#!/bin/bash
# Developer tool (GPU box): HBM traffic of one MPC launch, FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 passes.  usage: tools/pmc_traffic.sh <outdir> <batch>
R=$PWD; OUT=$R/$1; B=$2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $cnt --kernel-trace -d $OUT/$cnt -o p -- python3 $R/tools/variant_bench.py $B 1 > $OUT/$cnt.log 2>&1
  echo "$cnt rc=$?"
done
cd $R
for cnt in FETCH_SIZE WRITE_SIZE; do python3 tools/rocpd_summary.py pmc $OUT/$cnt/p_results.db _kernel | python3 -c "
import json,sys
rows=json.load(sys.stdin)
for r in rows: print(r['counter'], r['dispatch_id'], r['value'], 'KiB', r['duration_ns']/1e6, 'ms')
"; done
