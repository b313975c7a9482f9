#!/bin/bash
# Developer tool (GPU box): rocprofv3 --kernel-trace of the pipelined headline for one library: per-dispatch
# start / duration of the MPC kernel (is the chip shared the way it should be?).  usage: tools/trace_pipelined.sh <lib.so> <out.txt>
R=$PWD
export FBSTAB_HIP_LIB=$R/$1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trp
timeout 300 rocprofv3 --kernel-trace -d /tmp/trp -o t -- python3 $R/bench.py --steps 16 --warmup 2 --extras 0 --cpu-sample 0 > /tmp/trp.log 2>&1
python3 - "$R/$2" <<'PY'
import sqlite3, sys, glob
db = glob.glob("/tmp/trp/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
out = open(sys.argv[1], "w")
view = "kernels" if "kernels" in tabs else None
if view is None:
    out.write("tables: %s\n" % tabs); sys.exit(0)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
out.write("columns: %s\n" % cols)
rows = con.execute("select name, start, end, queue_id, stream_id, grid_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels order by start").fetchall()
t0 = rows[0][1]
for r in rows:
    if "fbstab_mpc" in r[0]:
        out.write("%-40s start %10.3f ms dur %9.3f ms %s\n" % (r[0][:40], (r[1] - t0) / 1e6, (r[2] - r[1]) / 1e6, r[3:]))
PY
grep '^{' /tmp/trp.log | python3 -c "import sys,json; [print('bench value', json.loads(l)['value']) for l in sys.stdin]"
