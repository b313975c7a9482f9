#!/bin/bash
# Developer tool (GPU box): the measurements a round's profiles/ entries come from.
#   tools/round_profile.sh <tag>      results under gpurun_out/<tag>/
T=$1; O=gpurun_out/$T; mkdir -p $O; R=$PWD
if [ -f tools/_build/var_stamp.so ]; then
  FBSTAB_HIP_LIB=tools/_build/var_stamp.so timeout 300 python tools/stamp_report.py 8192 2>&1 | grep -v amdgpu > $O/mpc_wave_time_shares.txt
  FBSTAB_HIP_LIB=tools/_build/var_stamp.so timeout 300 python tools/dense_stamp.py 2>&1 | grep -v amdgpu > $O/dense_wave_time_shares.txt
  tail -8 $O/mpc_wave_time_shares.txt; tail -24 $O/dense_wave_time_shares.txt
fi
bash tools/pmc_lib.sh $O/pmc_mpc fbstab_amd/libfbstab_hip.so 8192 > $O/pmc_mpc.log 2>&1; tail -1 $O/pmc_mpc.log | cut -c1-400
PMC_PROG=tools/dense_bench.py bash tools/pmc_lib.sh $O/pmc_dense fbstab_amd/libfbstab_hip.so > $O/pmc_dense.log 2>&1; tail -1 $O/pmc_dense.log | cut -c1-300
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_line.err; tail -c 300 $O/bench_line.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o s -- python3 $R/bench.py --cpu-sample 0 --extras 0 > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err)
python3 tools/rocpd_summary.py stats $O/stats/s_results.db $O/kernel_stats_pipelined_bench.csv 2>/dev/null; rm -rf $O/stats; head -5 $O/kernel_stats_pipelined_bench.csv
timeout 300 python tools/sharded_rehearsal.py 8192 2>&1 | grep -v amdgpu | tail -1 > $O/sharded_rehearsal.json; cat $O/sharded_rehearsal.json
python3 -c "
import json; d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'serial', round(d['serial']['value']), 'ltv', round(d['ltv_dense_rows']['value']), 'dense', round(d['dense']['value']), d['dense']['kernel_ms'], 'receding', round(d['receding']['value']), 'cpu', round(d['cpu_baseline']['value']), d['cpu_baseline'].get('affinity_cpus'), d['cpu_baseline'].get('cgroup_cpu_quota'))
"
