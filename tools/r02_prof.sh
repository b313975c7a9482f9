# developer script: final profiles of the round (kernel stats of the pipelined bench, traffic, dense)
set -x
O=gpurun_out/r02_$1; mkdir -p $O
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o p -- python3 $R/bench.py --cpu-sample 0 --extras 0 > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err)
python tools/rocpd_summary.py stats $O/stats/p_results.db $O/kernel_stats.csv; cut -c1-160 $O/kernel_stats.csv | head -5
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats_dense -o p -- python3 $R/tools/dense_bench.py > $R/$O/dense_under_rocprof.txt 2>&1)
python tools/rocpd_summary.py stats $O/stats_dense/p_results.db $O/dense_kernel_stats.csv; cut -c1-160 $O/dense_kernel_stats.csv | head -3
bash tools/pmc_traffic.sh $O/traffic 8192 2>&1 | tail -8
bash tools/pmc_dense.sh $O/pmc_dense 2>&1 | tail -8
rm -rf $O/stats/*.db.bak
