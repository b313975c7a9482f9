set -x
O=gpurun_out/r02_k; mkdir -p $O
python tools/sweep_probe.py > $O/sweep.txt 2>&1; cat $O/sweep.txt | tail -2
R=$PWD; (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python3 $R/tools/sweep_probe.py 4096 60 > $R/$O/sweep_prof.txt 2>&1)
python tools/rocpd_summary.py stats $O/prof/p_results.db $O/sweep_kernel_stats.csv; cat $O/sweep_kernel_stats.csv | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense" > $O/pytest_dense.txt 2>&1; tail -n 5 $O/pytest_dense.txt
python tools/dense_bench.py > $O/dense_wave.txt 2>&1; tail -n 3 $O/dense_wave.txt
FBSTAB_HIP_DENSE_THREADS=256 python tools/dense_bench.py > $O/dense_256.txt 2>&1; tail -n 3 $O/dense_256.txt
