# developer script: A/B of two library builds on the dense path (tests on the new one)
# usage: r02_dense3.sh <tag>
set -x
O=gpurun_out/r02_$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense" > $O/pytest_dense.txt 2>&1; tail -n 5 $O/pytest_dense.txt
for rep in 1 2; do
FBSTAB_HIP_LIB=$PWD/fbstab_amd/var_base.so timeout 300 python tools/dense_bench.py > $O/dense_base_$rep.txt 2>&1; head -n 3 $O/dense_base_$rep.txt | tail -n 2 | head -1; tail -n 1 $O/dense_base_$rep.txt
timeout 300 python tools/dense_bench.py > $O/dense_new_$rep.txt 2>&1; head -n 3 $O/dense_new_$rep.txt | tail -n 2 | head -1; tail -n 1 $O/dense_new_$rep.txt
done
