# developer script: dense parity tests + config-2 timing of the one-wavefront kernel
# usage: r02_dense2.sh <tag> [lib]   (lib: a build without the record kernels is enough)
set -x
O=gpurun_out/r02_$1; mkdir -p $O
L=${2:-fbstab_amd/libfbstab_hip.so}
FBSTAB_HIP_LIB=$L timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense" > $O/pytest_dense.txt 2>&1; tail -n 5 $O/pytest_dense.txt
FBSTAB_HIP_LIB=$L timeout 300 python tools/dense_bench.py > $O/dense_wave.txt 2>&1; tail -n 3 $O/dense_wave.txt
for W in 2 4 6; do FBSTAB_HIP_WGS_PER_CU=$W FBSTAB_HIP_LIB=$L timeout 300 python tools/dense_bench.py > $O/dense_wave_w$W.txt 2>&1; tail -n 3 $O/dense_wave_w$W.txt | head -1; done
if [ -f fbstab_amd/var_dense_clock.so ]; then FBSTAB_HIP_LIB=fbstab_amd/var_dense_clock.so timeout 300 python tools/dense_stamp.py > $O/dense_stamp_wave.txt 2>&1; cat $O/dense_stamp_wave.txt; fi
