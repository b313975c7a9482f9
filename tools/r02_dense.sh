set -x
O=gpurun_out/r02_$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_display.py tests/test_facade.py -m gpu -x -q -k "dense or display or facade" > $O/pytest_dense.txt 2>&1; tail -n 3 $O/pytest_dense.txt
python tools/dense_bench.py > $O/dense_256.txt 2>&1; tail -n 3 $O/dense_256.txt
FBSTAB_HIP_DENSE_THREADS=64 python tools/dense_bench.py > $O/dense_wave.txt 2>&1; tail -n 3 $O/dense_wave.txt
FBSTAB_HIP_LIB=fbstab_amd/var_clock.so python tools/dense_stamp.py > $O/dense_stamp_256.txt 2>&1; cat $O/dense_stamp_256.txt
