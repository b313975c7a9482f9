set -x
O=gpurun_out/r02_$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "32_constraint or random_time_varying or synthetic_batch" > $O/pytest_boxed.txt 2>&1; tail -n 8 $O/pytest_boxed.txt
python tools/boxed_bench.py 8192 > $O/boxed_r16.txt 2>&1; tail -n 1 $O/boxed_r16.txt
FBSTAB_HIP_GENERIC=1 python tools/boxed_bench.py 2048 > $O/boxed_generic.txt 2>&1; tail -n 1 $O/boxed_generic.txt
python tools/variant_bench.py 8192 2 2>&1 | tail -n 1
