set -x
O=gpurun_out/r02_$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reactor or reference_tests or reliable" > $O/pytest_r32.txt 2>&1; tail -n 12 $O/pytest_r32.txt
python tools/reactor_bench.py 4096 80 > $O/reactor_r32.txt 2>&1; tail -n 1 $O/reactor_r32.txt
FBSTAB_HIP_GENERIC=1 python tools/reactor_bench.py 4096 80 > $O/reactor_generic.txt 2>&1; tail -n 1 $O/reactor_generic.txt
python tools/variant_bench.py 8192 2 2>&1 | tail -n 1
