# developer script: a subset of the GPU tests by -k expression.  usage: r02_k.sh <tag> "<expr>"
set -x
O=gpurun_out/r02_$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$2" > $O/pytest_k.txt 2>&1; tail -n 25 $O/pytest_k.txt
