# developer script: a subset of the GPU tests by -k expression.  usage: r02_k.sh <tag> "<expr>"
set -x
O=gpurun_out/r02_$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "$2" > $O/pytest_k.txt 2>&1; grep -E "passed|failed|^FAILED|^E  " $O/pytest_k.txt | cut -c1-220 | head -40
