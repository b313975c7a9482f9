# developer script: A/B of build variants on one box.  argv: tag lib1 lib2 ...
set -x
T=$1; shift
O=gpurun_out/r02_$T
mkdir -p $O
for L in "$@"; do
  N=$(basename $L .so)
  FBSTAB_HIP_LIB=$L timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "newton_step or paths_agree or synthetic_batch or time_varying or smaller_shapes or random_time" > $O/pytest_$N.txt 2>&1
  tail -n 2 $O/pytest_$N.txt
  FBSTAB_HIP_LIB=$L python tools/variant_bench.py 8192 3 > $O/serial_$N.txt 2>&1
  FBSTAB_HIP_LIB=$L python bench.py --cpu-sample 0 > $O/bench_$N.json 2> $O/bench_$N.err
  FBSTAB_HIP_LIB=$L python bench.py --cpu-sample 0 >> $O/bench_$N.json 2>> $O/bench_$N.err
done
python - "$O" "$@" <<'PY'
import json, sys, os
O = sys.argv[1]
for L in sys.argv[2:]:
    N = os.path.basename(L)[:-3]
    try:
        vals = [json.loads(l) for l in open(f"{O}/bench_{N}.json") if l.startswith("{")]
        print(N, [round(d["value"]) for d in vals], "QP/s", [round(d["ms_per_step"], 2) for d in vals], vals[0]["all_converged"])
    except Exception as e:
        print(N, "bench failed", e)
    print(open(f"{O}/serial_{N}.txt").read().strip().splitlines()[-1][:200])
PY
