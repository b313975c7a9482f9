# developer script: one GPU round of tests + measurements; argv[1] = tag
set -x
T=${1:-step}
O=gpurun_out/r02_$T
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -n 5 $O/pytest.txt
python bench.py --cpu-sample 0 > $O/bench.json 2> $O/bench.err
python bench.py --cpu-sample 0 --pipeline 1 --steps 6 > $O/bench_serial.json 2>> $O/bench.err

FBSTAB_HIP_LIB=fbstab_amd/var_clock.so python tools/stamp_report.py 8192 > $O/clock_8192.txt 2>&1
FBSTAB_HIP_LIB=fbstab_amd/var_clock.so python tools/stamp_report.py 4 > $O/clock_4.txt 2>&1
python - <<PY
import json
for f in ("bench", "bench_serial", "bench_half_occupancy"):
    try:
        d = json.loads(open("$O/%s.json" % f).read())
        print(f, round(d["value"]), "QP/s", round(d["ms_per_step"], 2), "ms/step", d["all_converged"], d["launch"])
    except Exception as e:
        print(f, "failed", e)
PY
