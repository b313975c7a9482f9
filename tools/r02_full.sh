set -x
O=gpurun_out/r02_$1; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -n 6 $O/pytest.txt
python bench.py > $O/bench_line.json 2> $O/bench_line.err; tail -c 300 $O/bench_line.err
python -c "
import json; d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'serial', round(d['serial']['value']), 'ltv', round(d['ltv_dense_rows']['value']), d['ltv_dense_rows']['mean_newton_iters'], 'dense', round(d['dense']['value']), d['dense']['kernel_ms'], 'receding', round(d['receding']['value']), d['receding']['wall_ms_per_step'], d['receding']['kernel_ms_median'], d['receding']['retired'], 'cpu', round(d['cpu_baseline']['value']))
"
