# developer script: the N > 1 code paths of bench.py rehearsed with ONE rank under
# torch.distributed.run (RCCL group of one: same gathers, same barriers), then the plain run
set -x
O=gpurun_out/r02_$1; mkdir -p $O
FBSTAB_BENCH_SHARDED_SWEEP=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 16 --warmup 2 --cpu-sample 0 > $O/bench_torchrun_1rank.json 2> $O/bench_torchrun_1rank.err; tail -c 400 $O/bench_torchrun_1rank.err
python - "$O" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_torchrun_1rank.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), d["config"]["parallelism"])
r = d["receding"]; print("receding", round(r["value"]), r["wall_ms_per_step"], r["n_gpus"], r["retired"], r["config"][-70:])
PY
