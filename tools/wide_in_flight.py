#!/usr/bin/env python3
"""Developer tool (GPU box): the bench line's `wide` workloads as a STREAM of batches - P launches in flight on P streams
(the headline's regime) beside one launch at a time.  Do launches of the row-pair instances share the chip?  (The
<24,8,*> kernels allocate 512 registers through their noinline sweeps: LABNOTES R6.3.)
usage: tools/wide_in_flight.py [steps [lanes:share,...]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from fbstab_amd import hip_api
from tools import fixtures as fx
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")

def workload(name):
    if name == "ltv_30_20_6_16":
        one = fx.random_ltv_mpc(np.random.default_rng(5), 64, 30, 20, 6, 16)
        p = fx.MpcProblem(30, 20, 6, 16)
        p.arrays = {k: np.ascontiguousarray(np.tile(a, (32, 1))) for k, a in one.arrays.items()}
        return p
    gen = fx.OcpGenerator()
    gen.CopolymerizationReactor(80)
    one = gen.GetFBstabInput()
    N, nx, nu, nc = one.sizes()
    B = 1024
    rng = np.random.default_rng(3)
    p = fx.MpcProblem(N, nx, nu, nc)
    p.arrays = {k: np.ascontiguousarray(np.broadcast_to(a, (B, a.shape[1]))).copy() for k, a in one.arrays.items()}
    p.arrays["x0"] = p.arrays["x0"] * (1.0 + 0.2 * rng.standard_normal((B, nx)))
    return p

def run(p, data, lanes_n, steps, share):
    B = p.batch
    lanes = []
    for _ in range(lanes_n):
        s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=B, handles_in_flight=share)
        mk = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
        lanes.append(dict(s=s, st=torch.cuda.Stream(device=dev), z=mk(p.nz), l=mk(p.nl), v=mk(p.nv), y=mk(p.nv),
                          out=torch.zeros((B, 40), dtype=torch.uint8, device=dev)))
    def step(k):
        ln = lanes[k % lanes_n]
        with torch.cuda.stream(ln["st"]):
            for a in (ln["z"], ln["l"], ln["v"]):
                a.zero_()
            ln["s"].Solve(data, ln["z"], ln["l"], ln["v"], ln["y"], out=ln["out"], stream=ln["st"].cuda_stream, async_=True)
    for k in range(lanes_n):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    o = hip_api.out_to_numpy(lanes[0]["out"])
    q = lanes[0]["s"].query()
    for ln in lanes:
        ln["s"].close()
    return B * steps / dt, float(o["newton_iters"].mean()), bool((o["eflag"] == 0).all()), q["workgroups"]

for name in ("ltv_30_20_6_16", "reactor_N80"):
    p = workload(name)
    data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
    v1, nm, ok, w1 = run(p, data, 1, max(steps // 2, 3), 1)
    print(f"{name:16s} one at a time            {v1:9.0f} QP/s  ({w1} workgroups)  newton mean {nm:.3f} ok={ok}", flush=True)
    combos = ((2, 1), (2, 2), (4, 2), (4, 4), (8, 8))
    if len(sys.argv) > 2:  # "lanes:share,lanes:share,..."
        combos = tuple(tuple(int(x) for x in c.split(":")) for c in sys.argv[2].split(","))
    for lanes_n, share in combos:
        v, nm, ok, w = run(p, data, lanes_n, steps, share)
        print(f"{name:16s} {lanes_n} in flight, share {share}    {v:9.0f} QP/s  ({w} workgroups each)  ok={ok}", flush=True)
    del data
    torch.cuda.empty_cache()
