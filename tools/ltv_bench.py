#!/usr/bin/env python3
"""Developer tool (GPU box): the bench line's time-varying workload alone (fixtures.synthetic_mpc_ltv_batch: every
stage its own matrices, sparse constraint rows), eight launches in flight and one at a time.
usage: FBSTAB_HIP_LIB=... tools/ltv_bench.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from fbstab_amd import hip_api
from tools import fixtures as fx
B, P = 8192, 8
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
p = fx.synthetic_mpc_ltv_batch(B)
data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
def run(lanes_n, steps):
    lanes = []
    for _ in range(lanes_n):
        s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=B, handles_in_flight=lanes_n)
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
    for ln in lanes:
        ln["s"].close()
    return B * steps / dt, float(o["newton_iters"].mean()), bool((o["eflag"] == 0).all())
v8, nm, ok = run(P, steps)
v1, _, _ = run(1, 6)
print(f"{os.path.basename(hip_api.current_library_path()):24s} ltv eight in flight {v8:9.0f} QP/s   one at a time {v1:9.0f} QP/s   newton mean {nm:.4f} ok={ok}")
