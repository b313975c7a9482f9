#!/usr/bin/env python3
"""Developer tool: time the dense kernel on BASELINE config 2
(batch 4096, nz=50 nl=10 nv=100) and compare with the oracle on a sample."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbstab_amd import hip_api
from tools import fixtures as fx
from oracle.oracle_py import Oracle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "default"  # default (= pivoted) | auto | pivoted | natural | <spread bits>[a<act bits>]
nz, nl, nv = 50, 10, 100
p = fx.synthetic_dense_batch(B, nz, nl, nv)
dev = torch.device("cuda:0")
os.environ["FBSTAB_HIP_DENSE_STICKY"] = "0" if mode.endswith("n") else "1"
if mode[0].isdigit():
    os.environ["FBSTAB_HIP_DENSE_ACT_BITS"] = mode.rstrip("n").split("a")[1] if "a" in mode else "0"
s = hip_api.FBstabDenseBatch(nz, nl, nv, max_batch=B)
if mode == "pivoted":
    s.SetFactorisation(s.ORDER_PIVOTED)
elif mode == "natural":
    s.SetFactorisation(s.ORDER_NATURAL)
elif mode == "auto":
    s.SetFactorisation(s.ORDER_AUTO)
elif mode != "default":
    s.SetFactorisation(s.ORDER_AUTO, int(mode.rstrip("n").split("a")[0]))
data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
mk = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
for rep in range(2):
    z, l, v, y = mk(nz), mk(nl), mk(nv), mk(nv)
    out = hip_api.out_to_numpy(s.Solve(data, z, l, v, y))
    ms = s.last_kernel_ms()
print(f"[{mode}] factorisation={s.Factorisation()}")
print(f"dense batch={B} kernel_ms={ms:.2f} QP/s={B / (ms * 1e-3):.0f} ok={(out['eflag'] == 0).all()} "
      f"newton mean={out['newton_iters'].mean():.2f} max={out['newton_iters'].max()} q={s.query()}")
alg = 68680 * B
print(f"algorithmic GB/s = {alg / (ms * 1e-3) / 1e9:.2f}")
orc = Oracle(False)
n = min(B, 512)
q = fx.synthetic_dense_batch(n, nz, nl, nv)
t0 = time.perf_counter(); o = orc.solve_dense(q, nthreads=orc.num_threads()); t1 = time.perf_counter()
print(f"oracle {orc.num_threads()} threads: {n / (t1 - t0):.0f} QP/s; newton equal: "
      f"{np.array_equal(o[4]['newton_iters'], out['newton_iters'][:n])}")
