#!/usr/bin/env python3
"""Developer tool: QP/s on a batch of perturbed CopolymerizationReactor problems
(nx=18, nu=5, nc=10; fbstab/test/ocp_generator.cc:73-174).  argv: batch [N]."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbstab_amd import hip_api
from tools import fixtures as fx
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 80
gen = fx.OcpGenerator(); gen.CopolymerizationReactor(N); one = gen.GetFBstabInput()
N, nx, nu, nc = one.sizes()
rng = np.random.default_rng(3)
p = fx.MpcProblem(N, nx, nu, nc)
p.arrays = {k: np.ascontiguousarray(np.broadcast_to(a, (B, a.shape[1]))).copy() for k, a in one.arrays.items()}
p.arrays["x0"] = p.arrays["x0"] * (1.0 + 0.2 * rng.standard_normal((B, nx)))
dev = torch.device("cuda:0")
s = hip_api.FBstabMpcBatch(N, nx, nu, nc, max_batch=B)
data = {k: torch.from_numpy(a).to(dev) for k, a in p.arrays.items()}
mk = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
for rep in range(2):
    z, l, v, y = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
    out = hip_api.out_to_numpy(s.Solve(data, z, l, v, y))
ms = s.last_kernel_ms()
print(f"{s.kernel_name()} batch={B} N={N} kernel_ms={ms:.2f} QP/s={B / (ms * 1e-3):.0f} eflags={np.bincount(out['eflag']).tolist()} "
      f"newton mean={out['newton_iters'].mean():.1f} max={out['newton_iters'].max()} q={s.query()}")
