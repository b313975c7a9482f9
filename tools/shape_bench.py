#!/usr/bin/env python3
"""Developer tool: QP/s of random time-varying MPC problems of a given shape, one launch
at a time (64 distinct problems tiled over the batch).  argv: N nx nu nc batch.
FBSTAB_HIP_GENERIC=1 times the flat-vector kernel on the same batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbstab_amd import hip_api
from tools import fixtures as fx
N, nx, nu, nc, B = (int(a) for a in sys.argv[1:6])
one = fx.random_ltv_mpc(np.random.default_rng(5), 64, N, nx, nu, nc)
p = fx.MpcProblem(N, nx, nu, nc)
p.arrays = {k: np.ascontiguousarray(np.tile(a, ((B + 63) // 64, 1))[:B]) for k, a in one.arrays.items()}
dev = torch.device("cuda:0")
s = hip_api.FBstabMpcBatch(N, nx, nu, nc, max_batch=B)
data = {k: torch.from_numpy(a).to(dev) for k, a in p.arrays.items()}
mk = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
for rep in range(2):
    z, l, v, y = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
    out = hip_api.out_to_numpy(s.Solve(data, z, l, v, y))
ms = s.last_kernel_ms()
print(f"{s.kernel_name()} shape=({N},{nx},{nu},{nc}) batch={B} kernel_ms={ms:.2f} QP/s={B / (ms * 1e-3):.0f} "
      f"eflags={np.bincount(out['eflag']).tolist()} newton mean={out['newton_iters'].mean():.1f} zsum={float(z.sum()):.9e}")
