#!/usr/bin/env python3
"""Developer tool (GPU box): kernel time of a dense shape, one launch at a time.
argv: nz nl nv [batch]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbstab_amd import hip_api
from tools import fixtures as fx
nz, nl, nv = (int(a) for a in sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
p = fx.synthetic_dense_batch(B, nz, nl, nv)
dev = torch.device("cuda:0")
s = hip_api.FBstabDenseBatch(nz, nl, nv, max_batch=B)
data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
mk = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
for rep in range(3):
    z, l, v, y = mk(nz), mk(nl), mk(nv), mk(nv)
    out = hip_api.out_to_numpy(s.Solve(data, z, l, v, y))
    ms = s.last_kernel_ms()
print(f"dense ({nz},{nl},{nv}) batch={B} threads={s.query()['threads']} kernel_ms={ms:.3f} QP/s={B / (ms * 1e-3):.0f} "
      f"ok={(out['eflag'] == 0).all()} newton mean={out['newton_iters'].mean():.2f}")
