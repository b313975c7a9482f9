#!/usr/bin/env python3
"""Developer tool: QP/s of the BASELINE plant with two-sided bounds on all sixteen stage
variables (32 constraint rows per stage: the <12,4,32> record instance), one launch at a
time.  argv: batch.  FBSTAB_HIP_GENERIC=1 times the flat-vector kernel on the same batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbstab_amd import hip_api
from tools import fixtures as fx
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = fx.boxed_mpc_batch(B)
dev = torch.device("cuda:0")
s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=B)
data = {k: torch.from_numpy(a).to(dev) for k, a in p.arrays.items()}
mk = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
for rep in range(2):
    z, l, v, y = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
    out = hip_api.out_to_numpy(s.Solve(data, z, l, v, y))
ms = s.last_kernel_ms()
print(f"{s.kernel_name()} batch={B} kernel_ms={ms:.2f} QP/s={B / (ms * 1e-3):.0f} eflags={np.bincount(out['eflag']).tolist()} "
      f"newton mean={out['newton_iters'].mean():.1f} max={out['newton_iters'].max()}")
