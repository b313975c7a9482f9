#!/usr/bin/env python3
"""Developer tool: the receding sweep alone (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbstab_amd import hip_api
from tools import fixtures as fx
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 200
p = fx.synthetic_mpc_batch(T)
A, B = fx.quadrotor_model()
dev = torch.device("cuda:0")
s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=T)
data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
mk = lambda n: torch.zeros((T, n), dtype=torch.float64, device=dev)
z, l, v, y = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
torch.cuda.synchronize()
t0 = time.perf_counter()
r = s.RecedingSweep(data, z, l, v, y, A, B, S, retire=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"wall {dt*1e3:.1f} ms, per step {dt/S*1e3:.3f} ms; kernel ms first/median/sum {r['kernel_ms'][0]:.2f}/{np.median(r['kernel_ms']):.3f}/{r['kernel_ms'].sum():.1f}; retired {r['stats']['retired_total'][-1]}")
