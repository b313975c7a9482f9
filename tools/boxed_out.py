#!/usr/bin/env python3
"""Developer tool (GPU box): SolverOut records of a few QPs of the boxed BASELINE plant."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbstab_amd import hip_api
from tools import fixtures as fx
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = fx.boxed_mpc_batch(B)
s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=B)
z = np.zeros((B, p.nz)); l = np.zeros((B, p.nl)); v = np.zeros((B, p.nv)); y = np.zeros((B, p.nv))
out = s.Solve({k: np.ascontiguousarray(a) for k, a in p.arrays.items()}, z, l, v, y)
for k in ("eflag", "newton_iters", "prox_iters", "residual", "initial_residual"):
    print(k, out[k].tolist())
print("zsum", float(np.abs(z).sum()))
