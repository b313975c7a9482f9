#!/usr/bin/env python3
"""Developer tool: solve synthetic MPC ids [first, first+n) on the GPU and list
the instances whose exit flag is not SUCCESS, next to the oracle's outcome."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbstab_amd import hip_api
from tools import fixtures as fx
from oracle.oracle_py import Oracle
first, n = int(sys.argv[1]), int(sys.argv[2])
p = fx.synthetic_mpc_batch(n, first_id=first)
dev = torch.device("cuda:0")
s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=n)
data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
mk = lambda m: torch.zeros((n, m), dtype=torch.float64, device=dev)
z, l, v, y = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
out = hip_api.out_to_numpy(s.Solve(data, z, l, v, y))
bad = np.nonzero(out["eflag"] != 0)[0]
print("non-success:", len(bad), "flags", np.bincount(out["eflag"], minlength=7))
orc = Oracle(False)
for b in bad[:10]:
    q = fx.synthetic_mpc_batch(1, first_id=first + int(b))
    o = orc.solve_mpc(q)[4]
    print("id", first + int(b), "gpu", out[b], "oracle", o[0])
