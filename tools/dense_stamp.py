#!/usr/bin/env python3
"""Developer tool: where the dense kernel's wave cycles go (-DFB_CLOCKSTAMP build,
FBSTAB_HIP_LIB=<that .so>) on BASELINE config 2."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbstab_amd import hip_api
from tools import fixtures as fx
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nz, nl, nv = 50, 10, 100
p = fx.synthetic_dense_batch(B, nz, nl, nv)
dev = torch.device("cuda:0")
s = hip_api.FBstabDenseBatch(nz, nl, nv, max_batch=B)
data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
mk = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
lib = hip_api.load_library()
st = (C.c_ulonglong * 32)()
for rep in range(2):
    z, l, v, y = mk(nz), mk(nl), mk(nv), mk(nv)
    lib.fbstab_hip_debug_stamps(st, 1)
    out = hip_api.out_to_numpy(s.Solve(data, z, l, v, y))
    ms = s.last_kernel_ms()
lib.fbstab_hip_debug_stamps(st, 1)
tot = float(st[28]) or 1.0
print(f"dense batch={B} kernel_ms={ms:.2f} newton mean={out['newton_iters'].mean():.2f} prox mean={out['prox_iters'].mean():.2f}")
# (natural-order path, the default: 3 = the unrolled factorisation with the forward elimination,
# 1 = the backward sweep; the four LDL' labels describe the pivoted fallback)
names = {0: "  LDL': pivot search", 1: "  LDL': pick | natural order: backward sweep", 2: "  LDL': column to LDS, multiplier", 3: "  LDL': trailing update | natural order: factorisation + forward elimination",
         4: "  solve: forward loads issued", 5: "  solve: forward chain", 6: "  solve: / D, backward loads issued", 7: "  solve: backward chain",
         19: "  assembly: H into the accumulators", 20: "  assembly: MFMA loop", 21: "  assembly: tiles to rows", 22: "  assembly: G blocks",
         9: "load_guess", 10: "pfb gradients", 11: "K assembly + rhs", 12: "LDL' (natural order: factorisation and both sweeps)", 13: "solve (pivoted fallback only)", 14: "dv, A dz, W",
         15: "residual", 16: "feasibility", 17: "norms_at (line search, loop top)", 18: "accept"}
for k, nm in names.items():
    print(f"   {nm:34s} {100.0 * st[k] / tot:5.1f} %")
