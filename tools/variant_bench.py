#!/usr/bin/env python3
"""Developer tool: time the MPC kernel of one build variant
(FBSTAB_HIP_LIB=<so>, FBSTAB_HIP_WGS_PER_CU=<n>) on the BASELINE workload and
print kernel ms, QPs/s and a checksum of the iteration counts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from fbstab_amd import hip_api
from tools import fixtures as fx  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
p = fx.synthetic_mpc_batch(B)
if os.environ.get("FB_WORKLOAD") == "ltv":  # the bench line's ltv_dense_rows block: per-stage matrices, dense rows
    p = fx.synthetic_mpc_ltv_batch(B)
same = os.environ.get("FB_SAME")
if same is not None:
    # every QP of the batch is a physical copy of instance `same` (no divergence
    # between the rows of a wavefront; same HBM traffic pattern)
    p1 = fx.synthetic_mpc_batch(1, first_id=int(same))
    for k in p.arrays:
        p.arrays[k] = np.ascontiguousarray(np.broadcast_to(p1.arrays[k], p.arrays[k].shape))
dev = torch.device("cuda:0")
s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=B)
data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
mk = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
ms = []
for r in range(reps + 1):
    z, l, v, y = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
    out = hip_api.out_to_numpy(s.Solve(data, z, l, v, y))
    ms.append(s.last_kernel_ms())
q = s.query()
print(f"{os.environ.get('FBSTAB_HIP_LIB', 'default'):>40s} wgs/cu={os.environ.get('FBSTAB_HIP_WGS_PER_CU', 'auto'):>4s} "
      f"wgs={q['workgroups']:5d} lds={q['lds_bytes']:6d} kernel_ms={min(ms[1:]):9.2f} "
      f"QP/s={B / (min(ms[1:]) * 1e-3):10.0f} ok={(out['eflag'] == 0).all()} "
      f"newton_sum={int(out['newton_iters'].sum())} zsum={float(z.abs().sum()):.9e}")
import ctypes as C
st = (C.c_ulonglong * 32)()
hip_api.load_library().fbstab_hip_debug_stamps(st, 1)
if any(st):
    tot = sum(st[:11])
    names = {0: "loads+pfb", 1: "K build", 2: "rhs/h", 3: "chol16", 4: "tri_inv16", 5: "transpose+t+stores",
             6: "AB load + W", 7: "WW'", 8: "chol12+T+Pinv", 9: "bwd stage", 10: "post stage",
             16: "norms(top)", 17: "newton_step total", 18: "linesearch+accept"}
    for k in sorted(names):
        print(f"   stamp[{k:2d}] {names[k]:22s} {st[k]:16d}  {100.0 * st[k] / max(tot, 1):6.2f}% of newton_step")
