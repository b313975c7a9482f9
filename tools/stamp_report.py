#!/usr/bin/env python3
"""Developer tool: per-phase cycle shares of the MPC kernel from a -DFB_STAMP
build (FBSTAB_HIP_LIB=tools/_build/var_stamp.so; per-phase lines need -DFB_STAMP, the
wave-level shares -DFB_CLOCKSTAMP).  argv: batch [wgs_per_cu]."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from fbstab_amd import hip_api
from tools import fixtures as fx  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = fx.synthetic_mpc_batch(B)
dev = torch.device("cuda:0")
s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=max(B, 8192))
data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
mk = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
lib = hip_api.load_library()
st = (C.c_ulonglong * 32)()
for rep in range(2):
    z, l, v, y = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
    lib.fbstab_hip_debug_stamps(st, 1)
    out = hip_api.out_to_numpy(s.Solve(data, z, l, v, y))
    ms = s.last_kernel_ms()
lib.fbstab_hip_debug_stamps(st, 1)
newton = max(int(st[31]) or int(st[27]), 1)   # wave-level Newton steps (st[31]: an older build's count by row 0)
stages = newton * 31
if st[30]:
    print("backtracking trials per newton step (row 0):", st[30] / newton)
names = {0: "fwd loads+pfb", 1: "K build", 2: "rhs/h", 3: "chol16", 4: "tri_inv16", 5: "transpose+t+stores",
         6: "AB load + W", 7: "WW'", 8: "chol12+T+Pinv", 9: "bwd solve", 10: "bwd post+trial",
         16: "loop top", 17: "newton_step total", 18: "linesearch"}
print(f"batch={B} kernel_ms={ms:.2f} newton_total={newton} q={s.query()}")
if any(st[k] for k in names):  # (a -DFB_CLOCKSTAMP build carries the wave-level counters only)
    for k in sorted(names):
        per = st[k] / stages if k < 16 else st[k] / newton
        unit = "cyc/stage" if k < 16 else "cyc/newton-iter"
        print(f"   [{k:2d}] {names[k]:22s} {per:12.0f} {unit}")

if st[27]:
    useful = int(out["newton_iters"].sum())
    print(f"wave-level calls: newton_step {st[27]} (useful row-steps {useful}, packing {useful / (4.0 * st[27]):.2f}), "
          f"close_subproblem {st[26]}, open_prox {st[25]}")
if sum(st[8:16]):
    h = [int(x) for x in st[8:16]]
    print("   backtracking depth histogram (row level, 0 = full step ... 7+):", h, f"mean {sum(i * x for i, x in enumerate(h)) / max(sum(h), 1):.2f}")
if st[20]:
    tot = float(st[28])
    for k, nm, cnt in ((20, "newton_step", st[27]), (21, "close_subproblem", st[26]), (22, "open_prox", st[25]),
                       (23, "norms_at_multi", st[24]), (19, "load_guess", 0)):
        per = f"{st[k] / cnt:10.0f} cyc/call x {cnt}" if cnt else ""
        print(f"   wave cycles in {nm:18s} {100.0 * st[k] / tot:5.1f} %  {per}")
if st[29]:
    print(f"mean shader clock over the wavefronts' lifetimes: {st[28] / (st[29] * 0.01):.0f} MHz "
          f"(sum of wave lifetimes {st[29] * 1e-5:.1f} ms)")

if (st[4] or st[3]) and not st[9]:  # (slots 0..5 are phase laps in a -DFB_STAMP build)
    tot = float(st[28])
    print(f"migration: invitations {st[4]}, solves taken over {st[3]}, sleep trips {st[2]}; wave cycles: idle-row block "
          f"{100.0 * st[0] / tot:.1f} %, owner block {100.0 * st[1] / tot:.1f} %, sleeping {100.0 * st[5] / tot:.1f} %")
