#!/usr/bin/env python3
"""Developer tool: random dense QP shapes (one-wavefront, four-wavefront and global-K
kernels) against the oracle.  argv: number of shapes [seed [largest nz + nl]]
(64 as the third argument keeps every shape on the one-wavefront kernel)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbstab_amd import hip_api
from tools import fixtures as fx
from oracle.oracle_py import Oracle, default_options
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 0
orc = Oracle(False)
bad = 0
ndiff = nqp = 0
for it in range(n):
    nz = int(rng.integers(1, 160)); nl = int(rng.integers(0, min(nz, 24) + 1)); nv = int(rng.integers(1, 240))
    if kmax:
        nz = int(rng.integers(1, kmax + 1)); nl = int(rng.integers(0, min(nz, 24, kmax - nz) + 1))
    B = int(rng.integers(1, 10))
    o = default_options()
    p = fx.synthetic_dense_batch(B, nz, nl, nv, first_id=int(rng.integers(0, 1 << 20)))
    s = hip_api.FBstabDenseBatch(nz, nl, nv, max_batch=B)
    z = np.zeros((B, nz)); l = np.zeros((B, nl)); v = np.zeros((B, nv)); y = np.zeros((B, nv))
    out = s.Solve({k: np.ascontiguousarray(a) for k, a in p.arrays.items()}, z, l, v, y)
    q = s.query(); s.close()
    c = orc.solve_dense(p, opts=o, nthreads=orc.num_threads())
    oc = c[4]
    dn = np.abs(out["newton_iters"].astype(int) - oc["newton_iters"].astype(int))
    okf = np.array_equal(out["eflag"], oc["eflag"]) and np.array_equal(out["prox_iters"], oc["prox_iters"])
    good = oc["eflag"] == 0
    dz = float(np.abs(z - c[0])[good].max()) if good.any() else 0.0
    flag = "" if (okf and dn.max() <= 2 and dz < 1e-4) else "  <-- CHECK"
    bad += flag != ""
    ndiff += int((dn != 0).sum()); nqp += B
    if flag:
        print("   device: eflag", out["eflag"].tolist(), "prox", out["prox_iters"].tolist(), "newton", out["newton_iters"].tolist(),
              "residual", [f"{r:.2e}" for r in out["residual"]])
        print("   oracle: eflag", oc["eflag"].tolist(), "prox", oc["prox_iters"].tolist(), "newton", oc["newton_iters"].tolist(),
              "residual", [f"{r:.2e}" for r in oc["residual"]])
    print(f"({nz},{nl},{nv}) B={B} threads={q['threads']} lds={q['lds_bytes']} flags_equal={okf} dnewton_max={dn.max()} nonzero={int((dn != 0).sum())} dz={dz:.2e}{flag}")
print("shapes to check:", bad, " QPs:", nqp, " QPs whose Newton count differs from the oracle's:", ndiff)
