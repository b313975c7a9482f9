#!/usr/bin/env python3
"""Developer tool (GPU box): accuracy of ONE Newton solve of the dense kernel named by
FBSTAB_HIP_LIB / FBSTAB_HIP_DENSE_THREADS near the solution of degenerate QPs, against the
exact step of the same Newton system (mpmath, 40 digits): relative residual of the
elimination-free system and relative error of (dz, dl, dv).
argv: nz nl nv first_id [number of QPs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpmath as mp
from tools import fixtures as fx
from fbstab_amd import hip_api as hip
from oracle.oracle_py import Oracle, default_options
from tests import helpers as H
from tests.test_gpu_components import _pfb_gradient, _pfb
mp.mp.dps = 40
nz, nl, nv, fid = (int(a) for a in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
p = fx.synthetic_dense_batch(B, nz, nl, nv, first_id=fid)
o = default_options()
orc = Oracle()
zs, ls, vs, ys, out = orc.solve_dense(p, opts=o, nthreads=orc.num_threads())
sigma, alpha = o.sigma0, o.alpha
s = hip.FBstabDenseBatch(nz, nl, nv, max_batch=1)
s.UpdateOptions(hip.DefaultOptions())
rng = np.random.default_rng(7)
print("kernel threads", s.query()["threads"], "lib", os.environ.get("FBSTAB_HIP_LIB", "default"))
for i in range(B):
    data = {k: a[i] for k, a in p.arrays.items()}
    Hm, f, G, h, A, b = H.dense_explicit(p, i)
    zb, lb, vb = zs[i], ls[i], vs[i]
    z = zb + 1e-3 * rng.standard_normal(nz); l = lb + 1e-3 * rng.standard_normal(nl)
    v = np.maximum(vb + 1e-3 * rng.standard_normal(nv) * (vb > 0), 0.0)
    g = s.debug_newton(data, z, l, v, zb, lb, vb)
    assert g["ok"]
    y = b - A @ z
    ysv = y + sigma * (v - vb)
    gam, mus = _pfb_gradient(ysv, v, alpha, sigma)
    r1 = -(Hm @ z + f + G.T @ l + A.T @ v + sigma * (z - zb))
    r2 = (h - G @ z) + sigma * (l - lb)
    r3 = -_pfb(ysv, v, alpha)
    n = nz + nl + nv
    K = mp.zeros(n, n); rhs = mp.zeros(n, 1)
    for a_ in range(nz):
        for c_ in range(nz):
            K[a_, c_] = mp.mpf(Hm[a_, c_])
        K[a_, a_] += mp.mpf(sigma)
        for q in range(nl):
            K[a_, nz + q] = mp.mpf(G[q, a_])
        for k in range(nv):
            K[a_, nz + nl + k] = mp.mpf(A[k, a_])
        rhs[a_] = mp.mpf(r1[a_])
    for q in range(nl):
        for c_ in range(nz):
            K[nz + q, c_] = mp.mpf(G[q, c_])
        K[nz + q, nz + q] = -mp.mpf(sigma)
        rhs[nz + q] = mp.mpf(r2[q])
    for k in range(nv):
        for c_ in range(nz):
            K[nz + nl + k, c_] = -mp.mpf(gam[k]) * mp.mpf(A[k, c_])
        K[nz + nl + k, nz + nl + k] = mp.mpf(mus[k])
        rhs[nz + nl + k] = mp.mpf(r3[k])
    xe = mp.lu_solve(K, rhs)
    xd = mp.matrix([mp.mpf(t) for t in np.concatenate([g["dz"], g["dl"], g["dv"]])])
    res = K * xd - rhs
    err = xd - xe
    nrm = lambda m, a_, b_: mp.sqrt(sum(m[t] ** 2 for t in range(a_, b_)))
    act = int((vb > 1e-7).sum())
    print(f"QP {i}: active {act} + nl {nl} vs nz {nz}; residual/|rhs| {float(nrm(res, 0, n) / nrm(rhs, 0, n)):.2e}; "
          f"relative error dz {float(nrm(err, 0, nz) / nrm(xe, 0, nz)):.2e}"
          + (f" dl {float(nrm(err, nz, nz + nl) / nrm(xe, nz, nz + nl)):.2e}" if nl else "")
          + f" dv {float(nrm(err, nz + nl, n) / nrm(xe, nz + nl, n)):.2e}; |step| {float(nrm(xe, 0, n)):.2e}")
s.close()
