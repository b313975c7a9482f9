#!/usr/bin/env python3
"""Developer study (GPU box): one shape of tools/fuzz_shapes.py's stream looked at closely - the QPs whose
counts differ from the oracle's: device (record kernel and flat-vector kernel) against the oracle, and the
first Newton step of both against the Newton system in extended precision (numpy longdouble).
usage: tools/fuzz_case.py <seed> <index of the shape in the stream> [all]   (all: probe every QP of the shape)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbstab_amd import hip_api
from tools import fixtures as fx
from tests import helpers as H
from oracle.oracle_py import Oracle, default_options
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
orc = Oracle(False)
for it in range(want + 1):
    nx = int(rng.integers(1, 27)); nu = int(rng.integers(1, 10)); nc = int(rng.integers(1, 34)); N = int(rng.integers(1, 13))
    B = int(rng.integers(1, 14))
    o = default_options()
    if rng.random() < 0.3:
        o = default_options(max_linesearch_iters=int(rng.integers(1, 12)), nonmonotone_linesearch=int(rng.random() < 0.5))
    p = fx.random_ltv_mpc(rng, B, N, nx, nu, nc)
print("shape", (N, nx, nu, nc), "B", B, "options: linesearch", o.max_linesearch_iters, "nonmonotone", o.nonmonotone_linesearch)
h = hip_api.Options()
for name, _ in h._fields_:
    setattr(h, name, getattr(o, name))
c = orc.solve_mpc(p, opts=o, nthreads=1)
oc = c[4]
res = {}
for gen in ("0", "1"):
    os.environ["FBSTAB_HIP_GENERIC"] = gen
    s = hip_api.FBstabMpcBatch(N, nx, nu, nc, max_batch=B)
    s.UpdateOptions(h)
    z = np.zeros((B, p.nz)); l = np.zeros((B, p.nl)); v = np.zeros((B, p.nv)); y = np.zeros((B, p.nv))
    out = s.Solve({k: np.ascontiguousarray(a) for k, a in p.arrays.items()}, z, l, v, y)
    res[gen] = (s.kernel_name(), out, z.copy())
    print(f"{s.kernel_name():34s} prox {out['prox_iters'].tolist()} newton {out['newton_iters'].tolist()}")
    print(" " * 34, "residual", [f"{r:.3e}" for r in out["residual"]])
    if gen == "0":
        rec = s
    else:
        s.close()
print(f"{'oracle':34s} prox {oc['prox_iters'].tolist()} newton {oc['newton_iters'].tolist()}")
print(" " * 34, "residual", [f"{r:.3e}" for r in oc["residual"]])
tol = [o.abs_tol + o.rel_tol * (1.0 + np.sqrt(sum(np.sum(p.arrays[k][q] ** 2) for k in ("q", "r", "c", "d", "x0")))) for q in range(B)]
print("stopping tolerance per QP (abs_tol + rel_tol (1 + |(f,h,b)|)):", [f"{t:.3e}" for t in tol])
bad = [q for q in range(B) if res["0"][1]["prox_iters"][q] != oc["prox_iters"][q] or res["0"][1]["newton_iters"][q] != oc["newton_iters"][q]]
LD = np.longdouble
if len(sys.argv) > 3 and sys.argv[3] == "all":
    bad = list(range(min(B, 4)))
for q in bad:
    print(f"--- QP {q}: first Newton step from the cold start (x = 0, xbar = 0, sigma = {o.sigma0:g})")
    one = fx.MpcProblem(N, nx, nu, nc, {k: np.ascontiguousarray(a[q:q + 1]) for k, a in p.arrays.items()})
    Hm, f, G, hh, A, b = H.mpc_explicit(p, q)
    zero = lambda n: np.zeros(n)
    data = {k: a[q] for k, a in p.arrays.items()}
    os.environ["FBSTAB_HIP_GENERIC"] = os.environ.get("FUZZ_CASE_PROBE_GENERIC", "0")  # 1: probe the flat-vector kernel
    s1 = hip_api.FBstabMpcBatch(N, nx, nu, nc, max_batch=1)
    print("    probe on", s1.kernel_name())
    s1.UpdateOptions(h)
    g = s1.debug_newton(data, zero(p.nz), zero(p.nl), zero(p.nv), zero(p.nz), zero(p.nl), zero(p.nv))
    s1.close()
    # the system in extended precision: x = 0 -> y = b, rz = f, rl = h, rv = phi(b, 0)
    sig, al = LD(o.sigma0), LD(o.alpha)
    yv, vv = b.astype(LD), np.zeros(p.nv, LD)
    rr = np.sqrt(yv * yv + vv * vv)
    gam = np.where(rr < 1e-13, al * (1 - 1 / np.sqrt(LD(2))), al * (1 - yv / np.where(rr > 0, rr, 1)))
    mu = np.where(rr < 1e-13, al * (1 - 1 / np.sqrt(LD(2))), al * (1 - vv / np.where(rr > 0, rr, 1)))
    pos = (yv > 0) & (vv > 0)
    gam = np.where(pos, gam + (1 - al) * vv, gam); mu = np.where(pos, mu + (1 - al) * yv, mu)
    mus = mu + sig * gam
    phi = al * (yv + vv - rr) + (1 - al) * np.maximum(yv, 0) * np.maximum(vv, 0)
    rz, rl, rv = -f.astype(LD), -hh.astype(LD), -phi
    def sysres(dz, dl, dv):
        dz, dl, dv = dz.astype(LD), dl.astype(LD), dv.astype(LD)
        e1 = Hm.astype(LD) @ dz + sig * dz + G.T.astype(LD) @ dl + A.T.astype(LD) @ dv - rz
        e2 = -G.astype(LD) @ dz + sig * dl - rl
        e3 = -gam * (A.astype(LD) @ dz) + mus * dv - rv
        return [float(np.abs(e).max()) for e in (e1, e2, e3)]
    print("    device step: |V dx - r| by block (z, l, v):", ["%.2e" % e for e in sysres(g["dz"], g["dl"], g["dv"])],
          " |dz| %.3e |dl| %.3e" % (np.abs(g["dz"]).max(), np.abs(g["dl"]).max()))
    pr = orc.probe(one, zero(p.nz), zero(p.nl), zero(p.nv), zero(p.nz), zero(p.nl), zero(p.nv), o.sigma0, o.alpha,
                   r=-np.concatenate([f, hh, np.asarray(phi, dtype=np.float64)]), want_dx=True)
    dx = pr["dx"]
    odz, odl, odv = dx[:p.nz], dx[p.nz:p.nz + p.nl], dx[p.nz + p.nl:p.nz + p.nl + p.nv]
    print("    oracle step: |V dx - r| by block (z, l, v):", ["%.2e" % e for e in sysres(odz, odl, odv)],
          " |dz| %.3e |dl| %.3e" % (np.abs(odz).max(), np.abs(odl).max()))
    print("    device - oracle: |ddz| %.3e  |ddl| %.3e  |ddv| %.3e" % (np.abs(g["dz"] - odz).max(), np.abs(g["dl"] - odl).max(), np.abs(g["dv"] - odv).max()))
