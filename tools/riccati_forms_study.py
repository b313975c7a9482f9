#!/usr/bin/env python3
"""Developer study (CPU, numpy): which evaluation of the Riccati recursion's VECTOR part leaves which
residual of the Newton system (riccati_linear_solver.cc:212-341, abstract_components.h:276-288).

The matrix part is the same in every variant (K_i = H_i + inv(Pi_i) + C'Gamma C, Lc = chol K_i,
W = [A B] inv(Lc)', Pi_{i+1} = sigma I + W W', inv(Pi) formed explicitly - as the reference does,
riccati_linear_solver.cc:143-145).  The vector part differs:

  hform   how h_i = inv(Pi_i) theta_i is applied:  pinv = product with the formed inverse T'T,
          ttt  = T'(T theta) with T = inv(L) formed explicitly,
          lt   = T'(L \\ theta): forward substitution with L, then the product with T',
          sub  = two substitutions with L (the reference)
  tform   inv = products with an explicitly inverted Lc;  sub = substitutions with Lc (the reference)
  dlform  pinv / ttt / lt / sub as above for dl_i = -inv(Pi_i)(theta_i + dx_i);
          row  = dl_i from the state rows of the system's first block row (fb_mpc_r16.h form (b))
  refine  number of refinement sweeps with the same factors (residual formed in working precision)

Residuals |V dx - r| by block (z, l, v) are evaluated in numpy longdouble (64-bit mantissa).
usage: tools/riccati_forms_study.py [baseline|fuzz42] [nqp]"""
import os, sys
import numpy as np
from scipy.linalg import solve_triangular as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import fixtures as fx
from tests import helpers as H

LD = np.longdouble


def stage_mats(p, q):
    N, nx, nu, nc = p.sizes()
    a = {k: v[q] for k, v in p.arrays.items()}
    mat = lambda key, k, r, c: a[key][k * r * c:(k + 1) * r * c].reshape(c, r).T
    out = []
    for i in range(N + 1):
        Hs = np.block([[mat("Q", i, nx, nx), mat("S", i, nu, nx).T], [mat("S", i, nu, nx), mat("R", i, nu, nu)]])
        C = np.hstack([mat("E", i, nc, nx), mat("L", i, nc, nu)])
        AB = np.hstack([mat("A", i, nx, nx), mat("B", i, nx, nu)]) if i < N else np.zeros((nx, nx + nu))
        out.append((Hs, C, AB))
    return out


def tri_inv(L):
    return st(L, np.eye(L.shape[0]), lower=True)


def apply_pinv(form, L, T, Pinv, x):
    if form == "pinv":
        return Pinv @ x
    if form == "ttt":
        return T.T @ (T @ x)
    if form == "lt":
        return T.T @ st(L, x, lower=True)
    return st(L.T, st(L, x, lower=True), lower=False)


class Riccati:
    def __init__(self, p, q, gam, mus, sigma):
        self.N, self.nx, self.nu, self.nc = p.sizes()
        self.sig = sigma
        self.M = stage_mats(p, q)
        N, nx, nu, nc = p.sizes()
        self.gam = gam.reshape(N + 1, nc)
        self.mus = mus.reshape(N + 1, nc)
        self.L = [np.sqrt(sigma) * np.eye(nx)]
        self.T, self.Pinv, self.Lc, self.Xc, self.W = [], [], [], [], []
        for i in range(N + 1):
            Hs, C, AB = self.M[i]
            T = tri_inv(self.L[i])
            Pinv = T.T @ T
            K = Hs + sigma * np.eye(nx + nu) + C.T @ ((self.gam[i] / self.mus[i])[:, None] * C)
            K[:nx, :nx] += Pinv
            Lc = np.linalg.cholesky(K)
            W = st(Lc, AB.T, lower=True).T
            self.T.append(T); self.Pinv.append(Pinv); self.Lc.append(Lc); self.Xc.append(tri_inv(Lc)); self.W.append(W)
            if i < N:
                self.L.append(np.linalg.cholesky(sigma * np.eye(nx) + W @ W.T))

    def solve(self, bz, bl, bv, hform, tform, dlform):
        N, nx, nu, nc, sig = self.N, self.nx, self.nu, self.nc, self.sig
        ns = nx + nu
        bz = bz.reshape(N + 1, ns); bl = bl.reshape(N + 1, nx); bv = bv.reshape(N + 1, nc)
        r1 = np.stack([bz[i] - self.M[i][1].T @ (bv[i] / self.mus[i]) for i in range(N + 1)])
        r2 = -bl
        th = [None] * (N + 1); t = [None] * (N + 1)
        thp = np.zeros(nx)
        for i in range(N + 1):
            th[i] = thp + r2[i]
            g = r1[i].copy()
            g[:nx] -= apply_pinv(hform, self.L[i], self.T[i], self.Pinv[i], th[i])
            t[i] = self.Xc[i] @ g if tform[0] == "i" else st(self.Lc[i], g, lower=True)
            thp = -(self.W[i] @ t[i])
        dz = np.zeros((N + 1, ns)); dl = np.zeros((N + 1, nx)); dv = np.zeros((N + 1, nc))
        lp = np.zeros(nx)
        for i in range(N, -1, -1):
            Hs, C, AB = self.M[i]
            u = AB.T @ lp
            s = t[i] - (self.Xc[i] @ u if tform[1] == "i" else st(self.Lc[i], u, lower=True))
            dz[i] = self.Xc[i].T @ s if tform[2] == "i" else st(self.Lc[i].T, s, lower=False)
            dv[i] = (bv[i] + self.gam[i] * (C @ dz[i])) / self.mus[i]
            if dlform == "row":
                w = Hs @ dz[i] + u + C.T @ dv[i]
                dl[i] = (w + sig * dz[i])[:nx] - bz[i][:nx]
            else:
                dl[i] = -apply_pinv(dlform, self.L[i], self.T[i], self.Pinv[i], th[i] + dz[i][:nx])
            lp = dl[i]
        return dz.ravel(), dl.ravel(), dv.ravel()


def run(p, q, sigma=1e-8, alpha=0.95, x=None, label=""):
    Hm, f, G, hh, A, b = H.mpc_explicit(p, q)
    nz, nl, nv = p.nz, p.nl, p.nv
    if x is None:
        z, l, v = np.zeros(nz), np.zeros(nl), np.zeros(nv)
    else:
        z, l, v = x
    zb, lb, vb = z.copy(), l.copy(), v.copy()
    y = b - A @ z
    ys = y + sigma * (v - vb)
    rr = np.sqrt(ys * ys + v * v)
    small = rr < 1e-13
    g0 = np.where(small, alpha * (1 - 1 / np.sqrt(2)), alpha * (1 - ys / np.where(rr > 0, rr, 1)))
    g1 = np.where(small, alpha * (1 - 1 / np.sqrt(2)), alpha * (1 - v / np.where(rr > 0, rr, 1)))
    pos = (ys > 0) & (v > 0)
    g0 = np.where(pos, g0 + (1 - alpha) * v, g0); g1 = np.where(pos, g1 + (1 - alpha) * ys, g1)
    mus = g1 + sigma * g0
    phi = alpha * (ys + v - rr) + (1 - alpha) * np.maximum(ys, 0) * np.maximum(v, 0)
    bz = -(Hm @ z + f + G.T @ l + A.T @ v + sigma * (z - zb))
    bl = -(hh - G @ z + sigma * (l - lb))
    bv = -phi
    R = Riccati(p, q, g0, mus, sigma)

    def sysres(dz, dl, dv, full=False):
        dz, dl, dv = dz.astype(LD), dl.astype(LD), dv.astype(LD)
        e1 = Hm.astype(LD) @ dz + LD(sigma) * dz + G.T.astype(LD) @ dl + A.T.astype(LD) @ dv - bz.astype(LD)
        e2 = -G.astype(LD) @ dz + LD(sigma) * dl - bl.astype(LD)
        e3 = -g0.astype(LD) * (A.astype(LD) @ dz) + mus.astype(LD) * dv - bv.astype(LD)
        if full:
            return e1, e2, e3
        return [float(np.abs(e).max()) for e in (e1, e2, e3)]

    def resid64(dz, dl, dv):  # the residual as the device would form it: working precision
        e1 = bz - (Hm @ dz + sigma * dz + G.T @ dl + A.T @ dv)
        e2 = bl - (-G @ dz + sigma * dl)
        e3 = bv - (-g0 * (A @ dz) + mus * dv)
        return e1, e2, e3

    rows = []
    variants = [
        ("reference: sub/sub/sub", "sub", "sss", "sub", 0),
        ("device (a): pinv/inv/pinv", "pinv", "iii", "pinv", 0),
        ("device (b): pinv/inv/row", "pinv", "iii", "row", 0),
        ("(a) + 1 refinement", "pinv", "iii", "pinv", 1),
        ("(b) + 1 refinement", "pinv", "iii", "row", 1),
        ("ttt/inv/ttt", "ttt", "iii", "ttt", 0),
        ("ttt/inv/row", "ttt", "iii", "row", 0),
        ("lt/inv/lt", "lt", "iii", "lt", 0),
        ("lt/inv/row", "lt", "iii", "row", 0),
        ("sub/inv/sub", "sub", "iii", "sub", 0),
        ("sub/inv/row", "sub", "iii", "row", 0),
        ("lt/sub/lt", "lt", "sss", "lt", 0),
        ("pinv/sub/pinv", "pinv", "sss", "pinv", 0),
        ("pinv/sub/row", "pinv", "sss", "row", 0),
        ("ttt/sub/row", "ttt", "sss", "row", 0),
        ("ttt/sub/ttt", "ttt", "sss", "ttt", 0),
        ("pinv/sii/pinv", "pinv", "sii", "pinv", 0),
        ("pinv/isi/pinv", "pinv", "isi", "pinv", 0),
        ("pinv/iis/pinv", "pinv", "iis", "pinv", 0),
        ("pinv/ssi/pinv", "pinv", "ssi", "pinv", 0),
        ("pinv/sis/pinv", "pinv", "sis", "pinv", 0),
        ("pinv/iss/pinv", "pinv", "iss", "pinv", 0),
        ("pinv/sii/row", "pinv", "sii", "row", 0),
        ("pinv/sis/row", "pinv", "sis", "row", 0),
        ("pinv/iss/row", "pinv", "iss", "row", 0),
    ]
    for name, hf, tf, df, nref in variants:
        dz, dl, dv = R.solve(bz, bl, bv, hf, tf, df)
        for _ in range(nref):
            e1, e2, e3 = resid64(dz, dl, dv)
            ez, el, ev = R.solve(e1, e2, e3, hf, tf, df)
            dz, dl, dv = dz + ez, dl + el, dv + ev
        e = sysres(dz, dl, dv)
        tot = float(np.sqrt(sum((x.astype(LD) ** 2).sum() for x in sysres(dz, dl, dv, True))))
        rows.append((name, e, tot))
    print(f"{label} |dz| {np.abs(dz).max():.2e} |dl| {np.abs(dl).max():.2e}")
    for name, e, tot in rows:
        print(f"   {name:28s} z {e[0]:.1e}  l {e[1]:.1e}  v {e[2]:.1e}   ||.||_2 {tot:.1e}")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "baseline"
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    if which == "baseline":
        p = fx.synthetic_mpc_batch(nq)
        for q in range(nq):
            run(p, q, label=f"BASELINE qp {q}, cold start:")
        from oracle.oracle_py import Oracle
        c = Oracle(False).solve_mpc(p)
        for q in range(nq):
            # near the solution, the duals perturbed so that the step is not zero
            rng = np.random.default_rng(q)
            x = (c[0][q] * (1 + 1e-3 * rng.standard_normal(p.nz)), c[1][q] * (1 + 1e-3 * rng.standard_normal(p.nl)),
                 np.maximum(c[2][q] * (1 + 1e-3 * rng.standard_normal(p.nv)), 0))
            run(p, q, x=x, label=f"BASELINE qp {q}, near the solution:")
    else:
        from oracle.oracle_py import default_options
        rng = np.random.default_rng(42)
        for it in range(128):
            nx = int(rng.integers(1, 27)); nu = int(rng.integers(1, 10)); nc = int(rng.integers(1, 34)); N = int(rng.integers(1, 13))
            B = int(rng.integers(1, 14))
            if rng.random() < 0.3:
                int(rng.integers(1, 12)); rng.random()
            p = fx.random_ltv_mpc(rng, B, N, nx, nu, nc)
        print("shape", p.sizes(), "B", B)
        for q in range(min(B, nq)):
            run(p, q, label=f"fuzz seed 42 shape 128 qp {q}, cold start:")
