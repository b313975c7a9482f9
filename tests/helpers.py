"""Shared helpers for the parity tests."""
import numpy as np

from fbstab_amd import fixtures as fx


def dense_from_kat(k):
    nz = len(k["f"])
    G = np.asarray(k.get("G", []), dtype=np.float64).reshape(-1, nz)
    return fx.dense_problem(k["H"], k["f"], G, k.get("h", []), k["A"], k["b"])


def mpc_from_kat(k):
    g = fx.OcpGenerator()
    getattr(g, k["name"])(k["N"])
    return g.GetFBstabInput()


def mpc_component_fixture(c):
    """N=2 double integrator with the FULL E at every stage
    (mpc_component_unit_tests.h:37-93)."""
    N = c["N"]
    col = lambda v: np.asarray(v, dtype=np.float64).reshape(-1, 1)
    m = lambda v: np.atleast_2d(np.asarray(v, dtype=np.float64))
    p = fx.MpcProblem(N, 2, 1, 6)
    seq = dict(Q=[m(c["Q"])] * (N + 1), R=[m(c["R"])] * (N + 1),
               S=[m(c["S"])] * (N + 1), q=[col(c["q"])] * (N + 1),
               r=[col(c["r"])] * (N + 1), A=[m(c["A"])] * N, B=[m(c["B"])] * N,
               c=[col(c["c"])] * N, E=[m(c["E"])] * (N + 1),
               L=[m(c["L"])] * (N + 1), d=[col(c["d"])] * (N + 1))
    p.arrays = {k: fx._colmajor(v).reshape(1, -1) for k, v in seq.items()}
    p.arrays["x0"] = np.asarray(c["x0"], dtype=np.float64).reshape(1, -1)
    return p


def mpc_explicit(p, b=0):
    """Explicit (H, f, G, h, A, b) of QP ``b`` of an MpcProblem, built from the
    definitions in fbstab_mpc.h:22-49 / mpc_data.cc (G=[-I; A B -I; ...],
    h=-(x0,c), b=-d)."""
    N, nx, nu, nc = p.sizes()
    ns = nx + nu
    a = {k: v[b] for k, v in p.arrays.items()}
    mat = lambda key, k, r, c: a[key][k * r * c:(k + 1) * r * c].reshape(c, r).T
    H = np.zeros((p.nz, p.nz))
    A = np.zeros((p.nv, p.nz))
    G = np.zeros((p.nl, p.nz))
    f = np.zeros(p.nz)
    h = np.zeros(p.nl)
    bb = np.zeros(p.nv)
    for i in range(N + 1):
        o = i * ns
        H[o:o + nx, o:o + nx] = mat("Q", i, nx, nx)
        H[o + nx:o + ns, o:o + nx] = mat("S", i, nu, nx)
        H[o:o + nx, o + nx:o + ns] = mat("S", i, nu, nx).T
        H[o + nx:o + ns, o + nx:o + ns] = mat("R", i, nu, nu)
        f[o:o + nx] = a["q"][i * nx:(i + 1) * nx]
        f[o + nx:o + ns] = a["r"][i * nu:(i + 1) * nu]
        A[i * nc:(i + 1) * nc, o:o + nx] = mat("E", i, nc, nx)
        A[i * nc:(i + 1) * nc, o + nx:o + ns] = mat("L", i, nc, nu)
        bb[i * nc:(i + 1) * nc] = -a["d"][i * nc:(i + 1) * nc]
        G[i * nx:(i + 1) * nx, o:o + nx] = -np.eye(nx)
        if i == 0:
            h[:nx] = -a["x0"]
        else:
            G[i * nx:(i + 1) * nx, o - ns:o - ns + nx] = mat("A", i - 1, nx, nx)
            G[i * nx:(i + 1) * nx, o - ns + nx:o] = mat("B", i - 1, nx, nu)
            h[i * nx:(i + 1) * nx] = -a["c"][(i - 1) * nx:i * nx]
    return H, f, G, h, A, bb


def dense_explicit(p, b=0):
    a = {k: v[b] for k, v in p.arrays.items()}
    H = a["H"].reshape(p.nz, p.nz).T
    G = a["G"].reshape(p.nz, p.nl).T
    A = a["A"].reshape(p.nz, p.nv).T
    return H, a["f"], G, a["h"], A, a["b"]


def natural_residual_norm(H, f, G, h, A, b, z, l, v):
    """||(Hz+f+G'l+A'v, h-Gz, min(b-Az, v))||, the KKT measure the reference's
    tests use (fbstab_dense_unit_tests.cc:172-176)."""
    rz = H @ z + f + G.T @ l + A.T @ v
    rl = h - G @ z
    rv = np.minimum(b - A @ z, v)
    return np.sqrt(rz @ rz + rl @ rl + rv @ rv)
