"""Shared helpers for the parity tests."""
import os

import numpy as np

from tools import fixtures as fx


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


# Variant builds of the product sources the tests load beside the product library (`with
# hip_api.library(VARIANT_LIBS[name])`); built by `make -C fbstab_amd/csrc <name>` into tests/_build/:
#   "pattern"  every automatic variable initialised to a bit pattern (-ftrivial-auto-var-init=pattern)
#              and the guard s_nop of every fused broadcast-FMA kept: a read of a value the code never
#              set gives the same garbage in every build instead of whatever the optimiser made of it
VARIANT_LIBS = {"pattern": os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libfbstab_hip_pattern.so")}


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


def newton_system_residual(p, q, step, x=None, xbar=None, sigma=1e-8, alpha=0.95):
    """|V dx - r| of the Newton system V(x, xbar, sigma) dx = -R(x, xbar, sigma) (abstract_components.h:276-288,
    full_residual.cc:49-74) for QP ``q`` of the MpcProblem ``p`` at x = (z, l, v) (default: zeros), evaluated in
    numpy longdouble (64-bit mantissa).  step: dict with dz, dl, dv.  Returns ([max |.| of the z, l, v block
    rows], 2-norm over all rows)."""
    LD = np.longdouble
    Hm, f, G, hh, A, b = (m.astype(LD) for m in mpc_explicit(p, q))
    zero = lambda n: np.zeros(n, LD)
    z, l, v = (t.astype(LD) for t in x) if x is not None else (zero(p.nz), zero(p.nl), zero(p.nv))
    zb, lb, vb = (t.astype(LD) for t in xbar) if xbar is not None else (z, l, v)
    sig, al = LD(sigma), LD(alpha)
    ys = (b - A @ z) + sig * (v - vb)
    rr = np.sqrt(ys * ys + v * v)
    c0 = al * (1 - 1 / np.sqrt(LD(2)))
    safe = np.where(rr > 0, rr, 1)
    gam = np.where(rr < 1e-13, c0, al * (1 - ys / safe))
    mu = np.where(rr < 1e-13, c0, al * (1 - v / safe))
    pos = (rr >= 1e-13) & (ys > 0) & (v > 0)
    gam = np.where(pos, gam + (1 - al) * v, gam)
    mu = np.where(pos, mu + (1 - al) * ys, mu)
    mus = mu + sig * gam
    phi = al * (ys + v - rr) + (1 - al) * np.maximum(ys, 0) * np.maximum(v, 0)
    rz = -(Hm @ z + f + G.T @ l + A.T @ v + sig * (z - zb))
    rl = -(hh - G @ z + sig * (l - lb))
    dz, dl, dv = (np.asarray(step[k]).astype(LD) for k in ("dz", "dl", "dv"))
    e1 = Hm @ dz + sig * dz + G.T @ dl + A.T @ dv - rz
    e2 = -G @ dz + sig * dl - rl
    e3 = -gam * (A @ dz) + mus * dv + phi
    blocks = [float(np.abs(e).max()) for e in (e1, e2, e3)]
    return blocks, float(np.sqrt((e1 * e1).sum() + (e2 * e2).sum() + (e3 * e3).sum()))


def fuzz_stream_shape(seed, index, family="random"):
    """Shape number ``index`` (from 0) of tools/fuzz_shapes.py's stream for ``seed``: (problem, options).
    ``family``: "random" (the default stream), "bounds" or "sparse" (the tool's options of those names)."""
    from oracle.oracle_py import default_options
    gen = {"random": fx.random_ltv_mpc, "bounds": fx.random_ltv_mpc_bounds, "sparse": fx.random_ltv_mpc_sparse_rows}[family]
    rng = np.random.default_rng(seed)
    for it in range(index + 1):
        nx = int(rng.integers(1, 27)); nu = int(rng.integers(1, 10)); nc = int(rng.integers(1, 34)); N = int(rng.integers(1, 13))
        B = int(rng.integers(1, 14))
        o = default_options()
        if rng.random() < 0.3:
            o = default_options(max_linesearch_iters=int(rng.integers(1, 12)), nonmonotone_linesearch=int(rng.random() < 0.5))
        p = gen(rng, B, N, nx, nu, nc)
    return p, o


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


# -- the reference's per-iteration display (fbstab_algorithm-impl.h:411-541) ----
EXIT_MESSAGES = {0: " Success\n", 1: " Divergence\n", 2: " Iteration limit exceeded\n",
                 3: " Primal Infeasibility\n", 4: " Dual Infeasibility\n",
                 5: " Primal-Dual Infeasibility\n"}


def format_display(records, level, out, opts):
    """Text of display level ``level`` (2 = ITER, 3 = ITER_DETAILED) from trace
    records ``(n, 8)`` = kind, i0, i1, v0..v4 (fbstab_trace_record_t) and the
    SolverOut record ``out``; the wall time is written as ``<t>``."""
    s = ""
    if level == 2:
        s += "%12s  %12s  %12s  %12s  %12s  %12s  %12s\n" % (
            "prox iter", "newton iters", "|rz|", "|rl|", "|rv|", "Inner res", "Inner tol")
    for r in records:
        kind, i0, i1, v = int(r[0]), int(r[1]), int(r[2]), r[3:]
        if kind == 1 and level == 2:
            s += "%12d  %12d  %12.4e  %12.4e  %12.4e  %12.4e  %12.4e\n" % (i0, i1, *v[:5])
        elif kind == 2 and level == 3:
            s += "Begin Prox Iter: %d, Total Newton Iters: %d, Residual: %6.4e\n" % (i0, i1, v[0])
            s += "%10s  %10s  %10s  %10s  %10s\n" % ("Iter", "Step Size", "|rz|", "|rl|", "|rv|")
        elif kind == 3 and level == 3:
            s += "%10d  %10e  %10e  %10e  %10e\n" % (i0, *v[:4])
        elif kind == 4 and level == 3:
            s += "Exiting inner loop. Inner residual: %6.4e, Inner tolerance: %6.4e\n" % (v[0], v[1])
        elif kind == 5:
            s += "\nOptimization completed!  Exit code:" + EXIT_MESSAGES[int(out["eflag"])]
            s += "Time elapsed: <t> ms (-1.0 indicates timing disabled)\n"
            s += "Proximal iterations: %d out of %d\n" % (out["prox_iters"], opts.max_prox_iters)
            s += "Newton iterations: %d out of %d\n" % (out["newton_iters"], opts.max_newton_iters)
            s += "%10s  %10s  %10s  %10s\n" % ("|rz|", "|rl|", "|rv|", "Tolerance")
            s += "%10.4e  %10.4e  %10.4e  %10.4e\n\n" % tuple(v[:4])
    return s


def normalise_time(text):
    import re
    return re.sub(r"Time elapsed: \S+ ms", "Time elapsed: <t> ms", text)


def display_texts_agree(a, b, rtol=5e-4, atol=1e-7):
    """Token-wise comparison of two display texts: words must be equal, numbers
    agree to ``atol + rtol*|b|`` (the display prints 5 significant digits; atol
    is 1e-7 of the largest residual of the solve), and numbers below 1e4*atol -
    residuals left behind by a converged Newton iteration, which squares the
    rounding differences of the step before - within a factor of two."""
    import re
    ta, tb = re.split(r"[\s,]+", a.strip()), re.split(r"[\s,]+", b.strip())
    if len(ta) != len(tb):
        return False, "token count %d vs %d" % (len(ta), len(tb))
    for x, y in zip(ta, tb):
        try:
            fx_, fy = float(x), float(y)
        except ValueError:
            if x != y:
                return False, "%r vs %r" % (x, y)
            continue
        err = abs(fx_ - fy)
        if err > atol + rtol * abs(fy) and not (abs(fy) < 1e4 * atol and err <= 0.5 * max(abs(fx_), abs(fy))):
            return False, "%r vs %r" % (x, y)
    return True, ""
