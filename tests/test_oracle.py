"""Pins the CPU oracle against every known-answer vector the reference's own
tests hold for the hot path, and against the reference's
own FBstabAlgorithm<> loop (oracle/_ref).  CPU only."""
import numpy as np
import pytest

from oracle.oracle_py import default_options, reliable_options
from tests import helpers as H


# -- end-to-end: fbstab/test/fbstab_dense_unit_tests.cc ----------------------
@pytest.mark.parametrize("idx", range(5))
def test_dense_end_to_end_kats(oracle, kats, idx):
    k = kats["dense_end_to_end"][idx]
    p = H.dense_from_kat(k)
    z, l, v, y, out = oracle.solve_dense(p, opts=default_options(abs_tol=k["abs_tol"]))
    assert out["eflag"][0] == k["eflag"], k["name"]
    if "zopt" in k:
        np.testing.assert_allclose(z[0], k["zopt"], atol=k["tol"], rtol=0)
    if "vopt" in k:
        np.testing.assert_allclose(v[0], k["vopt"], atol=k["tol"], rtol=0)
    if "z0" in k:  # DegenerateQP
        assert abs(z[0, 0] - k["z0"]) <= k["tol"]
        assert k["z1_range"][0] <= z[0, 1] <= k["z1_range"][1]
        Hm, f, G, h, A, b = H.dense_explicit(p)
        r1 = Hm @ z[0] + f + A.T @ v[0]
        r2 = np.minimum(y[0], v[0])
        assert np.linalg.norm(r1) + np.linalg.norm(r2) <= k["kkt_tol"]


# -- end-to-end: fbstab/test/fbstab_mpc_unit_tests.cc -------------------------
@pytest.mark.parametrize("idx", range(5))
def test_mpc_end_to_end_kats(oracle, kats, idx):
    k = kats["mpc_end_to_end"][idx]
    p = H.mpc_from_kat(k)
    z, l, v, y, out = oracle.solve_mpc(p, opts=default_options(abs_tol=k["abs_tol"]))
    assert out["eflag"][0] == k["eflag"]
    assert out["residual"][0] <= k["residual_max"]
    if "zopt" in k:
        np.testing.assert_allclose(z[0], k["zopt"], atol=k["tol"], rtol=0)
        np.testing.assert_allclose(l[0], k["lopt"], atol=k["tol"], rtol=0)
        np.testing.assert_allclose(v[0], k["vopt"], atol=k["tol"], rtol=0)
    # independent KKT check on the explicit QP
    Hm, f, G, h, A, b = H.mpc_explicit(p)
    assert H.natural_residual_norm(Hm, f, G, h, A, b, z[0], l[0], v[0]) <= 1e-6


# -- components: fbstab/components/test/mpc_component_unit_tests.h ------------
def test_mpc_data_goldens(oracle, kats):
    c = kats["mpc_components"]
    p = H.mpc_component_fixture(c)
    ramp = lambda n: np.arange(1, n + 1, dtype=np.float64)
    for op, x in (("gemvH", ramp(p.nz)), ("gemvA", ramp(p.nz)),
                  ("gemvG", ramp(p.nz)), ("gemvGT", ramp(p.nl)),
                  ("gemvAT", ramp(p.nv))):
        y = oracle.mpc_data_op(p, op, x, 1.0, 0.0)
        assert np.array_equal(y, np.asarray(c[op]["expected"], float)), op
    for op in ("axpyf", "axpyh", "axpyb"):
        y = oracle.mpc_data_op(p, op, None, c[op]["a"], 0.0, y=c[op]["y"])
        assert np.array_equal(y, np.asarray(c[op]["expected"], float)), op


def test_mpc_data_matches_explicit_matrices(oracle):
    """gemv*/axpy* against the explicit (H,G,A,f,h,b) on a random LTV problem,
    including the a=-1 and general-a branches (mpc_data.cc:43-61)."""
    rng = np.random.default_rng(0)
    from tools import fixtures as fx
    N, nx, nu, nc = 3, 4, 2, 5
    p = fx.MpcProblem(N, nx, nu, nc)
    p.arrays = {k: rng.standard_normal((1, n)) for k, n in p.seq_lengths().items()}
    Hm, f, G, h, A, b = H.mpc_explicit(p)
    xz, xl, xv = (rng.standard_normal(n) for n in (p.nz, p.nl, p.nv))
    for a_ in (1.0, -1.0, 0.37):
        for b_ in (0.0, 1.0, -0.5):
            for op, M, x in (("gemvH", Hm, xz), ("gemvA", A, xz), ("gemvG", G, xz),
                             ("gemvAT", A.T, xv), ("gemvGT", G.T, xl)):
                if op == "gemvGT" and a_ == 0.37:
                    continue  # reference quirk: B' term dropped (mpc_data.cc:192-194)
                y0 = rng.standard_normal(M.shape[0])
                y = oracle.mpc_data_op(p, op, x, a_, b_, y=y0)
                np.testing.assert_allclose(y, a_ * (M @ x) + b_ * y0, atol=1e-12)
    for op, vec in (("axpyf", f), ("axpyh", h), ("axpyb", b)):
        y0 = rng.standard_normal(vec.size)
        y = oracle.mpc_data_op(p, op, None, 1.7, 0.0, y=y0)
        np.testing.assert_allclose(y, y0 + 1.7 * vec, atol=1e-13)


def test_mpc_variable_axpy_golden(oracle, kats):
    """mpc_component_unit_tests.h:279-313: x.Fill(1); y.Fill(1); x.axpy(-2,y)
    -> y-margin = (b - A*1) - 2*((b - A*1) - b)."""
    c = kats["mpc_components"]
    p = H.mpc_component_fixture(c)
    one = lambda n: np.ones(n)
    r = oracle.probe(p, one(p.nz), one(p.nl), one(p.nv), one(p.nz), one(p.nl),
                     one(p.nv), 1.0)
    bvec = -p.arrays["d"][0]
    yx = r["x_y"] + (-2.0) * r["xbar_y"] - (-2.0) * bvec
    np.testing.assert_array_equal(yx, np.asarray(c["variable_axpy"]["y"], float))


def test_mpc_inner_residual_golden(oracle, kats):
    c = kats["mpc_components"]
    g = c["inner_residual"]
    p = H.mpc_component_fixture(c)
    f = lambda n, a: np.full(n, a)
    r = oracle.probe(p, f(p.nz, 2.0), f(p.nl, 2.0), f(p.nv, 2.0), f(p.nz, -2.0),
                     f(p.nl, -2.0), f(p.nv, -2.0), 1.0, alpha=0.95)
    rz, rl, rv = np.split(r["inner"], [p.nz, p.nz + p.nl])
    np.testing.assert_allclose(rz, g["rz"], atol=g["tol"], rtol=0)
    np.testing.assert_allclose(rl, g["rl"], atol=g["tol"], rtol=0)
    np.testing.assert_allclose(rv, g["rv"], atol=g["tol"], rtol=0)


def _newton_system_residual(Hm, G, A, sigma, gamma, mus, dx, r, nz, nl, nv):
    dz, dl, dv, dy = np.split(dx, [nz, nz + nl, nz + nl + nv])
    rz, rl, rv = np.split(r, [nz, nz + nl])
    e1 = (Hm @ dz + sigma * dz + G.T @ dl + A.T @ dv) - rz
    e2 = (-G @ dz + sigma * dl) - rl
    e3 = (-gamma * (A @ dz) + mus * dv) - rv
    return e1, e2, e3, dy


def test_riccati_recursion_residual(oracle, kats):
    """mpc_component_unit_tests.h:386-461: the Riccati step solves the Newton
    system (28) block by block."""
    c = kats["mpc_components"]
    tol = c["riccati_recursion"]["tol"]
    p = H.mpc_component_fixture(c)
    f = lambda n, a: np.full(n, a)
    r = np.full(p.nz + p.nl + p.nv, 2.5)
    out = oracle.probe(p, f(p.nz, 1.0), f(p.nl, 2.0), f(p.nv, 4.0), f(p.nz, 2.0),
                       f(p.nl, 1.0), f(p.nv, 3.0), 1.0, r=r, want_dx=True)
    assert out["rc"] == 0
    Hm, fv, G, h, A, b = H.mpc_explicit(p)
    e1, e2, e3, dy = _newton_system_residual(Hm, G, A, 1.0, out["gamma"], out["mus"],
                                             out["dx"], r, p.nz, p.nl, p.nv)
    assert np.abs(e1).max() <= 10 * tol
    assert np.abs(e2).max() <= 10 * tol
    assert np.abs(e3).max() <= 10 * tol
    dz = out["dx"][:p.nz]
    np.testing.assert_allclose(dy, b - A @ dz, atol=10 * tol)


def test_riccati_vs_dense_kkt_on_baseline_shape(oracle):
    """Riccati step vs a dense KKT solve at sigma=1e-8 on the BASELINE MPC
    shape (cond(K) ~ 1e11): relative agreement <= 1e-7."""
    from tools import fixtures as fx
    p = fx.synthetic_mpc_batch(1, first_id=3)
    rng = np.random.default_rng(1)
    z, l = rng.standard_normal(p.nz), rng.standard_normal(p.nl)
    v = np.abs(rng.standard_normal(p.nv))
    r = rng.standard_normal(p.nz + p.nl + p.nv)
    sigma = 1e-8
    out = oracle.probe(p, z, l, v, z * 0.9, l * 0.9, v * 0.9, sigma, r=r, want_dx=True)
    assert out["rc"] == 0
    Hm, fv, G, h, A, b = H.mpc_explicit(p)
    nz, nl, nv = p.nz, p.nl, p.nv
    K = np.block([[Hm + sigma * np.eye(nz), G.T, A.T],
                  [-G, sigma * np.eye(nl), np.zeros((nl, nv))],
                  [-out["gamma"][:, None] * A, np.zeros((nv, nl)), np.diag(out["mus"])]])
    ref = np.linalg.solve(K, r)
    got = out["dx"][:nz + nl + nv]
    assert np.abs(got - ref).max() <= 1e-7 * (1 + np.abs(ref).max())


# -- components: fbstab/components/test/dense_unit_tests.h --------------------
def test_dense_residual_goldens(oracle, kats):
    c = kats["dense_components"]
    p = H.dense_from_kat(dict(H=c["H"], f=c["f"], A=c["A"], b=c["b"]))
    g = c["inner_residual"]
    e = np.zeros(0)
    r = oracle.probe(p, g["x_z"], e, g["x_v"], g["xbar_z"], e, g["xbar_v"], g["sigma"])
    np.testing.assert_allclose(r["inner"][:2], g["rz"], atol=g["tol"] * 10, rtol=0)
    np.testing.assert_allclose(r["inner"][2:], g["rv"], atol=g["tol"] * 10, rtol=0)
    g = c["natural_residual"]
    np.testing.assert_allclose(r["natural"][:2], g["rz"], atol=g["tol"] * 10, rtol=0)
    np.testing.assert_allclose(r["natural"][2:], g["rv"], atol=g["tol"] * 10, rtol=0)


def test_dense_linear_solver_residual(oracle, kats):
    c = kats["dense_components"]
    g = c["linear_solver"]
    p = H.dense_from_kat(dict(H=c["H"], f=c["f"], A=c["A"], b=c["b"]))
    e = np.zeros(0)
    r = np.full(4, g["r_fill"])
    out = oracle.probe(p, g["x_z"], e, g["x_v"], g["xbar_z"], e, g["xbar_v"],
                       g["sigma"], r=r, want_dx=True)
    Hm, f, G, h, A, b = H.dense_explicit(p)
    e1, e2, e3, dy = _newton_system_residual(Hm, G, A, g["sigma"], out["gamma"],
                                             out["mus"], out["dx"], r, 2, 0, 2)
    assert np.sqrt(e1 @ e1 + e3 @ e3) <= g["tol"]


def test_dense_ldlt_against_numpy(oracle):
    """Pivoted LDL' restatement: Newton step on a random 50/10/100 QP equals a
    dense solve of the un-eliminated system."""
    from tools import fixtures as fx
    p = fx.synthetic_dense_batch(1, 50, 10, 100, first_id=5)
    rng = np.random.default_rng(2)
    z, l = rng.standard_normal(50), rng.standard_normal(10)
    v = np.abs(rng.standard_normal(100))
    r = rng.standard_normal(160)
    sigma = 1e-8
    out = oracle.probe(p, z, l, v, 0 * z, 0 * l, 0 * v, sigma, r=r, want_dx=True)
    Hm, f, G, h, A, b = H.dense_explicit(p)
    K = np.block([[Hm + sigma * np.eye(50), G.T, A.T],
                  [-G, sigma * np.eye(10), np.zeros((10, 100))],
                  [-out["gamma"][:, None] * A, np.zeros((100, 10)), np.diag(out["mus"])]])
    ref = np.linalg.solve(K, r)
    assert np.abs(out["dx"][:160] - ref).max() <= 1e-7 * (1 + np.abs(ref).max())


def test_feasibility_certificates(oracle, kats):
    c = kats["dense_components"]
    g = c["primal_infeasibility"]
    p = H.dense_from_kat(g)
    e = np.zeros(0)
    r = oracle.probe(p, [0, 0], e, g["v"], [0, 0], e, g["v"], 1.0, feas_tol=g["tol"])
    assert r["feas"] == 1  # PRIMAL_INFEASIBLE only
    g = c["dual_infeasibility"]
    p = H.dense_from_kat(g)
    r = oracle.probe(p, g["z"], e, np.zeros(4), g["z"], e, np.zeros(4), 1.0,
                     feas_tol=g["tol"])
    assert r["feas"] == 2  # DUAL_INFEASIBLE only


# -- the restated loop is the reference's loop ---------------------------------
def test_restated_loop_equals_reference_template(oracle, ref_oracle, kats):
    """oracle/_ref drives the same components with the reference's own
    fbstab_algorithm.h; outputs must be identical (bitwise)."""
    from tools import fixtures as fx
    cases = []
    for k in kats["dense_end_to_end"]:
        cases.append(("dense", H.dense_from_kat(k), default_options(abs_tol=1e-8)))
    for k in kats["mpc_end_to_end"][:4]:
        cases.append(("mpc", H.mpc_from_kat(k), default_options(abs_tol=1e-8)))
    cases.append(("mpc", fx.synthetic_mpc_batch(6), default_options()))
    cases.append(("mpc", fx.synthetic_mpc_batch(3, first_id=11), reliable_options(display_level=0)))
    cases.append(("dense", fx.synthetic_dense_batch(6, 20, 5, 40), default_options()))
    cases.append(("dense", fx.synthetic_dense_batch(3, 50, 10, 100), default_options(max_newton_iters=7)))
    for kind, p, o in cases:
        fn = (lambda orc: orc.solve_dense(p, opts=o)) if kind == "dense" else (
            lambda orc: orc.solve_mpc(p, opts=o))
        a = fn(oracle)
        b = fn(ref_oracle)
        for i in range(4):
            assert np.array_equal(a[i], b[i])
        for fld in ("eflag", "newton_iters", "prox_iters", "residual", "initial_residual"):
            assert np.array_equal(a[4][fld], b[4][fld]), fld


def test_options_validate_and_profiles(oracle):
    """ValidateOptions clamps (fbstab_algorithm-impl.h:7-31) are applied by
    UpdateParameters: absurd options still solve."""
    from tools import fixtures as fx
    p = fx.synthetic_dense_batch(2, 20, 5, 40)
    o = default_options(alpha=7.0, beta=-1.0, eta=5.0, delta=9.0, max_newton_iters=-3,
                        max_linesearch_iters=0)
    z, l, v, y, out = oracle.solve_dense(p, opts=o)
    assert (out["newton_iters"] == 1).all() and (out["eflag"] == 2).all()
    z, l, v, y, out = oracle.solve_dense(p, opts=reliable_options())
    assert (out["eflag"] == 0).all()
    assert np.abs(z - p.solution["z"]).max() < 1e-3


def test_synthetic_workloads_solve(oracle):
    from tools import fixtures as fx
    p = fx.synthetic_mpc_batch(8)
    z, l, v, y, out = oracle.solve_mpc(p, nthreads=2)
    assert (out["eflag"] == 0).all() and (out["residual"] <= 1.1e-6).all()
    q = fx.synthetic_mpc_batch(3, first_id=5)
    for k in p.arrays:  # instance ids, not batch positions, key the data
        assert np.array_equal(p.arrays[k][5:8], q.arrays[k])
    d = fx.synthetic_dense_batch(8, 50, 10, 100)
    z, l, v, y, out = oracle.solve_dense(d)
    assert (out["eflag"] == 0).all()
    assert np.abs(z - d.solution["z"]).max() < 1e-6
