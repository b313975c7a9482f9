"""CPU-only check of the DEVICE solver logic: fbstab_amd/csrc/fb_*.h compiled
single-threaded for the host (tests/hostsim) against the oracle.  This guards
the kernel arithmetic where no GPU exists; the real parity tests are the
``-m gpu`` ones."""
import numpy as np
import pytest

from tools import fixtures as fx
from oracle.oracle_py import default_options, reliable_options
from tests import helpers as H


@pytest.fixture(scope="module")
def hostsim():
    from tests.hostsim import HostSim
    return HostSim()


def _same(a, b, abs_tol):
    assert np.array_equal(a[4]["eflag"], b[4]["eflag"])
    assert np.array_equal(a[4]["prox_iters"], b[4]["prox_iters"])
    assert np.abs(a[4]["newton_iters"].astype(int) - b[4]["newton_iters"].astype(int)).max() <= 1
    for i in range(4):
        if a[i].size:
            assert np.abs(a[i] - b[i]).max() <= 10 * abs_tol * (1 + np.abs(b[i]).max())


def test_reference_tests_through_device_logic(hostsim, oracle, kats):
    o = default_options(abs_tol=1e-8)
    for k in kats["dense_end_to_end"]:
        p = H.dense_from_kat(k)
        a, b = hostsim.solve_dense(p, opts=o), oracle.solve_dense(p, opts=o)
        assert a[4]["eflag"][0] == k["eflag"] == b[4]["eflag"][0]
        assert a[4]["newton_iters"][0] == b[4]["newton_iters"][0]
        if k["eflag"] == 0:
            _same(a, b, 1e-8)
    for k in kats["mpc_end_to_end"]:
        p = H.mpc_from_kat(k)
        a, b = hostsim.solve_mpc(p, opts=o), oracle.solve_mpc(p, opts=o)
        _same(a, b, 1e-8)
        if "zopt" in k:
            np.testing.assert_allclose(a[0][0], k["zopt"], atol=k["tol"], rtol=0)
            np.testing.assert_allclose(a[1][0], k["lopt"], atol=k["tol"], rtol=0)


def test_synthetic_workloads_through_device_logic(hostsim, oracle):
    p = fx.synthetic_mpc_batch(24)
    for o in (default_options(), reliable_options(), default_options(max_newton_iters=4)):
        _same(hostsim.solve_mpc(p, opts=o), oracle.solve_mpc(p, opts=o), max(o.abs_tol, 1e-7))
    d = fx.synthetic_dense_batch(24, 50, 10, 100)
    for o in (default_options(), default_options(check_feasibility=0, nonmonotone_linesearch=0)):
        _same(hostsim.solve_dense(d, opts=o), oracle.solve_dense(d, opts=o), o.abs_tol)


def test_natural_order_elimination_of_the_dense_kkt_matrix():
    """The one-wavefront dense kernel factors K = [E G'; G -sigma I] in the NATURAL order and
    eliminates the right-hand side along with the matrix (fb_dense_wave.h,
    factor_solve_static): step k scales column k by 1/d_k, updates the columns behind it and
    the right-hand side; what step k leaves in row k - d_k and row k of D L' - serves the
    backward sweep, every row taking its own dot product.  Same statements here in numpy, on
    quasi-definite matrices of the kind the Newton step assembles (sigma = 1.5e-8, Gamma up
    to 1/sigma), against numpy's pivoted LU: the natural order needs no pivoting for them."""
    rng = np.random.default_rng(5)
    for (nz, nl, nv) in ((50, 10, 100), (30, 20, 64), (7, 0, 9), (48, 16, 131)):
        sigma = 1.5e-8
        M = rng.standard_normal((nz, nz))
        H = M @ M.T / nz
        A = rng.standard_normal((nv, nz))
        G = rng.standard_normal((nl, nz))
        gam = np.where(rng.random(nv) < 0.3, 1.0 / sigma, rng.random(nv))
        n = nz + nl
        K = np.zeros((n, n))
        K[:nz, :nz] = H + sigma * np.eye(nz) + A.T @ (gam[:, None] * A)
        K[nz:, :nz] = G
        K[:nz, nz:] = G.T
        K[nz:, nz:] = -sigma * np.eye(nl)
        b = rng.standard_normal(n)
        Kr, x = K.copy(), b.copy()          # row t of Kr <-> lane t
        dinv = np.zeros(n)
        for k in range(n):
            d = Kr[k, k]
            assert abs(d) > 0 and np.isfinite(d)
            dinv[k] = 1.0 / d
            col = Kr[:, k].copy()           # K[t][k] = K[k][t]: the pivot row, by symmetry
            nlm = np.where(np.arange(n) > k, -col * dinv[k], 0.0)
            x += nlm * x[k]
            Kr[:, k + 1:] += np.outer(nlm, col[k + 1:])
        x *= dinv                           # D L' w = y, row t holds row t of D L' in columns t+1..
        for j in range(n - 1, 0, -1):
            u = np.where(np.arange(n) < j, Kr[:, j] * dinv, 0.0)
            x -= u * x[j]
        # (cond(K) up to 1e17 where the active rows and the equalities together outnumber the
        # variables: what an elimination can promise is a small residual, in the norm)
        ref = np.linalg.solve(K, b)
        bound = lambda v: 1e-12 * (np.abs(K).sum(axis=1).max() * np.abs(v).max() + np.abs(b).max())
        assert np.abs(K @ x - b).max() <= bound(x)
        assert np.abs(K @ ref - b).max() <= bound(ref)
        assert np.sign(np.diag(Kr)[:nz]).min() > 0 and (nl == 0 or np.sign(np.diag(Kr)[nz:]).max() < 0)


def _fuzz42_shape128():
    """The shape of the one deviation round 4 found (tools/fuzz_shapes.py seed 42, shape 128: N=3, nx=23,
    nu=6, nc=1, ten QPs; tests/test_gpu_components.py::test_one_step_qp_...)."""
    p, _ = H.fuzz_stream_shape(42, 127)
    assert p.sizes() == (3, 23, 6, 1) and p.batch == 10
    return p


def _newton_system_residual(p, q, step):
    return H.newton_system_residual(p, q, step)[0]


def test_substitution_keeps_the_wide_stage_in_the_oracles_class_and_a_refinement_sweep_reaches_rounding(hostsim, oracle):
    """VERDICT r4 item 1.  Up to round 4 the kernels multiplied with explicitly inverted triangular factors
    where the reference substitutes (riccati_linear_solver.cc:234-325); on the 29-wide stage of the fuzz
    stream's one deviation the step left |V dx - r| = 4.6e-6 in the z block against the oracle's 1e-7.  The
    flat-vector logic now substitutes with M and SG (solve_lower, solve_right_t): every one of the ten QPs
    is left within 3 x the oracle's residual.  The optional refinement sweep (MpcProblem::refine_step) takes
    every block to rounding level, and the measure the option decides by (linear_residual2) sees both."""
    p = _fuzz42_shape128()
    zero = lambda n: np.zeros(n)
    x = (zero(p.nz), zero(p.nl), zero(p.nv))
    N, nx, nu, nc = p.sizes()
    for q in range(p.batch):
        s0 = hostsim.newton_mpc(p, q, x, x, 1e-8, 0.95, 0)
        s1 = hostsim.newton_mpc(p, q, x, x, 1e-8, 0.95, 1)
        assert s0["ok"] and s1["ok"]
        one = fx.MpcProblem(N, nx, nu, nc, {k: np.ascontiguousarray(a[q:q + 1]) for k, a in p.arrays.items()})
        pr = oracle.probe(one, *x, *x, 1e-8, 0.95)
        pr = oracle.probe(one, *x, *x, 1e-8, 0.95, r=-pr["inner"], want_dx=True)
        dx = pr["dx"]
        ostep = {"dz": dx[:p.nz], "dl": dx[p.nz:p.nz + p.nl], "dv": dx[p.nz + p.nl:p.nz + p.nl + p.nv]}
        (e0, n0), (e1, n1), (eo, no) = (H.newton_system_residual(p, q, s) for s in (s0, s1, ostep))
        assert n0 <= 3 * no, (q, e0, eo)
        assert max(e1) <= 1e-12, (q, e0, e1)
        assert e1[2] <= 1e-14 and e0[2] <= 1e-14  # the third block row holds exactly either way
        # what the option measures is the z and l residual itself (to the rounding of its own evaluation)
        assert abs(np.sqrt(s1["lin2_before"]) - np.sqrt(sum(np.square(e0[:2])))) <= 3 * max(e0[:2])
        assert np.sqrt(s1["lin2_after"]) <= 1e-11  # (evaluated in working precision: cancellation of O(1) terms)


def test_the_one_step_qp_takes_the_oracles_counts(hostsim, oracle):
    """The same ten QPs through the whole flat-vector solve: the counts are the oracle's on every QP (the QP
    whose one Newton step ends at the accuracy of the linear solve took one more iteration of each kind as
    long as the kernels multiplied with explicit inverses) - with the refinement option off and on."""
    p = _fuzz42_shape128()
    b = oracle.solve_mpc(p, opts=default_options())
    for reserved in (0, 1):
        o = default_options()
        o.reserved = reserved
        a = hostsim.solve_mpc(p, opts=o)
        assert np.array_equal(a[4]["eflag"], b[4]["eflag"]) and (a[4]["eflag"] == 0).all()
        assert np.array_equal(a[4]["prox_iters"], b[4]["prox_iters"]), (reserved, a[4]["prox_iters"], b[4]["prox_iters"])
        assert np.array_equal(a[4]["newton_iters"], b[4]["newton_iters"]), (reserved, a[4]["newton_iters"], b[4]["newton_iters"])


def test_refinement_is_an_option_because_a_more_accurate_step_also_parts_from_the_reference(hostsim, oracle):
    """Why iterative refinement is not the default (Solver::wants_refinement): the reference's own linear
    solve has an error, and its iteration counts are what parity compares with.  On the reference's
    servo-motor problem (ocp_generator.cc:113-200) the second proximal iteration ends at a residual 11 %
    UNDER abs_tol; the oracle's leftover takes its residual over the tolerance there and it runs a third
    iteration - and so does the device logic as it is (3 / 29 on both).  With the option at its mildest
    (reserved = 1: refine only a step whose leftover alone exceeds the tolerance) the refined step is more
    accurate than the oracle's and the solve stops one proximal iteration EARLIER: 2 / 28.  The other
    generator problems do not care."""
    for name in ("DoubleIntegrator", "ServoMotor", "SpacecraftRelativeMotion", "CopolymerizationReactor"):
        gen = fx.OcpGenerator()
        getattr(gen, name)()
        p = gen.GetFBstabInput()
        b = oracle.solve_mpc(p)
        a = hostsim.solve_mpc(p)
        assert np.array_equal(a[4]["eflag"], b[4]["eflag"]), name
        assert np.array_equal(a[4]["prox_iters"], b[4]["prox_iters"]), (name, a[4]["prox_iters"], b[4]["prox_iters"])
        assert np.array_equal(a[4]["newton_iters"], b[4]["newton_iters"]), (name, a[4]["newton_iters"], b[4]["newton_iters"])
        if name == "ServoMotor":
            o = default_options()
            o.reserved = 1
            c = hostsim.solve_mpc(p, opts=o)
            assert (b[4]["prox_iters"][0], b[4]["newton_iters"][0]) == (3, 29)
            assert (c[4]["prox_iters"][0], c[4]["newton_iters"][0]) == (2, 28)
            assert 0.8e-6 <= c[4]["residual"][0] <= 1e-6


def test_a_count_that_only_the_references_rounding_decides(hostsim, oracle):
    """The limit of count parity, pinned.  tools/fuzz_shapes.py's sparse-row family (seed 301, shape 144:
    N = 8, nx = 10, nu = 2, one constraint row of two or three entries, nine QPs) holds ONE QP in ~8,400 whose
    counts part from the oracle's on every kernel: a one-step QP (no constraint active).  The exact Newton step
    lands on the proximal point, residual sigma |z - zbar| = 2.0e-7; the device logic's step leaves 3e-8 of
    the Newton system behind and stops there, 1 / 1.  The ORACLE's own step leaves 1.2e-6 - above abs_tol
    by itself (two inputs cannot reach a state of ten in one stage: Pi = sigma I + [A B] inv(K) [A B]' has
    eigenvalues of order sigma, and what is left of dl after the cancellation depends on the order of every
    sum) - and it runs a second iteration, 2 / 2.  Both end within 3e-7 of each other, a third of the tolerance.
    No arrangement of the device arithmetic short of the reference's own operation order AND rounding
    reproduces that leftover: the refined step agrees with the device's, not with the oracle's, and the
    oracle itself takes 1 / 1 once its compiler may contract a * b + c (last assertion)."""
    p, o = H.fuzz_stream_shape(301, 143, "sparse")
    assert p.sizes() == (8, 10, 2, 1) and p.batch == 9
    N, nx, nu, nc = p.sizes()
    b = oracle.solve_mpc(p, opts=o)
    a = hostsim.solve_mpc(p, opts=o)
    others = np.arange(9) != 4
    assert (a[4]["eflag"] == 0).all() and (b[4]["eflag"] == 0).all()
    assert np.array_equal(a[4]["prox_iters"][others], b[4]["prox_iters"][others])
    assert np.array_equal(a[4]["newton_iters"][others], b[4]["newton_iters"][others])
    assert (a[4]["prox_iters"][4], a[4]["newton_iters"][4]) == (1, 1)
    assert (b[4]["prox_iters"][4], b[4]["newton_iters"][4]) == (2, 2)
    assert a[4]["residual"][4] <= 2.2e-7 and np.abs(a[0][4] - b[0][4]).max() <= 3e-7
    # the Newton step from the cold start, three ways: its leftover in the Newton system
    x = (np.zeros(p.nz), np.zeros(p.nl), np.zeros(p.nv))
    one = fx.MpcProblem(N, nx, nu, nc, {k: np.ascontiguousarray(v[4:5]) for k, v in p.arrays.items()})
    pr = oracle.probe(one, *x, *x, 1e-8, 0.95)
    dx = oracle.probe(one, *x, *x, 1e-8, 0.95, r=-pr["inner"], want_dx=True)["dx"]
    ostep = {"dz": dx[:p.nz], "dl": dx[p.nz:p.nz + p.nl], "dv": dx[p.nz + p.nl:p.nz + p.nl + p.nv]}
    s0 = hostsim.newton_mpc(p, 4, x, x, 1e-8, 0.95, 0)
    s1 = hostsim.newton_mpc(p, 4, x, x, 1e-8, 0.95, 1)
    n0, n1, no = (H.newton_system_residual(p, 4, s)[1] for s in (s0, s1, ostep))
    assert no > 1e-6 > 10 * n0 and n1 <= 1e-13
    assert np.abs(s0["dl"] - s1["dl"]).max() <= 0.1 * np.abs(ostep["dl"] - s1["dl"]).max()
    # ... and the oracle ITSELF, compiled once more with fused multiply-adds allowed (same algorithm, same order
    # of operations - tools/oracle_rounding_sensitivity_mpc.py), takes the device's counts on this QP
    import ctypes, os, subprocess, tempfile
    from oracle.oracle_py import Oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "liboracle_fma.so")
        subprocess.check_call(["g++", "-O3", "-std=c++11", "-fPIC", "-fopenmp", "-ffp-contract=fast", "-mfma", "-shared",
                               "-o", so, os.path.join(root, "oracle", "oracle_capi.cc")])
        fma = Oracle(False)
        fma.lib = ctypes.CDLL(so)
        fma.lib.fbo_last_error.restype = ctypes.c_char_p
        f = fma.solve_mpc(p, opts=o)[4]
    assert np.array_equal(f["prox_iters"], a[4]["prox_iters"]) and np.array_equal(f["newton_iters"], a[4]["newton_iters"])
