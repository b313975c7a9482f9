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
