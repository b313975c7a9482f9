"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes loader for the CPU oracle (``oracle/liboracle.so``) and, where it has
been built, the reference-loop variant (``oracle/_ref/libfbstab_ref.so``).
Importable only from tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg; the product package ``fbstab_amd`` never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = "/root/reference"


class SolverOut(C.Structure):
    """fbstab_solver_out_t (include/fbstab_types.h)."""
    _fields_ = [("eflag", C.c_int), ("pad_", C.c_int), ("residual", C.c_double),
                ("newton_iters", C.c_int), ("prox_iters", C.c_int),
                ("solve_time", C.c_double), ("initial_residual", C.c_double)]


class Options(C.Structure):
    """fbstab_options_t (include/fbstab_types.h)."""
    _fields_ = [(n, C.c_double) for n in (
        "sigma0", "sigma_max", "sigma_min", "alpha", "beta", "eta", "delta",
        "gamma", "abs_tol", "rel_tol", "stall_tol", "infeas_tol",
        "inner_tol_max", "inner_tol_min")] + [(n, C.c_int) for n in (
            "max_newton_iters", "max_prox_iters", "max_inner_iters",
            "max_linesearch_iters", "check_feasibility",
            "nonmonotone_linesearch", "display_level", "reserved")]


def default_options(**kw) -> Options:
    """AlgorithmParameters::DefaultParameters (fbstab_algorithm-impl.h:33-59)
    with display OFF unless overridden."""
    o = Options(1e-8, 1e-6, 1e-12, 0.95, 0.75, 1e-8, 0.2, 0.1, 1e-6, 1e-12, 1e-10,
                1e-8, 1e-2, 1e-12, 200, 30, 50, 20, 1, 1, 0, 0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def reliable_options(**kw) -> Options:
    """AlgorithmParameters::ReliableParameters (fbstab_algorithm-impl.h:61-74)."""
    o = default_options(sigma0=1e-4, sigma_max=1e-2, sigma_min=1e-10, beta=0.9,
                        abs_tol=1e-4, rel_tol=1e-6, max_linesearch_iters=40,
                        max_newton_iters=500, max_prox_iters=100,
                        nonmonotone_linesearch=0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def build(ref: bool = False, quiet: bool = True, fma: bool = False) -> None:
    """``make`` (and ``make ref`` when the reference tree is present, ``make fma`` on request)."""
    out = subprocess.DEVNULL if quiet else None
    subprocess.check_call(["make", "-C", _HERE], stdout=out)
    if fma:
        subprocess.check_call(["make", "-C", _HERE, "fma"], stdout=out)
    if ref and os.path.isdir(REFERENCE_ROOT):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=out)


_dp = C.POINTER(C.c_double)


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


class Oracle:
    """Handle on one of the two oracle libraries."""

    def __init__(self, use_reference_loop: bool = False, fma: bool = False):
        """``fma``: liboracle_fma.so - the same sources compiled with fused multiply-adds allowed
        (oracle/Makefile: ``make fma``)."""
        assert not (use_reference_loop and fma)
        path = (os.path.join(_HERE, "_ref", "libfbstab_ref.so") if use_reference_loop else
                os.path.join(_HERE, "liboracle_fma.so" if fma else "liboracle.so"))
        if not os.path.exists(path):
            build(ref=use_reference_loop, fma=fma)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.lib = C.CDLL(path)
        self.lib.fbo_last_error.restype = C.c_char_p
        assert bool(self.lib.fbo_uses_reference_algorithm()) == use_reference_loop

    def last_error(self) -> str:
        return self.lib.fbo_last_error().decode()

    def num_threads(self) -> int:
        return int(self.lib.fbo_num_threads())

    # -- solves -------------------------------------------------------------
    def solve_mpc(self, prob, x0guess=None, opts: Optional[Options] = None,
                  nthreads: int = 1):
        """Solve every QP of ``prob`` (fixtures.MpcProblem).  Returns
        ``(z, l, v, y, out)`` with arrays ``(batch, n)`` and ``out`` a numpy
        structured array of SolverOut records."""
        opts = opts or default_options()
        B = prob.batch
        z = np.zeros((B, prob.nz))
        l = np.zeros((B, prob.nl))
        v = np.zeros((B, prob.nv))
        y = np.zeros((B, prob.nv))
        if x0guess is not None:
            z[:] = x0guess[0]
            l[:] = x0guess[1]
            v[:] = x0guess[2]
        a = {k: np.ascontiguousarray(prob.arrays[k], dtype=np.float64)
             for k in ("Q", "R", "S", "q", "r", "A", "B", "c", "E", "L", "d", "x0")}
        strides = (C.c_longlong * 16)(*([a[k].shape[1] for k in
                                         ("Q", "R", "S", "q", "r", "A", "B", "c",
                                          "E", "L", "d", "x0")] +
                                        [prob.nz, prob.nl, prob.nv, prob.nv]))
        out = (SolverOut * B)()
        N, nx, nu, nc = prob.sizes()
        nfail = self.lib.fbo_mpc_solve_batch(
            B, N, nx, nu, nc, _p(a["Q"]), _p(a["R"]), _p(a["S"]), _p(a["q"]),
            _p(a["r"]), _p(a["A"]), _p(a["B"]), _p(a["c"]), _p(a["E"]),
            _p(a["L"]), _p(a["d"]), _p(a["x0"]), _p(z), _p(l), _p(v), _p(y),
            strides, C.byref(opts), out, nthreads)
        if nfail:
            raise RuntimeError(self.last_error())
        return z, l, v, y, _out_to_numpy(out)

    def solve_dense(self, prob, x0guess=None, opts: Optional[Options] = None,
                    nthreads: int = 1, raise_on_error: bool = True):
        opts = opts or default_options()
        B = prob.batch
        z = np.zeros((B, prob.nz))
        l = np.zeros((B, prob.nl))
        v = np.zeros((B, prob.nv))
        y = np.zeros((B, prob.nv))
        if x0guess is not None:
            z[:] = x0guess[0]
            l[:] = x0guess[1]
            v[:] = x0guess[2]
        a = {k: np.ascontiguousarray(prob.arrays[k], dtype=np.float64)
             for k in ("H", "f", "G", "h", "A", "b")}
        strides = (C.c_longlong * 10)(*([a[k].shape[1] for k in
                                         ("H", "f", "G", "h", "A", "b")] +
                                        [prob.nz, prob.nl, prob.nv, prob.nv]))
        out = (SolverOut * B)()
        nfail = self.lib.fbo_dense_solve_batch(
            B, prob.nz, prob.nl, prob.nv, _p(a["H"]), _p(a["f"]), _p(a["G"]),
            _p(a["h"]), _p(a["A"]), _p(a["b"]), _p(z), _p(l), _p(v), _p(y),
            strides, C.byref(opts), out, nthreads)
        if nfail and raise_on_error:
            raise RuntimeError(self.last_error())
        return z, l, v, y, _out_to_numpy(out)

    def solve_display(self, prob, x0guess=None, opts: Optional[Options] = None,
                      text_cap: int = 1 << 20, trace_cap: int = 1 << 14):
        """Solve QP 0 of ``prob`` with its display captured.  Returns
        ``(z, l, v, y, out, text, records)``: the reference-loop library fills
        ``text`` with what the reference prints at ``opts.display_level``
        (fbstab_algorithm-impl.h:411-541), the restated library fills
        ``records`` (``(n, 8)`` array of fbstab_trace_record_t)."""
        opts = opts or default_options()
        z, l, v, y = (np.zeros(n) for n in (prob.nz, prob.nl, prob.nv, prob.nv))
        if x0guess is not None:
            z[:], l[:], v[:] = (np.asarray(g, dtype=np.float64).reshape(-1) for g in x0guess)
        A = {k: np.ascontiguousarray(prob.arrays[k][0], dtype=np.float64) for k in prob.arrays}
        out = SolverOut()
        text = C.create_string_buffer(text_cap)
        rec = np.zeros((trace_cap, 8))
        nrec = C.c_int(0)
        tail = (_p(z), _p(l), _p(v), _p(y), C.byref(opts), C.byref(out), text, text_cap,
                _p(rec), trace_cap, C.byref(nrec))
        if hasattr(prob, "N"):
            N, nx, nu, nc = prob.sizes()
            rc = self.lib.fbo_mpc_solve_display(
                N, nx, nu, nc, _p(A["Q"]), _p(A["R"]), _p(A["S"]), _p(A["q"]), _p(A["r"]),
                _p(A["A"]), _p(A["B"]), _p(A["c"]), _p(A["E"]), _p(A["L"]), _p(A["d"]),
                _p(A["x0"]), *tail)
        else:
            G = A["G"] if prob.nl else np.zeros(1)
            h = A["h"] if prob.nl else np.zeros(1)
            rc = self.lib.fbo_dense_solve_display(
                prob.nz, prob.nl, prob.nv, _p(A["H"]), _p(A["f"]), _p(G), _p(h), _p(A["A"]),
                _p(A["b"]), *tail)
        if rc:
            raise RuntimeError(self.last_error())
        o = _out_to_numpy((SolverOut * 1)(out))
        return z, l, v, y, o, text.value.decode(), rec[:min(nrec.value, trace_cap)].copy()

    # -- component probes -----------------------------------------------------
    def mpc_data_op(self, prob, which: str, x, a=1.0, b=0.0, y=None):
        idx = ["gemvH", "gemvA", "gemvG", "gemvAT", "gemvGT", "axpyf", "axpyh",
               "axpyb"].index(which)
        nout = {0: prob.nz, 1: prob.nv, 2: prob.nl, 3: prob.nz, 4: prob.nz,
                5: prob.nz, 6: prob.nl, 7: prob.nv}[idx]
        yv = np.zeros(nout) if y is None else np.array(y, dtype=np.float64)
        xv = np.zeros(1) if x is None else np.array(x, dtype=np.float64)
        A = {k: np.ascontiguousarray(prob.arrays[k][0]) for k in prob.arrays}
        N, nx, nu, nc = prob.sizes()
        rc = self.lib.fbo_mpc_data_op(
            N, nx, nu, nc, _p(A["Q"]), _p(A["R"]), _p(A["S"]), _p(A["q"]),
            _p(A["r"]), _p(A["A"]), _p(A["B"]), _p(A["c"]), _p(A["E"]),
            _p(A["L"]), _p(A["d"]), _p(A["x0"]), idx, _p(xv), xv.size,
            C.c_double(a), C.c_double(b), _p(yv), yv.size)
        if rc:
            raise RuntimeError(self.last_error())
        return yv

    def probe(self, prob, z, l, v, zb, lb, vb, sigma, alpha=0.95, r=None,
              want_dx=False, feas_tol=None):
        """Component probe (see fbo_mpc_probe / fbo_dense_probe).  Returns a
        dict with x_y, xbar_y, inner, natural, pnr and optionally dx, gamma,
        mus, feas."""
        nz, nl, nv = prob.nz, prob.nl, prob.nv
        f64 = lambda a, n: np.ascontiguousarray(
            np.asarray(a, dtype=np.float64).reshape(-1)) if n else np.zeros(1)
        z, l, v = f64(z, nz), f64(l, nl), f64(v, nv)
        zb, lb, vb = f64(zb, nz), f64(lb, nl), f64(vb, nv)
        res = dict(x_y=np.zeros(nv), xbar_y=np.zeros(nv),
                   inner=np.zeros(nz + nl + nv), natural=np.zeros(nz + nl + nv),
                   pnr=np.zeros(nz + nl + nv))
        dx = gamma = mus = None
        rr = None
        if want_dx:
            rr = f64(r, 1)
            assert rr.size == nz + nl + nv
            dx = np.zeros(nz + nl + 2 * nv)
            gamma = np.zeros(nv)
            mus = np.zeros(nv)
        feas = C.c_int(-1)
        A = {k: np.ascontiguousarray(prob.arrays[k][0]) for k in prob.arrays}
        tail = (_p(z), _p(l), _p(v), _p(zb), _p(lb), _p(vb), C.c_double(sigma),
                C.c_double(alpha), _p(rr), _p(res["x_y"]), _p(res["xbar_y"]),
                _p(res["inner"]), _p(res["natural"]), _p(res["pnr"]), _p(dx),
                _p(gamma), _p(mus),
                C.byref(feas) if feas_tol is not None else None,
                C.c_double(feas_tol or 0.0))
        if hasattr(prob, "N"):
            N, nx, nu, nc = prob.sizes()
            rc = self.lib.fbo_mpc_probe(
                N, nx, nu, nc, _p(A["Q"]), _p(A["R"]), _p(A["S"]), _p(A["q"]),
                _p(A["r"]), _p(A["A"]), _p(A["B"]), _p(A["c"]), _p(A["E"]),
                _p(A["L"]), _p(A["d"]), _p(A["x0"]), *tail)
        else:
            G = A["G"] if nl else np.zeros(1)
            h = A["h"] if nl else np.zeros(1)
            rc = self.lib.fbo_dense_probe(
                nz, nl, nv, _p(A["H"]), _p(A["f"]), _p(G), _p(h), _p(A["A"]),
                _p(A["b"]), *tail)
        if rc == 1:
            raise RuntimeError(self.last_error())
        res["rc"] = rc
        if want_dx:
            res.update(dx=dx, gamma=gamma, mus=mus)
        if feas_tol is not None:
            res["feas"] = feas.value
        return res


OUT_DTYPE = np.dtype([("eflag", np.int32), ("pad_", np.int32),
                      ("residual", np.float64), ("newton_iters", np.int32),
                      ("prox_iters", np.int32), ("solve_time", np.float64),
                      ("initial_residual", np.float64)])


def _out_to_numpy(out) -> np.ndarray:
    return np.frombuffer(bytes(out), dtype=OUT_DTYPE).copy()
