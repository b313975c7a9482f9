"""TEST INFRASTRUCTURE ONLY: loader for the single-threaded host compilation of
the device solver logic (see hostsim.cc)."""
import ctypes as C
import os
import subprocess

import numpy as np

from oracle.oracle_py import SolverOut, Options, default_options, _out_to_numpy, _p

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhostsim.so")


def build():
    src = os.path.join(_HERE, "hostsim.cc")
    shim = os.path.join(_HERE, "shim")  # <hip/hip_runtime.h> for a host of one thread
    deps = [src, os.path.join(shim, "hip", "hip_runtime.h")] + [
        os.path.join(_HERE, "..", "..", "fbstab_amd", "csrc", f) for f in ("fb_common.h", "fb_algorithm.h", "fb_mpc.h", "fb_dense.h")]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(d) for d in deps):
        return
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I" + shim,
                           "-Wno-attributes", "-Wno-unknown-pragmas", "-o", _SO, src])


class HostSim:
    def __init__(self):
        build()
        self.lib = C.CDLL(_SO)

    def solve_mpc(self, prob, x0guess=None, opts=None):
        opts = opts or default_options()
        B = prob.batch
        z = np.zeros((B, prob.nz)); l = np.zeros((B, prob.nl))
        v = np.zeros((B, prob.nv)); y = np.zeros((B, prob.nv))
        if x0guess is not None:
            z[:] = x0guess[0]; l[:] = x0guess[1]; v[:] = x0guess[2]
        out = (SolverOut * B)()
        N, nx, nu, nc = prob.sizes()
        for b in range(B):
            a = {k: np.ascontiguousarray(prob.arrays[k][b]) for k in prob.arrays}
            self.lib.hostsim_mpc_solve(
                N, nx, nu, nc, _p(a["Q"]), _p(a["R"]), _p(a["S"]), _p(a["q"]), _p(a["r"]),
                _p(a["A"]), _p(a["B"]), _p(a["c"]), _p(a["E"]), _p(a["L"]), _p(a["d"]),
                _p(a["x0"]), _p(z[b]), _p(l[b]), _p(v[b]), _p(y[b]), C.byref(opts),
                C.byref(out[b]))
        return z, l, v, y, _out_to_numpy(out)

    def solve_dense(self, prob, x0guess=None, opts=None):
        opts = opts or default_options()
        B = prob.batch
        z = np.zeros((B, prob.nz)); l = np.zeros((B, max(prob.nl, 1)))[:, :prob.nl]
        l = np.ascontiguousarray(l)
        v = np.zeros((B, prob.nv)); y = np.zeros((B, prob.nv))
        if x0guess is not None:
            z[:] = x0guess[0]; l[:] = x0guess[1]; v[:] = x0guess[2]
        out = (SolverOut * B)()
        for b in range(B):
            a = {k: np.ascontiguousarray(prob.arrays[k][b]) for k in prob.arrays}
            lb = l[b] if prob.nl else np.zeros(1)
            G = a["G"] if prob.nl else np.zeros(1)
            h = a["h"] if prob.nl else np.zeros(1)
            self.lib.hostsim_dense_solve(
                prob.nz, prob.nl, prob.nv, _p(a["H"]), _p(a["f"]), _p(G), _p(h),
                _p(a["A"]), _p(a["b"]), _p(z[b]), _p(lb), _p(v[b]), _p(y[b]),
                C.byref(opts), C.byref(out[b]))
            if prob.nl:
                l[b] = lb
        return z, l, v, y, _out_to_numpy(out)

    def newton_mpc(self, prob, b, x, xbar, sigma, alpha, refine_sweeps=0):
        """One Newton step of the flat-vector device logic for QP ``b`` at x = (z, l, v), xbar (hostsim.cc:
        hostsim_mpc_newton_refined), followed by ``refine_sweeps`` refinement sweeps.  Returns a dict."""
        N, nx, nu, nc = prob.sizes()
        a = {k: np.ascontiguousarray(prob.arrays[k][b]) for k in prob.arrays}
        nz, nl, nv = prob.nz, prob.nl, prob.nv
        out = np.zeros(3 * nz + 3 * nl + 2 * nv + 2)
        z, l, v = (np.ascontiguousarray(t, dtype=np.float64) for t in x)
        zb, lb, vb = (np.ascontiguousarray(t, dtype=np.float64) for t in xbar)
        self.lib.hostsim_mpc_newton_refined.argtypes = [C.c_int] * 4 + [C.c_void_p] * 18 + [C.c_double, C.c_double, C.c_int, C.c_void_p]
        rc = self.lib.hostsim_mpc_newton_refined(
            N, nx, nu, nc, _p(a["Q"]), _p(a["R"]), _p(a["S"]), _p(a["q"]), _p(a["r"]), _p(a["A"]), _p(a["B"]),
            _p(a["c"]), _p(a["E"]), _p(a["L"]), _p(a["d"]), _p(a["x0"]), _p(z), _p(l), _p(v), _p(zb), _p(lb), _p(vb),
            C.c_double(sigma), C.c_double(alpha), refine_sweeps, _p(out))
        o, r = 0, {"ok": rc == 0}
        for name, n in (("dz", nz), ("dl", nl), ("dv", nv), ("adz", nv), ("wz", nz), ("wl", nl), ("rz", nz), ("rl", nl)):
            r[name] = out[o:o + n].copy()
            o += n
        r["lin2_before"], r["lin2_after"] = out[o], out[o + 1]
        return r
