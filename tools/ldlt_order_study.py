#!/usr/bin/env python3
"""Developer study (CPU, oracle only): the elimination order of Eigen's rule from one Newton step of a
dense QP to the next - could a kernel assume the previous step's order and verify it?
usage: tools/ldlt_order_study.py [n_qps]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import fixtures as fx
from oracle import oracle_py
so = os.path.join(ROOT, "tools", "_build", "liboracle_order_study.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["g++", "-O3", "-std=c++11", "-fPIC", "-fopenmp", "-ffp-contract=off",
                       "-include", os.path.join(ROOT, "tools/cpp/ldlt_order_observer.h"), "-I" + os.path.join(ROOT, "oracle"),
                       "-shared", "-o", so, os.path.join(ROOT, "oracle/oracle_capi.cc"),
                       os.path.join(ROOT, "tools/cpp/ldlt_order_observer_api.cc")])
class Study(oracle_py.Oracle):
    def __init__(self):
        self.path = so
        self.lib = C.CDLL(so)
        self.lib.fbo_last_error.restype = C.c_char_p
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nz, nl, nv = 50, 10, 100
o = Study()
p = fx.synthetic_dense_batch(n, nz, nl, nv)
newton = 0
for q in range(n):
    o.lib.fbo_obs_mark(nz)
    one = fx.DenseProblem(nz, nl, nv, {k: np.ascontiguousarray(a[q:q + 1]) for k, a in p.arrays.items()})
    out = o.solve_dense(one, nthreads=1)
    newton += int(out[4]["newton_iters"][0])
buf = (C.c_longlong * 69)()
o.lib.fbo_obs_read(buf)
f, same, first, zfirst = buf[0], buf[1], buf[2], buf[3]
print(f"{n} QPs of config 2 ({nz}/{nl}/{nv}): {f} factorisations ({newton} Newton steps), {first} first of their QP")
print(f"  order identical to the previous step's: {same} ({100.0 * same / max(1, f - first):.1f} % of the steps that have a predecessor)")
print(f"  orders that eliminate the whole leading block first: {zfirst} ({100.0 * zfirst / f:.1f} %)")
h = list(buf[4:])
cum = 0
print("  common prefix with the previous order (steps): count")
for lo, hi in ((0, 0), (1, 4), (5, 9), (10, 19), (20, 29), (30, 39), (40, 49), (50, 59), (60, 64)):
    c = sum(h[lo:hi + 1]); print(f"    {lo:2d}..{hi:2d}: {c}")
