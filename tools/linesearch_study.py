#!/usr/bin/env python3
"""Developer study (CPU, oracle only): how many step lengths the line search rejects per Newton step on
the BASELINE MPC workload (the record kernel evaluates the first trial inside the backward sweep and
every further one in a pass of its own over the records).  usage: tools/linesearch_study.py [n_qps]"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
o = Study()
p = fx.synthetic_mpc_batch(n)
out = o.solve_mpc(p, nthreads=1)[4]
h = (C.c_longlong * 32)()
o.lib.fbo_obs_read_linesearch(h)
h = list(h)
steps = sum(h)
print(f"{n} QPs of the BASELINE MPC workload: {steps} Newton steps ({int(out['newton_iters'].sum())} reported), "
      f"{sum(i * c for i, c in enumerate(h))} rejected trials = {sum(i * c for i, c in enumerate(h)) / steps:.2f} per step")
print("  rejected step lengths per Newton step: share of steps (share of all rejected trials)")
rej = sum(i * c for i, c in enumerate(h))
for i, c in enumerate(h):
    if c:
        print(f"    {i:2d}: {100.0 * c / steps:5.1f} %  ({100.0 * i * c / max(rej, 1):5.1f} %)")
