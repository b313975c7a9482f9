#!/usr/bin/env python3
"""Developer tool (GPU box): one Newton step of the boxed BASELINE plant through the probe
entry point (fbstab_hip_mpc_debug_newton) under two libraries; prints which outputs differ.
argv: libA libB"""
import os, subprocess, sys, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) == 3:
    res = []
    for lib in sys.argv[1:3]:
        out = subprocess.run([sys.executable, __file__, lib, "--child", "/tmp/_probe.pkl"], env=dict(os.environ, FBSTAB_HIP_LIB=lib),
                             capture_output=True, text=True)
        if out.returncode:
            print(out.stderr[-2000:]); sys.exit(1)
        res.append(pickle.load(open("/tmp/_probe.pkl", "rb")))
    a, b = res
    for k in a:
        if isinstance(a[k], np.ndarray) and a[k].size:
            d = np.abs(a[k] - b[k])
            print(f"{k:8s} max|A| {np.abs(a[k]).max():.3e}  max|A-B| {d.max():.3e}  at {int(d.argmax())} of {a[k].size}  nan A/B {int(np.isnan(a[k]).sum())}/{int(np.isnan(b[k]).sum())}")
        else:
            print(k, a[k], b[k])
    sys.exit(0)
from tools import fixtures as fx
from fbstab_amd import hip_api as hip
p = fx.boxed_mpc_batch(1)
s = hip.FBstabMpcBatch(*p.sizes(), max_batch=1)
data = {k: a[0] for k, a in p.arrays.items()}
rng = np.random.default_rng(3)
z = 0.1 * rng.standard_normal(p.nz); l = 0.1 * rng.standard_normal(p.nl); v = np.abs(0.1 * rng.standard_normal(p.nv))
zb = z + 0.01 * rng.standard_normal(p.nz); lb = l + 0.01 * rng.standard_normal(p.nl); vb = np.abs(v + 0.01 * rng.standard_normal(p.nv))
g = s.debug_newton(data, z, l, v, zb, lb, vb)
pickle.dump({k: (np.asarray(x) if not np.isscalar(x) else x) for k, x in g.items()}, open(sys.argv[3], "wb"))
