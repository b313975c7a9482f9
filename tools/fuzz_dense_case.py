#!/usr/bin/env python3
"""Developer tool (GPU box): ONE shape of the dense fuzz stream (tools/fuzz_dense.py <n> <seed> <kmax>) looked at closely -
device (default order) against the oracle and against the oracle built with fused multiply-adds allowed, per QP:
counts, residuals, and the residual after every Newton count the two disagree on.  argv: seed kmax shape-index."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbstab_amd import hip_api
from tools import fixtures as fx
from oracle.oracle_py import Oracle, default_options

seed, kmax, want = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
for it in range(want + 1):
    nz = int(rng.integers(1, 160)); nl = int(rng.integers(0, min(nz, 24) + 1)); nv = int(rng.integers(1, 240))
    if kmax:
        nz = int(rng.integers(1, kmax + 1)); nl = int(rng.integers(0, min(nz, 24, kmax - nz) + 1))
    B = int(rng.integers(1, 10))
    first = int(rng.integers(0, 1 << 20))
o = default_options()
p = fx.synthetic_dense_batch(B, nz, nl, nv, first_id=first)
print(f"shape ({nz},{nl},{nv}) B={B} first_id={first}")
res = {}
for name, orc in (("oracle", Oracle(False)), ("oracle_fma", Oracle(False, fma=True))):
    res[name] = orc.solve_dense(p, opts=o)
s = hip_api.FBstabDenseBatch(nz, nl, nv, max_batch=B)
z = np.zeros((B, nz)); l = np.zeros((B, nl)); v = np.zeros((B, nv)); y = np.zeros((B, nv))
out = s.Solve({k: np.ascontiguousarray(a) for k, a in p.arrays.items()}, z, l, v, y)
s.close()
for name in ("oracle", "oracle_fma"):
    oc = res[name][4]
    print(f"{name:11s} eflag {oc['eflag'].tolist()} prox {oc['prox_iters'].tolist()} newton {oc['newton_iters'].tolist()} residual {[f'{r:.3e}' for r in oc['residual']]}")
print(f"{'device':11s} eflag {out['eflag'].tolist()} prox {out['prox_iters'].tolist()} newton {out['newton_iters'].tolist()} residual {[f'{r:.3e}' for r in out['residual']]}")
for name in ("oracle", "oracle_fma"):
    oc = res[name][4]
    same = bool(np.array_equal(out["eflag"], oc["eflag"]) and np.array_equal(out["prox_iters"], oc["prox_iters"])
                and np.array_equal(out["newton_iters"], oc["newton_iters"]))
    print(f"device counts equal to {name}: {same};  max |z - z_{name}| = {np.abs(z - res[name][0]).max():.3e}")
