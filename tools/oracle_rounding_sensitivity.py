#!/usr/bin/env python3
"""Developer tool (CPU only): how stable are the ORACLE's own iteration counts against
rounding?  The restatement is compiled a second time with fused multiply-adds allowed
(-ffp-contract=fast -mfma; the checked-in build uses -ffp-contract=off, as the reference's
CMake build does) into gpurun_out/ - scratch, never loaded by anything else - and both
builds solve the dense shapes tools/fuzz_dense.py draws for the same seed.  Same algorithm,
same pivoting, same order of operations; only the rounding of a*b+c differs.
argv: number of shapes [seed [largest nz + nl]]"""
import ctypes as C
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import fixtures as fx
from oracle import oracle_py
from oracle.oracle_py import Oracle, default_options

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "gpurun_out", "oracle_fma")
os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "liboracle_fma.so")
subprocess.check_call(["g++", "-O3", "-std=c++11", "-fPIC", "-fopenmp", "-ffp-contract=fast", "-mfma", "-shared",
                       "-o", so, os.path.join(ROOT, "oracle", "oracle_capi.cc")])
a = Oracle(False)
b = Oracle(False)
b.lib = C.CDLL(so)
b.lib.fbo_last_error.restype = C.c_char_p

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nqp = nprox = nnewton = nflag = 0
dmax = 0
for it in range(n):
    nz = int(rng.integers(1, 160)); nl = int(rng.integers(0, min(nz, 24) + 1)); nv = int(rng.integers(1, 240))
    if kmax:
        nz = int(rng.integers(1, kmax + 1)); nl = int(rng.integers(0, min(nz, 24, kmax - nz) + 1))
    B = int(rng.integers(1, 10))
    p = fx.synthetic_dense_batch(B, nz, nl, nv, first_id=int(rng.integers(0, 1 << 20)))
    oa = a.solve_dense(p, opts=default_options(), nthreads=a.num_threads())[4]
    ob = b.solve_dense(p, opts=default_options(), nthreads=a.num_threads())[4]
    dn = np.abs(oa["newton_iters"].astype(int) - ob["newton_iters"].astype(int))
    dp = oa["prox_iters"] != ob["prox_iters"]
    nqp += B; nprox += int(dp.sum()); nnewton += int((dn != 0).sum()); nflag += int((oa["eflag"] != ob["eflag"]).sum())
    dmax = max(dmax, int(dn.max()))
    if dp.any() or dn.any():
        print(f"({nz},{nl},{nv}) B={B}: newton {oa['newton_iters'].tolist()} | {ob['newton_iters'].tolist()}  prox {oa['prox_iters'].tolist()} | {ob['prox_iters'].tolist()}")
print(f"{n} shapes, {nqp} QPs: exit flags differ on {nflag}, proximal counts on {nprox}, Newton counts on {nnewton} (largest difference {dmax})")
