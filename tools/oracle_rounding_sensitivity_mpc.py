#!/usr/bin/env python3
"""Developer tool (CPU only): the ORACLE's own iteration counts on the MPC shape fuzz against rounding -
the restatement compiled a second time with fused multiply-adds allowed (-ffp-contract=fast -mfma; the
checked-in build uses -ffp-contract=off, as the reference's CMake build does) into gpurun_out/ (scratch).
Same algorithm, same order of operations; only the rounding of a*b+c differs.  The shapes are
tools/fuzz_shapes.py's stream for the seed (and its `bounds` / `sparse` family).  argv: number of shapes [seed] [bounds | sparse]"""
import ctypes as C
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import fixtures as fx
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 42)
gen = fx.random_ltv_mpc_bounds if "bounds" in sys.argv[3:] else fx.random_ltv_mpc_sparse_rows if "sparse" in sys.argv[3:] else fx.random_ltv_mpc
nqp = nprox = nnewton = nflag = 0
for it in range(n):
    nx = int(rng.integers(1, 27)); nu = int(rng.integers(1, 10)); nc = int(rng.integers(1, 34)); N = int(rng.integers(1, 13))
    B = int(rng.integers(1, 14))
    o = default_options()
    if rng.random() < 0.3:
        o = default_options(max_linesearch_iters=int(rng.integers(1, 12)), nonmonotone_linesearch=int(rng.random() < 0.5))
    p = gen(rng, B, N, nx, nu, nc)
    oa = a.solve_mpc(p, opts=o, nthreads=a.num_threads())[4]
    ob = b.solve_mpc(p, opts=o, nthreads=a.num_threads())[4]
    dn = oa["newton_iters"].astype(int) - ob["newton_iters"].astype(int)
    dp = oa["prox_iters"].astype(int) - ob["prox_iters"].astype(int)
    nqp += B; nprox += int((dp != 0).sum()); nnewton += int((dn != 0).sum()); nflag += int((oa["eflag"] != ob["eflag"]).sum())
    if dp.any() or dn.any():
        print(f"shape {it} ({N},{nx},{nu},{nc}) B={B}: newton {oa['newton_iters'].tolist()} | {ob['newton_iters'].tolist()}  prox {oa['prox_iters'].tolist()} | {ob['prox_iters'].tolist()}")
        print("     residual", [f"{r:.2e}" for r in oa["residual"]], "|", [f"{r:.2e}" for r in ob["residual"]])
print(f"{n} shapes, {nqp} QPs: exit flags differ on {nflag}, proximal counts on {nprox}, Newton counts on {nnewton}")
