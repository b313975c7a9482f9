import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from tools import fixtures as fx
from fbstab_amd import hip_api as hip
from oracle.oracle_py import Oracle, default_options
orc = Oracle(False)
for problem in ("ServoMotor", "SpacecraftRelativeMotion", "DoubleIntegrator"):
    gen = fx.OcpGenerator(); getattr(gen, problem)(); one = gen.GetFBstabInput()
    N, nx, nu, nc = one.sizes(); B = 8
    rng = np.random.default_rng(11)
    p = fx.MpcProblem(N, nx, nu, nc)
    p.arrays = {k: np.ascontiguousarray(np.broadcast_to(a, (B, a.shape[1]))).copy() for k, a in one.arrays.items()}
    p.arrays["x0"] = p.arrays["x0"] * (1.0 + 0.3 * rng.standard_normal((B, nx)))
    s = hip.FBstabMpcBatch(N, nx, nu, nc, max_batch=B)
    z = np.zeros((B, p.nz)); l = np.zeros((B, p.nl)); v = np.zeros((B, p.nv)); y = np.zeros((B, p.nv))
    out = s.Solve(p.arrays, z, l, v, y)
    zo, lo, vo, yo, oo = orc.solve_mpc(p, opts=default_options())
    print(problem, os.environ.get("FBSTAB_HIP_LIB"), s.kernel_name())
    print("  gpu newton", out["newton_iters"], "prox", out["prox_iters"], "res", out["residual"])
    print("  cpu newton", oo["newton_iters"], "prox", oo["prox_iters"], "res", oo["residual"])
    print("  |z-zo|", np.abs(z - zo).max(), "|l-lo|", np.abs(l-lo).max(), "|v-vo|", np.abs(v-vo).max())
    s.close()
