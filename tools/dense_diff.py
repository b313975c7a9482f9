"""Developer tool (GPU box): dense shape (nz, nl, nv) solved by the library named in
FBSTAB_HIP_LIB against the oracle: which variables differ, on which QPs.
usage: python tools/dense_diff.py nz nl nv first_id [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import fixtures as fx
from fbstab_amd import hip_api
from oracle.oracle_py import Oracle
from tests.test_gpu_parity import _solve_dense_host, default_options
nz, nl, nv, fid = (int(a) for a in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 48
p = fx.synthetic_dense_batch(B, nz, nl, nv, first_id=fid)
o = default_options()
orc = Oracle()
gpu = _solve_dense_host(hip_api, p, o)
cpu = orc.solve_dense(p, opts=o, nthreads=orc.num_threads())
for name, g, c in zip("zlvy", gpu[:4], cpu[:4]):
    if c.size:
        d = np.abs(g - c).max(axis=1)
        print(name, "max diff", d.max(), "at QP", int(d.argmax()), "QPs over 1e-6:", int((d > 1e-6).sum()))
og, oc = gpu[4], cpu[4]
print("newton gpu", og["newton_iters"].tolist())
print("newton cpu", oc["newton_iters"].tolist())
print("residual gpu", og["residual"].max(), "cpu", oc["residual"].max())
