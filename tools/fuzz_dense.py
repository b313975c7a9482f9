#!/usr/bin/env python3
"""Developer tool: random dense QP shapes (one-wavefront, four-wavefront and global-K
kernels) against the oracle.  argv: number of shapes [seed [largest nz + nl [modes]]]
(64 as the third argument keeps every shape on the one-wavefront kernel).  `modes` is a
comma-separated list of factorisation settings of the one-wavefront kernel, each run on the
same QPs against ONE oracle solve: "default" (the handle's default: pivoted), "auto" (AUTO at its default thresholds), "pivoted", "natural", or a
number = AUTO with that many spread bits (fbstab_hip_dense_set_factorisation)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbstab_amd import hip_api
from tools import fixtures as fx
from oracle.oracle_py import Oracle, default_options
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 0
modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["default"]
orc = Oracle(False)
tot = {m: dict(bad=0, ndiff=0, nprox=0, steps=0, newton=0) for m in modes}
nqp = 0
for it in range(n):
    nz = int(rng.integers(1, 160)); nl = int(rng.integers(0, min(nz, 24) + 1)); nv = int(rng.integers(1, 240))
    if kmax:
        nz = int(rng.integers(1, kmax + 1)); nl = int(rng.integers(0, min(nz, 24, kmax - nz) + 1))
    B = int(rng.integers(1, 10))
    o = default_options()
    p = fx.synthetic_dense_batch(B, nz, nl, nv, first_id=int(rng.integers(0, 1 << 20)))
    c = orc.solve_dense(p, opts=o, nthreads=orc.num_threads())
    oc = c[4]
    nqp += B
    for m in modes:
        # (developer switch: a trailing "n" = a QP that went to the pivoted path does NOT stay there)
        os.environ["FBSTAB_HIP_DENSE_STICKY"] = "0" if m.endswith("n") else "1"
        if m[0].isdigit():
            os.environ["FBSTAB_HIP_DENSE_ACT_BITS"] = m.rstrip("n").split("a")[1] if "a" in m else "0"
        else:
            os.environ.pop("FBSTAB_HIP_DENSE_ACT_BITS", None)
        s = hip_api.FBstabDenseBatch(nz, nl, nv, max_batch=B)
        if m == "pivoted":
            s.SetFactorisation(s.ORDER_PIVOTED)
        elif m == "natural":
            s.SetFactorisation(s.ORDER_NATURAL)
        elif m == "auto":
            s.SetFactorisation(s.ORDER_AUTO)
        elif m != "default":
            s.SetFactorisation(s.ORDER_AUTO, int(m.rstrip("n").split("a")[0]))
        z = np.zeros((B, nz)); l = np.zeros((B, nl)); v = np.zeros((B, nv)); y = np.zeros((B, nv))
        out = s.Solve({k: np.ascontiguousarray(a) for k, a in p.arrays.items()}, z, l, v, y)
        q = s.query(); fac = s.Factorisation(); s.close()
        dn = np.abs(out["newton_iters"].astype(int) - oc["newton_iters"].astype(int))
        okf = np.array_equal(out["eflag"], oc["eflag"]) and np.array_equal(out["prox_iters"], oc["prox_iters"])
        good = oc["eflag"] == 0
        dz = float(np.abs(z - c[0])[good].max()) if good.any() else 0.0
        # (the default / pivoted order claims the oracle's counts: strict; the opt-in orders pivot differently by the caller's choice)
        flag = "" if (okf and dn.max() <= (0 if m in ("default", "pivoted") else 2) and dz < 1e-4) else "  <-- CHECK"
        t = tot[m]
        t["bad"] += flag != ""
        t["ndiff"] += int((dn != 0).sum())
        t["nprox"] += int((out["prox_iters"] != oc["prox_iters"]).sum())
        t["steps"] += max(fac["pivoted_steps"], 0)
        t["newton"] += int(out["newton_iters"].sum())
        if flag:
            print("   device: eflag", out["eflag"].tolist(), "prox", out["prox_iters"].tolist(), "newton", out["newton_iters"].tolist(),
                  "residual", [f"{r:.2e}" for r in out["residual"]])
            print("   oracle: eflag", oc["eflag"].tolist(), "prox", oc["prox_iters"].tolist(), "newton", oc["newton_iters"].tolist(),
                  "residual", [f"{r:.2e}" for r in oc["residual"]])
        print(f"[{m}] ({nz},{nl},{nv}) B={B} threads={q['threads']} lds={q['lds_bytes']} flags_equal={okf} "
              f"dnewton_max={dn.max()} nonzero={int((dn != 0).sum())} dz={dz:.2e} pivoted_steps={fac['pivoted_steps']}{flag}")
for m in modes:
    t = tot[m]
    print(f"[{m}] shapes to check: {t['bad']}  QPs: {nqp}  QPs whose Newton count differs from the oracle's: {t['ndiff']}"
          f"  whose proximal count differs: {t['nprox']}  Newton steps handed to the pivoted path: {t['steps']} of {t['newton']}")
