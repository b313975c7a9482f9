#!/usr/bin/env python3
"""Developer tool: random MPC shapes over every record instance (and the flat-vector
kernel) against the oracle: exit flags, proximal counts equal; Newton counts equal on
all but a few.  argv: number of shapes [seed] [r16] [bounds | sparse] [warm] [only=<part of a kernel name>].  With `bounds` the constraints are bounds on single stage
variables (fixtures.random_ltv_mpc_bounds).  With `r16` every shape is drawn inside the
headline instance <12,4,20> (nx <= 12, nu <= 4, nc <= 20); a shape is flagged ("CHECK") as soon as ANY
count differs from the oracle's (strict), otherwise when flags / proximal counts differ, a Newton
count differs by more than two or the solutions part.  The last column counts the Newton steps the kernel
refined (fbstab_hip_mpc_refined_steps).  With `warm` (round 6) every shape is solved a SECOND time: x0 perturbed,
warm-started from the device's first solution - device and oracle get the same guess (the device's), so every
second solve is an independent comparison on identical inputs; its counts are compared as strictly, and the
summary line counts the warm solves' flips separately."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbstab_amd import hip_api
from tools import fixtures as fx
from oracle.oracle_py import Oracle, default_options
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
r16 = len(sys.argv) > 3 and "r16" in sys.argv[3:]
bounds = "bounds" in sys.argv[3:]  # bound constraints (one +-1 entry per row): the row form of the costate step
sparse = "sparse" in sys.argv[3:]  # two or three entries of order one per row: the row form without the bounds' shortcut
warm = "warm" in sys.argv[3:]      # a second, warm-started solve per shape (teacher-forced: the oracle gets the device's guess)
only = next((a.split("=", 1)[1] for a in sys.argv[3:] if a.startswith("only=")), None)  # only shapes whose kernel name contains this (the streams stay as they are)
strict = r16 or os.environ.get("FUZZ_STRICT", "1") != "0"
nqp = nref = 0
nwarm = warm_flips = warm_bad = warm_loose = warm_fma_agrees = warm_refine_closes = 0
wrng = np.random.default_rng(1_000_003 + (int(sys.argv[2]) if len(sys.argv) > 2 else 1))  # (the shape stream stays as it is)
orc = Oracle(False)
bad = 0
for it in range(n):
    if r16:
        nx = int(rng.integers(1, 13)); nu = int(rng.integers(1, 5)); nc = int(rng.integers(1, 21)); N = int(rng.integers(1, 33))
    else:
        nx = int(rng.integers(1, 27)); nu = int(rng.integers(1, 10)); nc = int(rng.integers(1, 34)); N = int(rng.integers(1, 13))
    B = int(rng.integers(1, 14))
    o = default_options()
    if rng.random() < 0.3:
        o = default_options(max_linesearch_iters=int(rng.integers(1, 12)), nonmonotone_linesearch=int(rng.random() < 0.5))
    p = (fx.random_ltv_mpc_bounds(rng, B, N, nx, nu, nc) if bounds else
         fx.random_ltv_mpc_sparse_rows(rng, B, N, nx, nu, nc) if sparse else fx.random_ltv_mpc(rng, B, N, nx, nu, nc))
    s = hip_api.FBstabMpcBatch(N, nx, nu, nc, max_batch=B)
    if only is not None and only not in s.kernel_name():
        s.close()
        if warm:  # (the warm family's own stream moves on as if the shape had been solved)
            wrng.standard_normal(p.arrays["x0"].shape); wrng.standard_normal(p.arrays["x0"].shape)
        continue
    h = hip_api.Options()
    for name, _ in h._fields_:
        setattr(h, name, getattr(o, name))
    s.UpdateOptions(h)
    z = np.zeros((B, p.nz)); l = np.zeros((B, p.nl)); v = np.zeros((B, p.nv)); y = np.zeros((B, p.nv))
    out = s.Solve({k: np.ascontiguousarray(a) for k, a in p.arrays.items()}, z, l, v, y)
    kn = s.kernel_name(); refined = s.refined_steps()
    wline = ""
    if warm:
        # second solve: the measured state moves (a few per cent and a small absolute kick), everything else stays
        p2 = type(p)(*p.sizes())
        p2.arrays = dict(p.arrays)
        x0 = p.arrays["x0"]
        p2.arrays["x0"] = np.ascontiguousarray(x0 * (1.0 + 0.05 * wrng.standard_normal(x0.shape)) + 0.01 * wrng.standard_normal(x0.shape))
        g = (z.copy(), l.copy(), v.copy())   # the device's first solution: BOTH warm starts
        z2, l2, v2, y2 = g[0].copy(), g[1].copy(), g[2].copy(), np.zeros((B, p.nv))
        out2 = s.Solve({k: np.ascontiguousarray(a) for k, a in p2.arrays.items()}, z2, l2, v2, y2)
        c2 = orc.solve_mpc(p2, g, opts=o, nthreads=orc.num_threads())
        oc2 = c2[4]
        dn2 = np.abs(out2["newton_iters"].astype(int) - oc2["newton_iters"].astype(int))
        dp2 = np.abs(out2["prox_iters"].astype(int) - oc2["prox_iters"].astype(int))
        ef2 = np.array_equal(out2["eflag"], oc2["eflag"])
        nwarm += B
        # QPs WITH a solution (both exit 0): every count strict.  QPs without one (an infeasibility verdict, an
        # iteration limit): flag and proximal count strict, the Newton count reported - their iterates run away
        # along the certificate's ray at norms of 1e8 and more, where the device's incrementally carried residual
        #  and the reference's freshly evaluated one part by more than an inner tolerance, and a
        # runaway subproblem ends after a handful of iterations on one side and at max_inner_iters on the other.
        conv = (oc2["eflag"] == 0) & (out2["eflag"] == 0)
        flips = int((((dn2 != 0) | (dp2 != 0)) & conv).sum())
        loose = int(((dn2 != 0) & ~conv).sum())
        warm_flips += flips
        warm_loose += loose
        wbad = (not ef2) or flips > 0 or bool(((dp2 != 0) & ~conv).any())
        if flips > 0 and ef2:
            # second opinion: the oracle built with fused multiply-adds allowed
            fm = Oracle(False, fma=True).solve_mpc(p2, g, opts=o, nthreads=orc.num_threads())[4]
            same_as_fma = ((out2["newton_iters"] == fm["newton_iters"]) & (out2["prox_iters"] == fm["prox_iters"]))[conv & ((dn2 != 0) | (dp2 != 0))]
            warm_fma_agrees += int(same_as_fma.sum())
            # ... and the device with its refinement option on (fbstab_options_t::reserved = 1): a QP on which the
            # explicit inverse of the one-row instances left more of a Newton system than the tolerance takes the
            # oracle's counts then (tests/test_gpu_components.py::test_warm_started_second_solve_...)
            h2 = hip_api.Options()
            for name, _ in h2._fields_:
                setattr(h2, name, getattr(o, name))
            h2.reserved = 1
            s.UpdateOptions(h2)
            z3, l3, v3 = g[0].copy(), g[1].copy(), g[2].copy()
            out3 = s.Solve({k: np.ascontiguousarray(a) for k, a in p2.arrays.items()}, z3, l3, v3, np.zeros((B, p.nv)))
            flipped = conv & ((dn2 != 0) | (dp2 != 0))
            warm_refine_closes += int(((out3["newton_iters"] == oc2["newton_iters"]) & (out3["prox_iters"] == oc2["prox_iters"]))[flipped].sum())
        warm_bad += wbad
        wline = (f" | warm: flags_equal={ef2} converged: prox_flips={int(((dp2 != 0) & conv).sum())} newton_flips={int(((dn2 != 0) & conv).sum())}"
                 f" no-solution QPs with another Newton count={loose} (dn_max={dn2.max()})" + ("  <-- CHECK(warm)" if wbad else ""))
        if wbad:
            print("   warm device: eflag", out2["eflag"].tolist(), "prox", out2["prox_iters"].tolist(), "newton", out2["newton_iters"].tolist(),
                  "residual", [f"{r:.2e}" for r in out2["residual"]])
            print("   warm oracle: eflag", oc2["eflag"].tolist(), "prox", oc2["prox_iters"].tolist(), "newton", oc2["newton_iters"].tolist(),
                  "residual", [f"{r:.2e}" for r in oc2["residual"]])
    s.close()
    nqp += B; nref += refined
    c = orc.solve_mpc(p, opts=o, nthreads=orc.num_threads())
    oc = c[4]
    dn = np.abs(out["newton_iters"].astype(int) - oc["newton_iters"].astype(int))
    okf = np.array_equal(out["eflag"], oc["eflag"]) and np.array_equal(out["prox_iters"], oc["prox_iters"])
    good = oc["eflag"] == 0
    dz = float(np.abs(z - c[0])[good].max()) if good.any() else 0.0
    flag = "" if (okf and dn.max() <= (0 if strict else 2) and dz < 1e-4) else "  <-- CHECK"
    bad += flag != ""
    if flag:
        print("   device: eflag", out["eflag"].tolist(), "prox", out["prox_iters"].tolist(), "newton", out["newton_iters"].tolist(),
              "residual", [f"{r:.2e}" for r in out["residual"]])
        print("   oracle: eflag", oc["eflag"].tolist(), "prox", oc["prox_iters"].tolist(), "newton", oc["newton_iters"].tolist(),
              "residual", [f"{r:.2e}" for r in oc["residual"]])
    print(f"({N},{nx},{nu},{nc}) B={B} {kn:32s} flags_equal={okf} dnewton_max={dn.max()} nonzero={int((dn != 0).sum())} dz={dz:.2e} refined={refined}{flag}{wline}")
print(f"shapes to check: {bad}   ({n} shapes, {nqp} QPs, {nref} refined Newton steps; strict={strict})"
      + (f"   warm solves: {nwarm}; QPs with a solution and any count different: {warm_flips} (of them the FMA build of the oracle takes "
         f"the device's counts on {warm_fma_agrees}, the device with refinement on takes the oracle's on {warm_refine_closes}), shapes to check {warm_bad}; QPs without a solution and another Newton count: {warm_loose}" if warm else ""))
