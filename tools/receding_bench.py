#!/usr/bin/env python3
"""Developer tool: BASELINE config 5 in short form on one GPU - T closed-loop
trajectories x S warm-started MPC steps (N=30, nx=12, nu=4, nc=20), problem
data resident on the device, only x0 changing (tests/closed_loop.py).
argv: trajectories steps [retire] [keep]

With "retire", a trajectory whose QP did not return SUCCESS (the closed loop
has no terminal constraint, so a few run into infeasible states) is parked at
the origin with a zero guess from then on - what a controller's fallback would
do - so that the step time shows the warm-started bulk rather than the handful
of QPs that run to the iteration limit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from fbstab_amd import hip_api  # noqa: E402
from tests import closed_loop as rh  # noqa: E402
from tools import fixtures as fx  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
RETIRE = "retire" in sys.argv[3:]
KEEP = "keep" in sys.argv[3:]  # FBSTAB_HIP_KEEP_MATRICES: the matrices do not change between steps
STAMPS = "stamps" in sys.argv[3:]  # -DFB_CLOCKSTAMP build: where the wavefronts' cycles go from step 20 on
p = fx.synthetic_mpc_batch(T)
N, nx, nu, nc = p.sizes()
A, B = fx.quadrotor_model()
dev = torch.device("cuda:0")
s = hip_api.FBstabMpcBatch(N, nx, nu, nc, max_batch=T)
data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
mk = lambda n: torch.zeros((T, n), dtype=torch.float64, device=dev)
y = mk(p.nv)
out_dev = torch.zeros((T, 40), dtype=torch.uint8, device=dev)


retired = torch.zeros(T, dtype=torch.bool, device=dev)
stream = torch.cuda.Stream(dev)  # a real stream: 0 would select the handle's own
events = []


def solve(x0, z, l, v):
    if STAMPS and len(events) == 20:
        import ctypes
        torch.cuda.synchronize()
        hip_api.load_library().fbstab_hip_debug_stamps((ctypes.c_ulonglong * 32)(), 1)
    # nothing here waits for the device: the retire mask, the plant update in
    # closed_loop() and the next launch are all queued behind the solve
    if RETIRE:
        m = retired.unsqueeze(1)
        for a in (x0, z, l, v):
            a.masked_fill_(m, 0.0)
    data["x0"] = x0.contiguous()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    s.Solve(data, z, l, v, y, out=out_dev, keep_matrices=KEEP, stream=stream.cuda_stream, async_=True)
    e1.record(stream)
    events.append((e0, e1))
    if RETIRE:
        retired.logical_or_(out_dev.view(torch.int32)[:, 0] != 0)
    return z, l, v, y, out_dev.clone()


x0 = data["x0"].clone()
Ad, Bd = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
z0, l0, v0 = mk(p.nz), mk(p.nl), mk(p.nv)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(stream):  # plant update, masks and solves all queue on this stream
    log = rh.closed_loop(solve, x0, z0, l0, v0, Ad, Bd, nx, nu, S)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
kernel_ms = [a.elapsed_time(b) for a, b in events]
for r in log:
    r["out"] = hip_api.out_to_numpy(r["out"])
it = np.array([r["out"]["newton_iters"].mean() for r in log])
ok = all((r["out"]["eflag"] == 0).all() for r in log)
for k in (0, 1, 2, 5, 10, S - 1):
    if k < S:
        o = log[k]["out"]
        print(f"   step {k:3d}: eflag counts {np.bincount(o['eflag'], minlength=6).tolist()} newton mean {o['newton_iters'].mean():.2f} "
              f"max {o['newton_iters'].max()} prox max {o['prox_iters'].max()} kernel {kernel_ms[k]:.2f} ms")
print(f"trajectories={T} steps={S}: {T * S / dt:.0f} QP/s wall ({dt / S * 1e3:.2f} ms per closed-loop step), "
      f"kernel ms first/median/last {kernel_ms[0]:.2f}/{np.median(kernel_ms):.2f}/{kernel_ms[-1]:.2f}, "
      f"mean Newton iterations first/last step {it[0]:.2f}/{it[-1]:.2f}, all converged {ok}, "
      f"retired {int(retired.sum())}")

if STAMPS:
    import ctypes
    st = (ctypes.c_ulonglong * 32)()
    hip_api.load_library().fbstab_hip_debug_stamps(st, 1)
    tot = float(st[28]) or 1.0
    print(f"steps 20..{S - 1}: sum of wave lifetimes {st[29] * 1e-5:.1f} ms")
    for k, nm, cnt in ((19, "load_guess", 0), (22, "open_prox", st[25]), (20, "newton_step", st[27]),
                       (23, "norms_at_multi", st[24]), (21, "close_subproblem", st[26])):
        print(f"   wave cycles in {nm:18s} {100.0 * st[k] / tot:5.1f} %" + (f"  ({cnt} calls)" if cnt else ""))
