#!/usr/bin/env python3
"""Developer tool (GPU box): do eight launches of the record kernel on eight streams share the chip, and does a
small kernel queued in front of each launch decide it?  (LABNOTES R6.3: a wavefront that allocates 504 of a SIMD's
512 registers leaves no room for any other kernel's wavefront.)  16 steps over 8 lanes like bench.py, the guess
zeroed (a) by torch's fill kernel on the lane's stream before every solve - what bench.py does - or (b) not at all
inside the timed region: every step gets buffers of its own, zeroed beforehand, so the same cold solves run with
NO other kernel queued between them.  usage: FBSTAB_HIP_LIB=... tools/overlap_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from fbstab_amd import hip_api
from tools import fixtures as fx
B, P, STEPS = 8192, 8, 16
dev = torch.device("cuda:0")
p = fx.synthetic_mpc_batch(B)
data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
lanes = []
for _ in range(P):
    s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=B, handles_in_flight=P)
    mk = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
    lanes.append(dict(s=s, st=torch.cuda.Stream(device=dev), z=mk(p.nz), l=mk(p.nl), v=mk(p.nv), y=mk(p.nv),
                      out=torch.zeros((B, 40), dtype=torch.uint8, device=dev)))
fresh = [[torch.zeros((B, n), dtype=torch.float64, device=dev) for n in (p.nz, p.nl, p.nv)] for _ in range(STEPS)]
def run(zero, in_flight):
    def step(k, timed):
        ln = lanes[k % in_flight]
        with torch.cuda.stream(ln["st"]):
            z, l, v = ln["z"], ln["l"], ln["v"]
            if zero or not timed:
                for a in (z, l, v):
                    a.zero_()
            else:
                z, l, v = fresh[k]
            ln["s"].Solve(data, z, l, v, ln["y"], out=ln["out"], stream=ln["st"].cuda_stream, async_=True)
    for k in range(P):
        step(k, False)
    for f in fresh:
        for a in f:
            a.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(STEPS):
        step(k, True)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / STEPS
print("library", hip_api.current_library_path())
for zero in (True, False):
    one = run(zero, 1)
    eight = run(zero, P)
    print(f"fill kernels between the solves: {zero!s:5s}  ms per step, one lane {one:7.2f}   eight lanes {eight:7.2f}   ratio {one / eight:4.2f}")
