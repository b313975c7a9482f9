#!/usr/bin/env python3
"""Developer tool (GPU box): wall ms of ONE cold solve of a small batch of the bench line's QPs on the record
kernel and on the flat-vector kernel (FBSTAB_HIP_GENERIC=1, read when the handle is created) - the numbers
behind the bench line's `latency` block.  argv: batch sizes (default 1 4 16)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbstab_amd import hip_api
from tools import fixtures as fx
dev = torch.device("cuda:0")
batches = [int(a) for a in sys.argv[1:]] or [1, 4, 16]
for b in batches:
    p = fx.synthetic_mpc_batch(b)
    data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
    for generic in ("0", "1"):
        os.environ["FBSTAB_HIP_GENERIC"] = generic
        s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=b, device=0)
        mk = lambda n: torch.zeros((b, n), dtype=torch.float64, device=dev)
        z, l, v, y = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
        ms, kms = [], []
        for k in range(8):
            for a in (z, l, v):
                a.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = s.Solve(data, z, l, v, y)
            torch.cuda.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0)); kms.append(s.last_kernel_ms())
        o = hip_api.out_to_numpy(out)
        print(f"batch {b:4d} {s.kernel_name():34s} wall median {np.median(ms[1:]):7.3f} ms  min {np.min(ms[1:]):7.3f}  kernel {np.median(kms[1:]):7.3f} ms"
              f"  newton max {int(o['newton_iters'].max())}  converged {bool((o['eflag'] == 0).all())}", flush=True)
        s.close()
os.environ.pop("FBSTAB_HIP_GENERIC", None)
