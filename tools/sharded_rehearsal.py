#!/usr/bin/env python3
"""Developer tool (GPU box): the C-ABI's multi-GPU entry rehearsed on the ONE device a
box has - BASELINE configs[2] (batch 8192) through fbstab_hip_mpc_solve_batch_sharded with
a one-device group, the shard's results travelling through grouped ncclSend / ncclRecv
(FBSTAB_HIP_SHARD_SELF_SEND=1) - next to the plain solve_batch call.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

os.environ["FBSTAB_HIP_SHARD_SELF_SEND"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from fbstab_amd import hip_api  # noqa: E402
from tools import fixtures as fx  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
p = fx.synthetic_mpc_batch(B)
data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
z = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=B)
g = hip_api.ShardGroup([0])
out = torch.zeros((B, 40), dtype=torch.uint8, device=dev)
root_out = torch.zeros((B, 40), dtype=torch.uint8, device=dev)
root = (z(p.nz), z(p.nl), z(p.nv), z(p.nv))
t_plain, t_shard = [], []
for rep in range(4):
    x = (z(p.nz), z(p.nl), z(p.nv), z(p.nv))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = hip_api.out_to_numpy(s.Solve(data, *x))
    torch.cuda.synchronize()
    t_plain.append(time.perf_counter() - t0)
    y = (z(p.nz), z(p.nl), z(p.nv), z(p.nv))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.Solve([s], [data], [y], [out], 0, root, root_out)
    torch.cuda.synchronize()
    t_shard.append(time.perf_counter() - t0)
    same = all(torch.equal(a, b) for a, b in zip(root, x))
o = hip_api.out_to_numpy(root_out)
print(json.dumps({
    "what": "fbstab_hip_mpc_solve_batch_sharded, group of one device, results through grouped ncclSend/ncclRecv to self",
    "batch": B, "plain_ms": round(1e3 * min(t_plain[1:]), 3), "sharded_ms": round(1e3 * min(t_shard[1:]), 3),
    "gathered_bytes": int(B * (8 * (p.nz + p.nl + 2 * p.nv) + 40)), "bitwise_equal": bool(same),
    "newton_sum": int(o["newton_iters"].sum()), "all_converged": bool((o["eflag"] == 0).all()), "group": g.stats()}))
