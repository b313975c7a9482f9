#!/usr/bin/env python3
"""Developer tool (GPU box): bench.py's dense block (BASELINE configs[1], eight launches in flight and one at a time, the
default order and the two opt-in ones) for ONE library (FBSTAB_HIP_LIB) - run it per library, interleaved, for an A/B.
usage: FBSTAB_HIP_LIB=... tools/dense_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from fbstab_amd import hip_api
from tools import fixtures as fx
r = bench.bench_dense(torch, torch.device("cuda:0"), fx, hip_api)
o = r["opt_in_orders"]
print(f"{os.path.basename(hip_api.current_library_path()):22s} default order: in flight {r['value']:9.0f} QP/s  one at a time {r['serial_value']:9.0f}  kernel {r['kernel_ms']:.3f} ms"
      f"  newton {r['mean_newton_iters']:.4f} ok={r['all_converged']} | auto {o['auto']['value']:9.0f} / {o['auto']['serial_value']:9.0f} | natural {o['natural']['value']:9.0f} / {o['natural']['serial_value']:9.0f}"
      f" | lds {r['launch']['lds_bytes']} wgs {r['launch']['workgroups']}")
