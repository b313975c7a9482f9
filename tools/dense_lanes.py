"""Developer tool (GPU box): dense configs[1] against the number of launches in flight.
usage: python tools/dense_lanes.py <lanes> [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tools import fixtures as fx
from fbstab_amd import hip_api
lanes = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0")
r = bench.bench_dense(torch, dev, fx, hip_api, steps=steps, lanes=lanes)
print("lanes", lanes, "hwq", os.environ.get("GPU_MAX_HW_QUEUES", "default"), "wgs/cu", os.environ.get("FBSTAB_HIP_WGS_PER_CU", "default"),
      round(r["value"]), "QP/s", round(r["ms_per_step"], 3), "ms/step", "kernel_ms", round(r["kernel_ms"], 3), r["launch"]["workgroups"])
