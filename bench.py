#!/usr/bin/env python3
"""Headline benchmark: QPs/sec of the batched FBstab MPC solve
(BASELINE.json: N=30, nx=12, nu=4, nc=20; config 3 = batch 8192 per GPU).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one fbstab_hip_mpc_solve_batch call over the rank's whole shard
(cold start, zero initial guess), problem data already resident in HBM.  For
N > 1 the batch is sharded by global instance id (weak scaling, 8192 QPs per
GPU) and each step ends with one RCCL gather of the solutions to rank 0.
Consecutive steps are issued on --pipeline (default 8, one per hardware queue) alternating HIP streams,
each with its own solver handle and output buffers, the way a stream of
batches is served: iteration counts differ 10x between QPs, so the tail of
one batch (a few slow QPs) overlaps the bulk of the next.  --pipeline 1
serialises the steps.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# One hardware queue per pipeline stream (HIP maps streams onto a small pool of
# queues; two streams on one queue cannot overlap their kernels).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ALG_BYTES_PER_QP = 217736      # data 188,928 + guess 11,904 + solution 16,864 + SolverOut 40
FLOP_PER_NEWTON = 0.94e6       # SURVEY.md 8(d), survey-derived flop model
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # AMD public MI355X FP64 vector/matrix spec


def cpu_baseline(sample_qps: int):
    """The oracle (CPU restatement, -O3) on a bounded sample of the same
    workload: OpenMP-over-batch on all host cores, plus one thread."""
    from fbstab_amd import fixtures as fx
    from oracle.oracle_py import Oracle
    orc = Oracle(False)
    cores = orc.num_threads()
    p = fx.synthetic_mpc_batch(sample_qps)
    orc.solve_mpc(fx.synthetic_mpc_batch(2 * cores), nthreads=cores)  # warm threads
    t0 = time.perf_counter()
    out = orc.solve_mpc(p, nthreads=cores)[4]
    t_omp = time.perf_counter() - t0
    n1 = max(16, sample_qps // max(cores, 1) // 2)
    p1 = fx.synthetic_mpc_batch(n1)
    t0 = time.perf_counter()
    orc.solve_mpc(p1, nthreads=1)
    t_1 = time.perf_counter() - t0
    return {
        "value": sample_qps / t_omp, "unit": "QPs/sec", "cores": cores, "kind": "port",
        "sample": f"{sample_qps} QPs of the same synthetic MPC workload (ids 0..{sample_qps - 1}), "
                  f"OpenMP schedule(dynamic) over the batch, {t_omp:.1f} s; "
                  f"single thread: {n1} QPs in {t_1:.1f} s",
        "single_thread_value": n1 / t_1,
        "mean_newton_iters": float(out["newton_iters"].mean()),
    }


def pmc_traffic(batch: int):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary
    (FETCH_SIZE and WRITE_SIZE, separate passes; see the note in the file).
    bench.py cannot run the profiler on itself, so this is the figure of the
    last profiled build for the same batch, or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("batch") == batch:
                return t["hbm_bytes_per_launch_raw"]
        except (OSError, ValueError, KeyError):
            pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8192, help="QPs per GPU")
    ap.add_argument("--pipeline", type=int, default=8,
                    help="steps in flight (alternating streams/handles); 1 = serial")
    ap.add_argument("--first-id", type=int, default=None,
                    help="developer option: first global instance id of this rank's shard "
                         "(default: rank * batch, the shard bench runs under torch.distributed.run)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="QPs for the CPU baseline (0 disables; default sized for ~15 s)")
    args = ap.parse_args()

    import torch
    from fbstab_amd import fixtures as fx
    from fbstab_amd import hip_api

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # under torch.distributed.run the process group (RCCL) is set up even for one
    # rank, so `--nproc-per-node 1` exercises the same gather path as N > 1
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    B = args.batch
    P = max(1, args.pipeline)
    from fbstab_amd import sharding
    first_id, _ = sharding.shard_range(rank, world, B)
    if args.first_id is not None:
        first_id = args.first_id
    p = fx.synthetic_mpc_batch(B, first_id=first_id)  # shard by global instance id
    data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
    nvar = p.nz + p.nl + 2 * p.nv

    class Lane:
        """One pipeline lane: solver handle (own device scratch), stream, outputs."""

        def __init__(self):
            self.solver = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=B, device=local_rank)
            self.stream = torch.cuda.Stream(device=dev)
            # z, l, v, y side by side in one record per QP so the gather is one buffer
            self.x = torch.zeros((B, nvar), dtype=torch.float64, device=dev)
            self.z, self.l = self.x[:, :p.nz], self.x[:, p.nz:p.nz + p.nl]
            self.v = self.x[:, p.nz + p.nl:p.nz + p.nl + p.nv]
            self.y = self.x[:, p.nz + p.nl + p.nv:]
            self.out = torch.zeros((B, 40), dtype=torch.uint8, device=dev)
            self.gx = self.go = None
            if dist is not None and rank == 0:
                self.gx = [torch.empty_like(self.x) for _ in range(world)]
                self.go = [torch.empty_like(self.out) for _ in range(world)]
            self.events = []

    lanes = [Lane() for _ in range(P)]

    def step(k, timed):
        ln = lanes[k % P]
        with torch.cuda.stream(ln.stream):
            ln.x.zero_()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(ln.stream)
            ln.solver.Solve(data, ln.z, ln.l, ln.v, ln.y, out=ln.out,
                            stream=ln.stream.cuda_stream, async_=True)
            e1.record(ln.stream)
            if timed:
                ln.events.append((e0, e1))
            if dist is not None:
                dist.gather(ln.x, ln.gx, dst=0)
                dist.gather(ln.out, ln.go, dst=0)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k, False)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, True)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [e0.elapsed_time(e1) for ln in lanes for (e0, e1) in ln.events]
    solver = lanes[0].solver
    out = lanes[(args.steps - 1) % P].out

    o = hip_api.out_to_numpy(out)
    ok = bool((o["eflag"] == 0).all())
    if rank == 0:
        total_qps = world * B * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms))
        achieved = ALG_BYTES_PER_QP * B / (k_ms * 1e-3) / 1e9
        mean_newton = float(o["newton_iters"].mean())
        rec = {
            "metric": "QPs/sec (batched MPC N=30 nx=12 nu=4 nc=20)",
            "value": total_qps, "unit": "QPs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: batched FBstabMpc, batch=8192 per GPU, "
                                   "N=30 nx=12 nu=4 nc=20, cold start, default options",
                       "batch_per_gpu": B, "global_batch": world * B, "steps_in_flight": P,
                       "parallelism": f"batch sharded over {world} GPU(s)" +
                                      (", RCCL gather to rank 0" if dist is not None else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(B),
                         "kernel": "fbstab_mpc_r16_kernel<12,4,20>", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_QP * B,
                         # `achieved` divides by the duration of one launch as rocprof sees it; with
                         # P launches sharing the GPU that duration is ~P times the time the GPU
                         # spends per batch, which ms_per_step measures:
                         "launches_in_flight": P,
                         "achieved_per_step": ALG_BYTES_PER_QP * B * world / (elapsed / args.steps) / 1e9,
                         # what the kernel actually moves (PMC bytes of one launch) per second of this
                         # run, per GPU: the figure to hold against the HBM peak
                         "traffic_per_step_GBps": (pmc_traffic(B) or 0.0) / (elapsed / args.steps) / 1e9 or None},
            "fp64": {"model_flop_per_newton_iter": FLOP_PER_NEWTON,
                     "mean_newton_iters": mean_newton,
                     # (rate of the whole GPU: flops of one batch over the time per step)
                     "achieved_tflops": FLOP_PER_NEWTON * mean_newton * B * world / (elapsed / args.steps) / 1e12 / world,
                     "peak_tflops": FP64_PEAK_TFLOPS},
            "all_converged": ok,
            "launch": solver.query(),
        }
        n_cpu = args.cpu_sample
        if n_cpu < 0:
            n_cpu = 256 * max(1, (os.cpu_count() or 1))
        if n_cpu > 0 and world == 1:
            rec["cpu_baseline"] = cpu_baseline(n_cpu)
        elif world > 1:
            rec["cpu_baseline"] = None
        print(json.dumps(rec))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
