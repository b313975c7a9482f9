#!/usr/bin/env python3
"""Headline benchmark: QPs/sec of the batched FBstab MPC solve
(BASELINE.json: N=30, nx=12, nu=4, nc=20; configs[2] = batch 8192 per GPU).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one fbstab_hip_mpc_solve_batch call over the rank's whole shard
(cold start, zero initial guess), problem data already resident in HBM.  For
N > 1 the batch is sharded by global instance id (weak scaling, 8192 QPs per
GPU) and each step ends with ONE RCCL gather of the solution records to rank 0
(fbstab_amd/sharding.py).  Consecutive steps are issued on --pipeline (default 8,
one per hardware queue) HIP streams, each with its own solver handle and output
buffers, the way a stream of batches is served: iteration counts differ 10x
between QPs, so the tail of one batch (a few slow QPs) overlaps the bulk of the
next.  Rank 0 prints ONE JSON line.

Besides the contract's keys the line carries, measured in the same run at N = 1:
  serial          the same steps one at a time (--pipeline 1): latency of a batch
  ltv_dense_rows  the same shape with per-stage-distinct matrices and dense
                  constraint rows (no matrix copy shared between stages, no
                  bound-constraint path in the kernel)
  dense           BASELINE configs[1]: batched FBstabDense, batch 4096, 50/10/100
  receding        BASELINE configs[4]: 4096 warm-started closed-loop trajectories
                  x 200 steps, plant step and retirement on the device
  latency         ms per COLD solve of small batches (1, 16, 256, 2048) through
                  fbstab_hip_mpc_solve_batch with device pointers, and of ONE
                  FBstabMpc::Solve through the C++ facade (host pointers, Display::OFF) -
                  the reference's only entry (fbstab/fbstab_mpc.h:181-195) - beside the
                  CPU restatement's single-thread ms per QP
  wide            the row-pair record instances (stage widths 17..32), one launch at a time: 2048 random
                  time-varying QPs of (30, 20, 6, 16) on <24,8,16>, the reference's reactor N = 80 on <18,5,10>
  cpu_baseline    the oracle on the host cores (bounded sample)
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
DENSE_ALG_BYTES_PER_QP = 8160 * 8 + 160 * 8 + 260 * 8 + 40   # data + guess (z,l,v) + solution + SolverOut
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_ACHIEVABLE_BPS = 6.29e12    # MI355X_MICROARCH.md: what a streaming kernel measures on this part
FP64_PEAK_TFLOPS = 78.6        # AMD public MI355X FP64 vector/matrix spec


def cpu_baseline(sample_qps: int):
    """The oracle (CPU restatement, -O3) on a bounded sample of the same
    workload: OpenMP-over-batch on the host cores the process is granted (`cores` = the
    threads used), plus one thread."""
    from tools import fixtures as fx
    from oracle.oracle_py import Oracle
    orc = Oracle(False)
    cores = orc.num_threads()
    # (a cgroup that grants fewer CPUs than the box shows: more threads than the grant only take turns)
    quota = cgroup_cpu_quota()
    if quota is not None and quota >= 1.0:
        cores = max(1, min(cores, int(quota + 0.999)))
    cores = max(1, min(cores, len(os.sched_getaffinity(0))))
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
        # what the box actually grants the process: `cores` threads may share fewer CPUs
        "affinity_cpus": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": cgroup_cpu_quota(),
        "os_cpu_count": os.cpu_count(),
        "sample": f"{sample_qps} QPs of the same synthetic MPC workload (ids 0..{sample_qps - 1}), "
                  f"OpenMP schedule(dynamic) over the batch, {t_omp:.1f} s; "
                  f"single thread: {n1} QPs in {t_1:.1f} s",
        "single_thread_value": n1 / t_1,
        "mean_newton_iters": float(out["newton_iters"].mean()),
    }


def cgroup_cpu_quota():
    """CPUs' worth of time the cgroup grants (cpu.max quota / period), or None."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                t = f.read().split()
            if path.endswith("cpu.max"):
                return None if t[0] == "max" else float(t[0]) / float(t[1])
            q = float(t[0])
            if q <= 0:
                return None
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                return q / float(f.read().split()[0])
        except (OSError, ValueError, IndexError):
            pass
    return None


def library_sha256():
    """sha256 of the solver library this process runs (the counter summaries record the same)."""
    import hashlib
    from fbstab_amd import hip_api
    try:
        with open(hip_api.current_library_path(), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


def stored_traffic(batch: int):
    """HBM bytes per launch of the record kernel from a committed rocprofv3 --pmc summary for
    this batch size (FETCH_SIZE and WRITE_SIZE in separate passes; bench.py cannot run the
    profiler on itself, so the figure is REPLAYED from profiles/, not measured in this run).
    The summary taken on THIS build (library_sha256 recorded by tools/pmc_lib.sh) is preferred;
    failing that the newest one by name, marked as another build's.  Returns a dict: corrected
    bytes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies the 16-byte-per-lane reads of
    the records at half their bytes on gfx950, so the read side is doubled - an estimate the
    slot-by-slot ledger of DESIGN.md 4.1 supports; WRITE_SIZE is exact), the raw counter sum,
    the regime the counters were taken in, and the source file."""
    import glob
    sha = library_sha256()
    found = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("batch") == batch:
                same = sha is not None and t.get("library_sha256") == sha
                found.append((same, {
                    "corrected": t["hbm_bytes_per_launch_fetch_doubled"], "raw": t["hbm_bytes_per_launch_raw"],
                    "regime": t.get("regime", "one launch at a time (rocprofv3 --pmc over tools/variant_bench.py)"),
                    "build_matches": same,
                    "source": "profiles/" + os.path.basename(path) + " (replayed; build " +
                              str(t.get("build", "unrecorded")) + ("" if same else "; NOT the library of this run") + ")"}))
        except (OSError, ValueError, KeyError):
            pass
    for same, t in found:
        if same:
            return t
    return found[0][1] if found else None


# tools/probes/issue_rate_probe.hip (profiles/r05_g_issue_rate_probe.txt): ONE wavefront alone on its SIMD - the
# record kernel's regime, 493 registers - issues an FP64 FMA (fused with a DPP broadcast or not) every 5.25
# cycles, any other vector instruction every ~4.5.  The ceiling below prices EVERY vector instruction at the
# FP64 figure, as VERDICT r5 did.
ISSUE_CYCLES_PER_VALU = 5.25
SIMDS = 1024                   # 256 CUs x 4


def stored_issue_counters(batch: int):
    """SQ_INSTS_VALU per launch of the record kernel and the shader clock it ran at, REPLAYED from the newest
    committed sq-counter summary (profiles/r*_r16_sq_counters.json; the summary taken on THIS build is
    preferred, like stored_traffic).  Returns None when there is none for this batch size."""
    import glob
    sha = library_sha256()
    found = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_r16_sq_counters.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("batch", 8192) != batch or "SQ_INSTS_VALU" not in t:
                continue
            same = sha is not None and t.get("library_sha256") == sha
            found.append((same, {"valu": float(t["SQ_INSTS_VALU"]),
                                 "clock_mhz": float(t.get("mean_shader_clock_mhz", 2155.0)),
                                 "clock_source": "this summary" if "mean_shader_clock_mhz" in t else
                                                 "profiles/r05_zzz_r16_wave_time_shares.txt (2155 MHz, round 5)",
                                 "build_matches": same,
                                 "source": "profiles/" + os.path.basename(path) + " (replayed" +
                                           ("" if same else "; NOT the library of this run") + ")"}))
        except (OSError, ValueError, KeyError):
            pass
    for same, t in found:
        if same:
            return t
    return found[0][1] if found else None


def lane_buffers(torch, dev, batch, sizes, world, rank, gather):
    """The device buffers of ONE pipeline lane (bench.py's Lane; a function of its own so that the CPU
    suite can run exactly this plan under gloo at world size 8, tests/test_sharding.py): the record
    (z, l, v, y and the 5 doubles of SolverOut of a QP side by side - the gather moves ONE buffer), the
    views the solver writes through, the SolverOut bytes, and on rank 0 the receive list whose slot `rank`
    IS the record (the root's own block is never copied)."""
    from fbstab_amd import sharding
    nz, nl, nv = sizes
    nvar = nz + nl + 2 * nv
    rec = torch.zeros((batch, nvar + sharding.OUT_DOUBLES), dtype=torch.float64, device=dev)
    b = dict(rec=rec, x=rec[:, :nvar], z=rec[:, :nz], l=rec[:, nz:nz + nl], v=rec[:, nz + nl:nz + nl + nv],
             y=rec[:, nz + nl + nv:nvar], out=torch.zeros((batch, 40), dtype=torch.uint8, device=dev), grec=None)
    if gather and rank == 0:
        b["grec"] = [rec if g == rank else torch.empty_like(rec) for g in range(world)]
    return b


def memory_plan(world, lanes, batch, sizes, data_bytes_per_qp, scratch_bytes_per_handle, rank=0):
    """Bytes of HBM `rank` holds while the headline runs with `lanes` steps in flight (the plan main()
    executes; asserted to fit for --gpus 8 by tests/test_sharding.py and stated in DESIGN.md section 6):
    the resident problem data, per lane the record, the SolverOut bytes and the solver handle's scratch,
    and on rank 0 the other ranks' receive buffers of every lane."""
    nz, nl, nv = sizes
    rec = batch * (nz + nl + 2 * nv + 5) * 8
    per_lane = rec + batch * 40 + scratch_bytes_per_handle
    recv = (world - 1) * rec * lanes if rank == 0 and world > 1 else 0
    return dict(data=batch * data_bytes_per_qp, lanes=lanes * per_lane, receive=recv,
                total=batch * data_bytes_per_qp + lanes * per_lane + recv, record_bytes=rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8192, help="QPs per GPU")
    ap.add_argument("--pipeline", type=int, default=8,
                    help="steps in flight (alternating streams/handles); 1 = serial")
    ap.add_argument("--first-id", type=int, default=None,
                    help="developer option: first global instance id of this rank's shard "
                         "(default: rank * batch, the shard bench runs under torch.distributed.run)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="QPs for the CPU baseline (0 disables; default sized for ~15 s)")
    ap.add_argument("--extras", type=int, default=-1,
                    help="secondary blocks (serial, ltv_dense_rows, dense, receding): 1/0; default on at N = 1")
    args = ap.parse_args()

    import torch
    from tools import fixtures as fx
    from fbstab_amd import hip_api, sharding

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
    gpu_t0 = time.time()

    B = args.batch
    first_id, _ = sharding.shard_range(rank, world, B)
    if args.first_id is not None:
        first_id = args.first_id

    def upload(p):
        return {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}

    class Lane:
        """One pipeline lane: solver handle (own device scratch), stream, outputs."""

        def __init__(self, p, gather, lanes):
            # (every lane's handle is told how many lanes share the device: fbstab_hip_mpc_create_in_flight)
            self.solver = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=p.batch, device=local_rank, handles_in_flight=lanes)
            self.stream = torch.cuda.Stream(device=dev)
            # (rank 0's own block is already where it has to be: its slot of the receive list IS its
            # record - the gather then moves the other ranks' blocks only)
            b = lane_buffers(torch, dev, p.batch, (p.nz, p.nl, p.nv), world, rank, gather)
            self.rec, self.x, self.z, self.l, self.v, self.y = b["rec"], b["x"], b["z"], b["l"], b["v"], b["y"]
            self.out, self.grec = b["out"], b["grec"]
            self.events = []

    def run_mpc(p, data, P, steps, warmup, gather):
        """`steps` timed solves of the batch `p`, P in flight.  Returns a dict."""
        lanes = [Lane(p, gather, P) for _ in range(P)]

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
                if gather:
                    sharding.gather_solutions(ln.x, ln.out, dst=0, record=ln.rec, gather_list=ln.grec, stack=False)

        def fence():
            torch.cuda.synchronize()
            if gather:
                dist.barrier()
            torch.cuda.synchronize()

        # W untimed steps as asked - and no lane makes its FIRST solve (handle set-up, first touch of
        # its 1.8 GB of scratch) inside the timed region: with P lanes and W < P the lanes W .. P-1
        # get one untimed solve each as well (reported as `untimed_steps`)
        for k in range(max(warmup, min(P, steps))):
            step(k, False)
        fence()
        w0 = time.time()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k, True)
        fence()
        elapsed = time.perf_counter() - t0
        w1 = time.time()
        if gather:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        kernel_ms = [e0.elapsed_time(e1) for ln in lanes for (e0, e1) in ln.events]
        # every lane's last results (and, on rank 0, every rank's) are inspected
        outs = [hip_api.out_to_numpy(ln.out) for ln in lanes[:min(P, steps + warmup)]]
        if gather and rank == 0:
            for ln in lanes[:min(P, steps + warmup)]:
                for g in ln.grec:
                    outs.append(sharding.unpack_out(g))
        eflag = np.concatenate([o["eflag"] for o in outs])
        newton = np.concatenate([o["newton_iters"] for o in outs])
        res = dict(elapsed=elapsed, kernel_ms=float(np.mean(kernel_ms)), ok=bool((eflag == 0).all()),
                   not_converged=int((eflag != 0).sum()), mean_newton=float(newton.mean()),
                   query=lanes[0].solver.query(), kernel=lanes[0].solver.kernel_name(), wall=(w0, w1),
                   untimed=max(warmup, min(P, steps)))
        for ln in lanes:
            ln.solver.close()
        del lanes
        torch.cuda.empty_cache()
        return res

    P = max(1, args.pipeline)
    p = fx.synthetic_mpc_batch(B, first_id=first_id)  # shard by global instance id
    data = upload(p)
    head = run_mpc(p, data, P, args.steps, args.warmup, dist is not None)

    extras = args.extras if args.extras >= 0 else (1 if world == 1 else 0)
    blocks = {}
    if extras and rank == 0:
        ser = run_mpc(p, data, 1, min(args.steps, 6), 1, False)
        blocks["serial"] = {"value": B * min(args.steps, 6) / ser["elapsed"], "unit": "QPs/sec",
                            "ms_per_step": 1e3 * ser["elapsed"] / min(args.steps, 6), "kernel_ms": ser["kernel_ms"],
                            "steps_in_flight": 1, "all_converged": ser["ok"]}
        del data
        torch.cuda.empty_cache()
        pl = fx.synthetic_mpc_ltv_batch(B, first_id=first_id)
        dl = upload(pl)
        nl_steps = max(8, min(args.steps, 24))
        ltv = run_mpc(pl, dl, P, nl_steps, args.warmup, False)
        blocks["ltv_dense_rows"] = {
            "value": B * nl_steps / ltv["elapsed"], "unit": "QPs/sec", "ms_per_step": 1e3 * ltv["elapsed"] / nl_steps,
            "mean_newton_iters": ltv["mean_newton"], "steps": nl_steps, "steps_in_flight": P,
            "newton_iters_per_sec": ltv["mean_newton"] * B * nl_steps / ltv["elapsed"],
            "workload": "same shape and initial states; every stage has its own Q, R, S, A, B, E, L and every "
                        "constraint row 2-3 nonzeros (tools/fixtures.py: synthetic_mpc_ltv_batch)",
            "all_converged": ltv["ok"]}
        del dl, pl
        torch.cuda.empty_cache()
        blocks["dense"] = bench_dense(torch, dev, fx, hip_api)
        blocks["receding"] = bench_receding(torch, dev, fx, hip_api)
        blocks["latency"] = bench_latency(torch, dev, fx, hip_api)
        blocks["wide"] = bench_wide(torch, dev, fx, hip_api)
    if dist is not None and args.extras != 0 and (world > 1 or os.environ.get("FBSTAB_BENCH_SHARDED_SWEEP") == "1"):
        # N > 1: configs[4] sharded by trajectory (every rank takes part: one gather at
        # the end).  (The environment variable rehearses this branch with one rank.)
        data = None
        torch.cuda.empty_cache()
        rec_blk = bench_receding(torch, dev, fx, hip_api, dist=dist, rank=rank, world=world)
        if rank == 0:
            blocks["receding"] = rec_blk
    gpu_t1 = time.time()

    if rank == 0:
        elapsed = head["elapsed"]
        total_qps = world * B * args.steps / elapsed
        per_step_s = elapsed / args.steps
        achieved = ALG_BYTES_PER_QP * B * world / per_step_s / 1e9 / world  # per GPU
        k_ms = head["kernel_ms"]
        tr = stored_traffic(B)
        ic = stored_issue_counters(B)
        # the ceiling of the instruction stream: every SIMD issuing this launch's vector instructions back to
        # back at the probed rate of a lone wavefront, at the clock the kernel ran at
        issue_qps = (B / (ic["valu"] * ISSUE_CYCLES_PER_VALU / SIMDS / (ic["clock_mhz"] * 1e6))) if ic else None
        traffic = tr["corrected"] if tr else None
        rec = {
            "metric": "QPs/sec (batched MPC N=30 nx=12 nu=4 nc=20)",
            "value": total_qps, "unit": "QPs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * per_step_s,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: batched FBstabMpc, batch=8192 per GPU, "
                                   "N=30 nx=12 nu=4 nc=20, cold start, default options",
                       "batch_per_gpu": B, "global_batch": world * B, "steps_in_flight": P,
                       # warm-up steps actually run: max(W, lanes), so that every lane's first solve is untimed
                       "untimed_steps": head["untimed"],
                       "parallelism": f"batch sharded over {world} GPU(s)" +
                                      (", one RCCL gather to rank 0 per batch" if dist is not None else "")},
            # achieved = algorithmic bytes of one batch over the time the GPU spends per
            # batch (ms_per_step), per GPU; the per-launch figure (duration of one launch
            # as HIP events and rocprof see it, with `launches_in_flight` sharing the GPU)
            # is the named extra
            # `bound`: whichever of the two measured ceilings is the lower one - the memory wall of the record
            # traffic (hbm_ceiling_qps) or the issue rate of the kernel's own instruction stream
            # (issue_bound_qps): "hbm" or "issue".  achieved / peak / frac stay the ALGORITHMIC bytes over
            # the HBM peak whatever the bound is called (the contract's accounting).
            "roofline": {"bound": ("issue" if (issue_qps is not None and tr is not None and
                                               issue_qps < B / (tr["corrected"] / HBM_ACHIEVABLE_BPS)) else "hbm"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "issue_bound_qps": issue_qps,
                         "issue_bound": ({"valu_instructions_per_launch": ic["valu"],
                                          "cycles_per_instruction": ISSUE_CYCLES_PER_VALU,
                                          "cycles_source": "tools/probes/issue_rate_probe.hip (profiles/r05_g_issue_rate_probe.txt): "
                                                           "one wavefront per SIMD, FP64 FMA fused with a DPP broadcast or not",
                                          "simds": SIMDS, "shader_clock_mhz": ic["clock_mhz"],
                                          "clock_source": ic["clock_source"], "counter_source": ic["source"],
                                          "build_matches": ic["build_matches"],
                                          "frac_of_issue_bound": (world * B * args.steps / elapsed / world) / issue_qps}
                                         if ic else None),
                         # corrected counter bytes per launch (2 x FETCH_SIZE + WRITE_SIZE), the raw
                         # sum beside it, their ratio to the algorithmic bytes and the regime the
                         # counters were taken in (the profiler serialises dispatches)
                         "traffic": traffic, "traffic_raw": tr["raw"] if tr else None,
                         "traffic_ratio": (traffic / (ALG_BYTES_PER_QP * B)) if traffic else None,
                         "traffic_regime": tr["regime"] if tr else None,
                         "traffic_source": tr["source"] if tr else None,
                         "traffic_build_matches": tr["build_matches"] if tr else None,
                         "traffic_per_step_GBps": (traffic / per_step_s / 1e9) if traffic else None,
                         # the rate at which HBM (6.29 TB/s achievable on this part, MI355X_MICROARCH.md) could
                         # move this kernel's counter bytes: the memory wall of the present record traffic
                         "hbm_ceiling_qps": (B / (traffic / HBM_ACHIEVABLE_BPS)) if traffic else None,
                         "kernel": head["kernel"], "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_QP * B,
                         "per_launch": {"achieved": ALG_BYTES_PER_QP * B / (k_ms * 1e-3) / 1e9,
                                        "frac": ALG_BYTES_PER_QP * B / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "launches_in_flight": P}},
            "fp64": {"model_flop_per_newton_iter": FLOP_PER_NEWTON,
                     "mean_newton_iters": head["mean_newton"],
                     # (rate of one GPU: flops of one batch over the time per step)
                     "achieved_tflops": FLOP_PER_NEWTON * head["mean_newton"] * B / per_step_s / 1e12,
                     "peak_tflops": FP64_PEAK_TFLOPS},
            "all_converged": head["ok"], "not_converged": head["not_converged"],
            "launch": head["query"],
            "gpu_leg": {"timed_region_unix": [head["wall"][0], head["wall"][1]],
                        "all_gpu_work_unix": [gpu_t0, gpu_t1]},
        }
        rec.update(blocks)
        n_cpu = args.cpu_sample
        if n_cpu < 0:
            n_cpu = 256 * max(1, (os.cpu_count() or 1))
        if n_cpu > 0 and world == 1:
            rec["cpu_baseline"] = cpu_baseline(n_cpu)
            if "latency" in rec:
                cpu_ms = 1e3 / rec["cpu_baseline"]["single_thread_value"]
                lat = rec["latency"]
                lat["cpu_single_thread_ms_per_qp_mean_over_sample"] = cpu_ms
        elif world > 1:
            rec["cpu_baseline"] = None
        print(json.dumps(rec))
    if dist is not None:
        dist.destroy_process_group()


def bench_dense(torch, dev, fx, hip_api, batch=4096, steps=24, lanes=8, order=None):
    """BASELINE configs[1]: batched FBstabDense, batch 4096, nz=50 nl=10 nv=100.  order: None =
    the handle's default elimination order of the KKT factorisation (Eigen's, the reference's:
    fbstab_hip_dense_set_factorisation), or "auto" / "natural" (the opt-in faster orders, whose
    iteration counts can differ from the reference's on degenerate QPs - not on this workload)."""
    if order is None:
        r = bench_dense(torch, dev, fx, hip_api, batch, steps, lanes, order="pivoted")
        r["factorisation_order"] = "pivoted (default: Eigen's rule, dense_cholesky_solver.cc:70-79)"
        r["opt_in_orders"] = {}
        for o in ("auto", "natural"):
            f = bench_dense(torch, dev, fx, hip_api, batch, steps, lanes, order=o)
            r["opt_in_orders"][o] = {k: f[k] for k in ("value", "ms_per_step", "kernel_ms", "serial_value",
                                                       "mean_newton_iters", "all_converged", "pivoted_steps_per_launch")}
            r["opt_in_orders"][o]["roofline_frac"] = f["roofline"]["frac"]
        return r
    p = fx.synthetic_dense_batch(batch, 50, 10, 100)
    data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
    mk = lambda n: torch.zeros((batch, n), dtype=torch.float64, device=dev)
    L = []
    for _ in range(lanes):
        L.append(dict(s=hip_api.FBstabDenseBatch(50, 10, 100, max_batch=batch), st=torch.cuda.Stream(device=dev),
                      z=mk(50), l=mk(10), v=mk(100), y=mk(100),
                      out=torch.zeros((batch, 40), dtype=torch.uint8, device=dev)))
        sv = L[-1]["s"]
        sv.SetFactorisation({"pivoted": sv.ORDER_PIVOTED, "auto": sv.ORDER_AUTO, "natural": sv.ORDER_NATURAL}[order])
    # one launch alone: kernel time
    ln = L[0]
    for _ in range(2):
        for a in (ln["z"], ln["l"], ln["v"]):
            a.zero_()
        ln["s"].Solve(data, ln["z"], ln["l"], ln["v"], ln["y"], out=ln["out"])
    k_ms = ln["s"].last_kernel_ms()
    handed = ln["s"].Factorisation()["pivoted_steps"]

    def step(k):
        ln = L[k % lanes]
        with torch.cuda.stream(ln["st"]):
            for a in (ln["z"], ln["l"], ln["v"]):
                a.zero_()
            ln["s"].Solve(data, ln["z"], ln["l"], ln["v"], ln["y"], out=ln["out"],
                          stream=ln["st"].cuda_stream, async_=True)
    for k in range(lanes):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    outs = [hip_api.out_to_numpy(ln["out"]) for ln in L]
    ok = all((o["eflag"] == 0).all() for o in outs)
    per_step = dt / steps
    ach = DENSE_ALG_BYTES_PER_QP * batch / per_step / 1e9
    # HBM bytes per launch from the newest committed counter summary of the dense kernel
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; replayed, not measured in this run)
    traffic, traffic_raw, traffic_src = None, None, None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_dense_wave_counters.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("order", "natural") != order:  # (summaries older than round 4: the natural order)
                continue
            traffic_raw = t["derived"]["hbm_bytes_per_launch_raw"]
            # corrected: the read side doubled (16-byte-per-lane reads, MI355X_MICROARCH.md HBM section)
            traffic = (2.0 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024.0
            traffic_src = ("profiles/" + os.path.basename(path) + " (replayed; one launch at a time under rocprofv3 --pmc; " +
                           ("this build" if t.get("library_sha256") == library_sha256() else "NOT the library of this run") + ")")
            break
        except (OSError, ValueError, KeyError):
            pass
    r = {"config": "BASELINE configs[1]: batched FBstabDense, batch=4096, nz=50 nl=10 nv=100, cold start",
         "value": batch * steps / dt, "unit": "QPs/sec", "ms_per_step": 1e3 * per_step, "steps": steps,
         "steps_in_flight": lanes, "kernel_ms": k_ms, "serial_value": batch / (k_ms * 1e-3),
         "mean_newton_iters": float(outs[0]["newton_iters"].mean()), "all_converged": ok,
         "pivoted_steps_per_launch": handed,
         "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_raw": traffic_raw,
                      "traffic_ratio": (traffic / (DENSE_ALG_BYTES_PER_QP * batch)) if traffic else None,
                      "traffic_source": traffic_src,
                      "kernel": "fbstab_dense_wave_kernel" if L[0]["s"].query()["threads"] == 64
                      else "fbstab_dense_kernel",
                      "algorithmic_bytes_per_launch": DENSE_ALG_BYTES_PER_QP * batch},
         "launch": L[0]["s"].query()}
    for ln in L:
        ln["s"].close()
    return r


def bench_latency(torch, dev, fx, hip_api, batches=(1, 16, 256, 2048), repeats=7):
    """What ONE caller sees: wall ms of a cold-started solve (zero guess) of a small batch, call
    to completion, one call at a time.  `device_pointers`: fbstab_hip_mpc_solve_batch on
    resident arrays (synchronous call).  `facade_host_pointers`: FBstabMpc::Solve of the C++
    facade on host arrays, Display::OFF (tools/cpp/facade_latency.cc, a child process; the
    staging copies of one QP's 189 KB are inside the time).  A batch of one occupies one
    16-lane row of one wavefront: its time is the dependent chain of its ~19 Newton steps."""
    import subprocess
    import tempfile
    res = {"unit": "ms per solve call (cold start, default options)", "repeats": repeats, "device_pointers": {}}
    for b in batches:
        p = fx.synthetic_mpc_batch(b)
        s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=b, device=dev.index or 0)
        data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
        mk = lambda n: torch.zeros((b, n), dtype=torch.float64, device=dev)
        z, l, v, y = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
        ms, kms = [], []
        for k in range(repeats + 1):
            for a in (z, l, v):
                a.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = s.Solve(data, z, l, v, y)     # synchronous
            torch.cuda.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0))
            kms.append(s.last_kernel_ms())
        o = hip_api.out_to_numpy(out)
        res["device_pointers"][str(b)] = {
            "ms_median": float(np.median(ms[1:])), "ms_min": float(np.min(ms[1:])),
            "kernel_ms_median": float(np.median(kms[1:])), "qps_per_sec": b / (1e-3 * float(np.median(ms[1:]))),
            "max_newton_iters": int(o["newton_iters"].max()), "all_converged": bool((o["eflag"] == 0).all())}
        s.close()
    # the CPU restatement on the SAME QP (instance 0, the batch-of-one above), one thread
    try:
        from oracle.oracle_py import Oracle
        orc = Oracle(False)
        p1 = fx.synthetic_mpc_batch(1)
        orc.solve_mpc(p1, nthreads=1)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            oo = orc.solve_mpc(p1, nthreads=1)[4]
            ts.append(1e3 * (time.perf_counter() - t0))
        res["cpu_same_qp_single_thread"] = {"ms_median": float(np.median(ts)), "ms_min": float(np.min(ts)),
                                            "newton_iters": int(oo["newton_iters"][0]), "kind": "port"}
        res["gpu_batch1_over_cpu_same_qp"] = res["device_pointers"]["1"]["ms_median"] / float(np.median(ts))
    except Exception as e:  # (the oracle is a checker: its absence does not stop the bench)
        res["cpu_same_qp_single_thread"] = {"error": str(e)}
    exe = os.path.join(ROOT, "tools", "cpp", "facade_latency")
    if os.path.exists(exe):
        p = fx.synthetic_mpc_batch(1)
        with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
            np.asarray(p.sizes(), dtype=np.int32).tofile(f)
            for k in hip_api.MPC_SEQ:
                np.ascontiguousarray(p.arrays[k][0], dtype=np.float64).tofile(f)
            path = f.name
        try:
            r = subprocess.run([exe, path, str(repeats)], capture_output=True, text=True, timeout=120)
            res["facade_host_pointers"] = json.loads(r.stdout) if r.returncode == 0 else {"error": r.stderr[-300:]}
        except (OSError, ValueError, subprocess.TimeoutExpired) as e:
            res["facade_host_pointers"] = {"error": str(e)}
        finally:
            os.unlink(path)
    else:
        res["facade_host_pointers"] = {"error": "tools/cpp/facade_latency not built (__graft_entry__.build())"}
    return res


def bench_receding(torch, dev, fx, hip_api, trajectories=4096, steps=200, dist=None, rank=0, world=1):
    """BASELINE configs[4]: warm-started receding-horizon sweep, plant step
    x+ = A x + B u0, retirement of failed trajectories and the warm start all on
    the device, one C call and ONE launch for the whole sweep
    (fbstab_hip_mpc_receding_sweep): every 16-lane row advances its own trajectory
    through all the steps, so no trajectory waits for the slowest solve of a step.
    Under torch.distributed every rank sweeps its own block of `trajectories`
    (global ids rank * trajectories ...; nothing is exchanged while the sweep runs) and
    the applied inputs go to rank 0 in ONE gather at the end (sharding.gather_input_log);
    the time is the slowest rank's, barrier to barrier, gather included."""
    p = fx.synthetic_mpc_batch(trajectories, first_id=rank * trajectories)
    N, nx, nu, nc = p.sizes()
    A, Bm = fx.quadrotor_model()
    s = hip_api.FBstabMpcBatch(N, nx, nu, nc, max_batch=trajectories, device=dev.index or 0)
    data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
    mk = lambda n: torch.zeros((trajectories, n), dtype=torch.float64, device=dev)
    z, l, v, y = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
    glist = None
    if dist is not None and rank == 0:
        glist = [torch.empty((steps, trajectories, nu), dtype=torch.float64, device=dev) for _ in range(world)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    r = s.RecedingSweep(data, z, l, v, y, A, Bm, steps, retire=True, log_inputs=dist is not None)
    if dist is not None:
        from fbstab_amd import sharding
        sharding.gather_input_log(r["u"], dst=0, gather_list=glist)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    st = r["stats"]
    res = {"config": f"BASELINE configs[4]: {world} x 4096 closed-loop trajectories x 200 steps, N=30 nx=12 nu=4 nc=20, "
                     "warm start (unshifted), x+ = A x + B u0 on the device, failed trajectories retired to the origin"
                     + ("; one gather of the applied inputs to rank 0 at the end" if dist is not None else ""),
           "value": world * trajectories * steps / dt, "unit": "QPs/sec", "wall_ms_per_step": 1e3 * dt / steps,
           "n_gpus": world, "scaling": "weak",
           # one launch: the device time of the launch divided by the steps, the same
           # figure for every step (the launch lasts as long as its slowest TRAJECTORY;
           # with a launch per step - FBSTAB_HIP_SWEEP_PER_STEP=1, the round-2 start -
           # every step waited for its slowest QP: 790 ms against 350 ms)
           "launches": 1, "kernel_ms_total": float(r["kernel_ms"].sum()),
           "kernel_ms_first": float(r["kernel_ms"][0]), "kernel_ms_median": float(np.median(r["kernel_ms"])),
           "kernel_ms_mean": float(r["kernel_ms"].mean()), "kernel_ms_max_after_first": float(r["kernel_ms"][1:].max()),
           "kernel_ms_last": float(r["kernel_ms"][-1]), "wall_over_kernel_sum": dt * 1e3 / float(r["kernel_ms"].sum()),
           "trajectories": trajectories, "steps": steps,
           "retired": int(st["retired_total"][-1]),
           "mean_newton_first_step": float(st["newton_sum"][0]) / trajectories,
           "mean_newton_last_step": float(st["newton_sum"][-1]) / trajectories,
           "host_syncs_per_step": 0}
    s.close()
    return res


def mpc_alg_bytes(N, nx, nu, nc):
    """Algorithmic bytes of one MPC QP (SURVEY 8d's accounting for any shape): the problem data, the guess
    (z, l, v), the solution (z, l, v, y) and the 40-byte SolverOut."""
    data = 8 * ((N + 1) * (nx * nx + nu * nu + nu * nx + nx + nu + nc * nx + nc * nu + nc) +
                N * (nx * nx + nx * nu + nx) + nx)
    nz, nl, nv = (N + 1) * (nx + nu), (N + 1) * nx, (N + 1) * nc
    return data + 8 * (nz + nl + nv) + 8 * (nz + nl + 2 * nv) + 40


def bench_wide(torch, dev, fx, hip_api, reps=3, in_flight=8, stream_steps=48):
    """The row-pair record instances (stage widths 17..32), one launch at a time, device pointers
    (the kernel furthest below its roofline had no driver number):
      ltv_30_20_6_16   2048 random time-varying QPs of (N, nx, nu, nc) = (30, 20, 6, 16) - 64 distinct problems
                       tiled over the batch, tools/shape_bench.py's workload - on fbstab_mpc_r32_kernel<24,8,16>;
      reactor_N80      1024 perturbed CopolymerizationReactor problems, N = 80, nx = 18, nu = 5, nc = 10
                       (fbstab/test/ocp_generator.cc:73-174; tools/reactor_bench.py) on <18,5,10>.
    `roofline`: the shape's own algorithmic bytes over the launch's duration against the HBM peak.
    `value` is ONE launch at a time; `in_flight` is the same batch as a stream, `in_flight` launches on as many
    streams (the headline's regime), `stream_steps` timed launches after one untimed launch per lane;
    `in_flight.handles_in_flight` is what each lane's handle was created with (1: it keeps the whole grid)."""
    out = {}
    def run(name, p, what, share):
        N, nx, nu, nc = p.sizes()
        B = p.batch
        s = hip_api.FBstabMpcBatch(N, nx, nu, nc, max_batch=B)
        data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
        mk = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
        ms = []
        for rep in range(reps + 1):   # (the first launch is the handle's warm-up)
            z, l, v, y = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
            o = hip_api.out_to_numpy(s.Solve(data, z, l, v, y))
            if rep:
                ms.append(s.last_kernel_ms())
        k_ms = float(np.median(ms))
        ab = mpc_alg_bytes(N, nx, nu, nc)
        ach = ab * B / (k_ms * 1e-3) / 1e9
        out[name] = {"workload": what, "shape": [N, nx, nu, nc], "batch": B, "kernel": s.kernel_name(),
                     "value": B / (k_ms * 1e-3), "unit": "QPs/sec", "kernel_ms": k_ms, "launches_timed": reps,
                     "mean_newton_iters": float(o["newton_iters"].mean()),
                     "us_per_newton_step_and_qp": 1e3 * k_ms / (B * float(o["newton_iters"].mean())),
                     "all_converged": bool((o["eflag"] == 0).all()), "not_converged": int((o["eflag"] != 0).sum()),
                     "launch": s.query(),
                     "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_qp": ab}}
        s.close()
        # the same batch as a STREAM: `lanes` launches in flight on as many streams - the headline's regime
        # (tools/wide_in_flight.py; a stream of batches fills the tail one launch leaves: LABNOTES R6.11)
        lanes = []
        for _ in range(in_flight):
            h = hip_api.FBstabMpcBatch(N, nx, nu, nc, max_batch=B, handles_in_flight=share)
            lanes.append(dict(s=h, st=torch.cuda.Stream(device=dev), z=mk(p.nz), l=mk(p.nl), v=mk(p.nv), y=mk(p.nv),
                              out=torch.zeros((B, 40), dtype=torch.uint8, device=dev)))
        def step(k):
            ln = lanes[k % in_flight]
            with torch.cuda.stream(ln["st"]):
                for a in (ln["z"], ln["l"], ln["v"]):
                    a.zero_()
                ln["s"].Solve(data, ln["z"], ln["l"], ln["v"], ln["y"], out=ln["out"], stream=ln["st"].cuda_stream, async_=True)
        for k in range(in_flight):
            step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(stream_steps):
            step(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        oo = [hip_api.out_to_numpy(ln["out"]) for ln in lanes]
        out[name]["in_flight"] = {"value": B * stream_steps / dt, "unit": "QPs/sec", "steps": stream_steps,
                                  "steps_in_flight": in_flight, "handles_in_flight": share, "ms_per_step": 1e3 * dt / stream_steps,
                                  "workgroups_per_launch": lanes[0]["s"].query()["workgroups"],
                                  "mean_newton_iters": float(np.mean([x["newton_iters"].mean() for x in oo])),
                                  "all_converged": bool(all((x["eflag"] == 0).all() for x in oo))}
        for ln in lanes:
            ln["s"].close()
        del lanes
        del data
        torch.cuda.empty_cache()
    one = fx.random_ltv_mpc(np.random.default_rng(5), 64, 30, 20, 6, 16)
    p = fx.MpcProblem(30, 20, 6, 16)
    p.arrays = {k: np.ascontiguousarray(np.tile(a, (32, 1))) for k, a in one.arrays.items()}
    # (the grid each lane's handle takes - fbstab_hip_mpc_create_in_flight's `handles_in_flight` - is the better of 1 and
    # `in_flight` per workload, LABNOTES R6.11: whole-grid launches for <24,8,16>, an eighth of the grid for <18,5,10>)
    run("ltv_30_20_6_16", p, "2048 random time-varying QPs (64 distinct, tiled), N=30 nx=20 nu=6 nc=16, dense constraint rows, cold start", 1)
    gen = fx.OcpGenerator()
    gen.CopolymerizationReactor(80)
    one = gen.GetFBstabInput()
    N, nx, nu, nc = one.sizes()
    B = 1024
    rng = np.random.default_rng(3)
    p = fx.MpcProblem(N, nx, nu, nc)
    p.arrays = {k: np.ascontiguousarray(np.broadcast_to(a, (B, a.shape[1]))).copy() for k, a in one.arrays.items()}
    p.arrays["x0"] = p.arrays["x0"] * (1.0 + 0.2 * rng.standard_normal((B, nx)))
    run("reactor_N80", p, "1024 CopolymerizationReactor problems (the reference's, N=80 nx=18 nu=5 nc=10), x0 perturbed by 20 %, cold start", in_flight)
    return out


if __name__ == "__main__":
    main()
