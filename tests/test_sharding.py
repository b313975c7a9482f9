"""Multi-process test of the N>1 path on CPU (gloo, world_size 2): each rank
owns a contiguous block of global instance ids, solves its shard (here with the
oracle, since there is no GPU), and ONE gather assembles the batch on rank 0 in
global order - bit-identical to solving the whole batch in one process."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, per_rank, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from tools import fixtures as fx
    from fbstab_amd import sharding
    from oracle.oracle_py import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, last = sharding.shard_range(rank, world, per_rank)
    p = fx.synthetic_mpc_batch(last - first, first_id=first)
    z, l, v, y, out = Oracle(False).solve_mpc(p)
    x = torch.from_numpy(np.concatenate([z, l, v, y], axis=1))
    o = torch.from_numpy(np.frombuffer(out.tobytes(), dtype=np.uint8).reshape(len(out), 40).copy())
    calls = []
    real_gather = dist.gather
    dist.gather = lambda *a, **k: (calls.append(1), real_gather(*a, **k))[1]
    X, O = sharding.gather_solutions(x, o, dst=0)
    # the way bench.py calls it: the solver writes into the first columns of a
    # preallocated record, receive buffers are preallocated on rank 0
    rec = torch.zeros((x.shape[0], x.shape[1] + sharding.OUT_DOUBLES), dtype=torch.float64)
    rec[:, :x.shape[1]] = x
    glist = [torch.empty_like(rec) for _ in range(world)] if rank == 0 else None
    X2, O2 = sharding.gather_solutions(rec[:, :x.shape[1]], o, dst=0, record=rec, gather_list=glist)
    # ... and without the stacked copy on rank 0 (bench.py: the receive buffers are kept)
    G3, none = sharding.gather_solutions(rec[:, :x.shape[1]], o, dst=0, record=rec, gather_list=glist, stack=False)
    assert none is None and (G3 is glist if rank == 0 else G3 is None)
    # ... and with rank 0's slot of the receive list being its record itself (bench.py: the root's own
    # block is not copied)
    alias = [rec if g == 0 else torch.empty_like(rec) for g in range(world)] if rank == 0 else None
    G4, _ = sharding.gather_solutions(rec[:, :x.shape[1]], o, dst=0, record=rec, gather_list=alias, stack=False)
    dist.gather = real_gather
    assert len(calls) == 4, "one collective per batch"
    if rank == 0:
        assert torch.equal(X, X2) and torch.equal(O, O2)
        assert G4[0] is rec and all(torch.equal(a, b) for a, b in zip(G4, glist))
        unpacked = np.concatenate([sharding.unpack_out(g) for g in glist])
        assert np.array_equal(unpacked["newton_iters"], np.frombuffer(O.numpy().tobytes(), dtype=out.dtype)["newton_iters"])
        q.put((X.numpy(), O.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_equals_single_process(oracle):
    import torch.multiprocessing as mp
    from tools import fixtures as fx
    from fbstab_amd import sharding
    world, per_rank = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    X, O = q.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p = fx.synthetic_mpc_batch(world * per_rank)
    z, l, v, y, out = oracle.solve_mpc(p)
    assert np.array_equal(X, np.concatenate([z, l, v, y], axis=1))
    got = np.frombuffer(O.tobytes(), dtype=out.dtype)
    for f in ("eflag", "newton_iters", "prox_iters", "residual"):
        assert np.array_equal(got[f], out[f])
    assert sharding.shard_range(1, 2, 8192) == (8192, 16384)
    with pytest.raises(ValueError):
        sharding.shard_range(2, 2, 1)


def _oracle_sweep(p, steps):
    """Closed loop of tools/fixtures' plant with the oracle as the solver (warm start
    unshifted, as the device sweep): the applied inputs, (steps, B, nu)."""
    from tools import fixtures as fx
    from tests.closed_loop import closed_loop
    from oracle.oracle_py import Oracle
    orc = Oracle(False)
    A, Bm = fx.quadrotor_model()
    B = p.batch

    def solve(x0, z, l, v):
        p.arrays["x0"] = np.ascontiguousarray(x0)
        return orc.solve_mpc(p, x0guess=(z, l, v))

    log = closed_loop(solve, p.arrays["x0"].copy(), np.zeros((B, p.nz)), np.zeros((B, p.nl)), np.zeros((B, p.nv)),
                      A, Bm, p.nx, p.nu, steps)
    return np.stack([r["u0"] for r in log])


def _sweep_worker(rank, world, port, per_rank, steps, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from tools import fixtures as fx
    from fbstab_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, last = sharding.shard_range(rank, world, per_rank)
    u = torch.from_numpy(_oracle_sweep(fx.synthetic_mpc_batch(last - first, first_id=first, N=8), steps))
    calls = []
    real_gather = dist.gather
    dist.gather = lambda *a, **k: (calls.append(1), real_gather(*a, **k))[1]
    logs = sharding.gather_input_log(u, dst=0)
    dist.gather = real_gather
    assert len(calls) == 1, "one collective per sweep"
    if rank == 0:
        q.put(torch.cat(logs, dim=1).numpy())
    else:
        assert logs is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_receding_sweep_equals_single_process():
    """BASELINE configs[4] over W ranks: trajectories are sharded by global id, every
    rank sweeps its block with no exchange, ONE gather of the input log at the end -
    bit-identical to sweeping all trajectories in one process."""
    import torch.multiprocessing as mp
    from tools import fixtures as fx
    world, per_rank, steps = 2, 3, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sweep_worker, args=(r, world, port, per_rank, steps, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    U = q.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    want = _oracle_sweep(fx.synthetic_mpc_batch(world * per_rank, N=8), steps)
    assert U.shape == (steps, world * per_rank, 4)
    assert np.array_equal(U, want)
    assert np.abs(want).max() > 0.1


# ---- world size 8: bench.py's own rank-0 buffer plan under gloo -------------
def _synthetic_results(first, count, sizes):
    """Deterministic stand-ins for a shard's results, a function of the GLOBAL instance id alone (no
    solver here: eight oracle processes on eight CPUs would only slow the suite down - what is under
    test is the exchange): every entry of x is id + column / 4096, the SolverOut record carries the id."""
    from fbstab_amd.hip_api import OUT_DTYPE
    nz, nl, nv = sizes
    nvar = nz + nl + 2 * nv
    ids = np.arange(first, first + count)
    x = ids[:, None].astype(np.float64) + np.arange(nvar)[None, :] / 4096.0
    out = np.zeros(count, dtype=OUT_DTYPE)
    out["eflag"] = ids % 3
    out["newton_iters"] = 7 + ids
    out["prox_iters"] = 1 + ids % 5
    out["residual"] = 1e-7 * (1 + ids)
    return x, out


def _world8_worker(rank, world, port, per_rank, lanes, steps, nsteps_sweep, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    from fbstab_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = (12, 5, 9)
    nvar = sizes[0] + sizes[1] + 2 * sizes[2]
    first, last = sharding.shard_range(rank, world, per_rank)
    dev = torch.device("cpu")
    # exactly what bench.py's Lane allocates, `lanes` of them: on rank 0 every lane's receive list with
    # the rank's own record as its own slot
    bufs = [bench.lane_buffers(torch, dev, per_rank, sizes, world, rank, True) for _ in range(lanes)]
    calls = []
    real_gather = dist.gather
    dist.gather = lambda *a, **k: (calls.append(1), real_gather(*a, **k))[1]
    for k in range(steps):   # the way run_mpc() steps through its lanes
        b = bufs[k % lanes]
        x, out = _synthetic_results(first + 1000 * k, per_rank, sizes)
        b["x"].copy_(torch.from_numpy(x))                    # "the solver writes through the views of the record"
        b["out"].copy_(torch.from_numpy(np.frombuffer(out.tobytes(), dtype=np.uint8).reshape(per_rank, 40).copy()))
        got, none = sharding.gather_solutions(b["x"], b["out"], dst=0, record=b["rec"], gather_list=b["grec"], stack=False)
        assert none is None
        if rank == 0:
            assert got is b["grec"] and got[0] is b["rec"]   # nothing stacked, the root's block not copied
            for g in range(world):
                xg, og = _synthetic_results(g * per_rank + 1000 * k, per_rank, sizes)
                assert torch.equal(got[g][:, :nvar], torch.from_numpy(xg)), (k, g)
                u = sharding.unpack_out(got[g])
                for f in ("eflag", "newton_iters", "prox_iters", "residual"):
                    assert np.array_equal(u[f], og[f]), (k, g, f)
        else:
            assert got is None
    # configs[4]: one gather of the input log when the sweep is over, preallocated receive list
    u_log = torch.from_numpy(np.arange(first, last)[None, :, None] + 0.001 * np.arange(nsteps_sweep)[:, None, None]
                             + np.zeros((1, 1, 4)))
    glist = [torch.empty_like(u_log) for _ in range(world)] if rank == 0 else None
    logs = sharding.gather_input_log(u_log, dst=0, gather_list=glist)
    dist.gather = real_gather
    assert len(calls) == steps + 1, "one collective per batch, one per sweep"
    if rank == 0:
        assert logs is glist
        full = torch.cat(logs, dim=1).numpy()
        assert np.array_equal(full[:, :, 0], np.arange(world * per_rank)[None, :] + 0.001 * np.arange(nsteps_sweep)[:, None])
        q.put(("ok", sum(t.numel() * 8 for b in bufs for t in b["grec"][1:])))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_8_gather_with_the_bench_lane_plan():
    """`bench.py --gpus 8` as far as it can be rehearsed without GPUs: eight gloo ranks, each with
    bench.lane_buffers() for several lanes in flight, stepping through them the way run_mpc() does -
    gather_solutions(record=, gather_list=, stack=False) with rank 0's slot aliased to its own record -
    and one gather_input_log() with a preallocated list at the end.  Every block arrives in global
    instance order in every lane, one collective per batch; the bytes rank 0 receives are the plan's."""
    import torch.multiprocessing as mp
    import bench
    world, per_rank, lanes, steps, sweep_steps = 8, 6, 3, 7, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, per_rank, lanes, steps, sweep_steps, q))
             for r in range(world)]
    for pr in procs:
        pr.start()
    status, recv_bytes = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert status == "ok"
    plan = bench.memory_plan(world, lanes, per_rank, (12, 5, 9), 0, 0)
    assert recv_bytes == plan["receive"]


def test_rank0_memory_plan_of_the_8_gpu_headline_fits():
    """The buffers `bench.py --gpus 8` holds on rank 0 at its defaults (batch 8192 per GPU, eight steps in
    flight): 8 lanes x 7 receive buffers of 138.5 MB beside the lanes' own records, eight solver handles'
    scratch (the size fbstab_hip_mpc_create_in_flight gives a handle that shares the device with seven
    others - taken from the record layout's constants, include/fbstab_hip.h documents the rule) and the
    resident problem data: ~11 GB of the 288 GB of an MI355X."""
    import bench
    N, nx, nu, nc = 30, 12, 4, 20
    sizes = ((N + 1) * (nx + nu), (N + 1) * nx, (N + 1) * nc)
    data_bytes = bench.ALG_BYTES_PER_QP - 11904 - 16864 - 40   # the problem data's share
    assert data_bytes == 188928
    scratch = 437 * 2 ** 20   # measured: fbstab_hip_mpc_query of a handle created with handles_in_flight = 8
    plan = bench.memory_plan(8, 8, 8192, sizes, data_bytes, scratch)
    assert plan["record_bytes"] == 8192 * (2108 + 5) * 8
    assert 7.5e9 < plan["receive"] < 8.0e9             # 8 lanes x 7 ranks x 138.5 MB
    assert plan["total"] < 0.05 * 288e9, plan           # ~13 GB: a twentieth of the HBM
    other = bench.memory_plan(8, 8, 8192, sizes, data_bytes, scratch, rank=3)
    assert other["receive"] == 0 and other["total"] < plan["total"]


# ---- the C-ABI's own multi-GPU entry (include/fbstab_hip.h: fbstab_hip_*_sharded) --------
def test_shard_group_argument_checks_without_gpu():
    """No GPU here: the group cannot be created (no CPU path), bad arguments are rejected
    before anything touches a device."""
    import ctypes as C
    from fbstab_amd import hip_api
    lib = hip_api.load_library()
    g = C.c_void_p()
    one = (C.c_int * 1)(0)
    assert lib.fbstab_hip_shard_group_create(0, one, C.byref(g)) == 1 and not g.value
    assert lib.fbstab_hip_shard_group_create(1, None, C.byref(g)) == 1
    if lib.fbstab_hip_device_count() == 0:
        assert lib.fbstab_hip_shard_group_create(1, one, C.byref(g)) == 2
        assert b"no HIP device" in lib.fbstab_hip_last_error()
    assert lib.fbstab_hip_shard_group_destroy(None) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("self_send", ["0", "1"])
def test_sharded_entries_on_one_device_equal_the_plain_calls(monkeypatch, self_send):
    """fbstab_hip_mpc_solve_batch_sharded / _dense_ / _receding_sweep_sharded with a
    group of ONE device (all the hardware a test box has): solutions, SolverOut records
    and the input log on the root equal those of the plain calls bit for bit, one
    collective per call.  With FBSTAB_HIP_SHARD_SELF_SEND=1 the root's own shard
    travels through grouped ncclSend / ncclRecv as a peer's would (RCCL loaded with
    dlopen, a one-rank communicator): the rehearsal of the xGMI path."""
    import torch
    from fbstab_amd import hip_api
    from tools import fixtures as fx
    monkeypatch.setenv("FBSTAB_HIP_SHARD_SELF_SEND", self_send)
    dev = torch.device("cuda:0")
    up = lambda p: {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
    g = hip_api.ShardGroup([0])
    B = 64
    for kind, p in (("mpc", fx.synthetic_mpc_batch(B, first_id=900)), ("dense", fx.synthetic_dense_batch(B, 50, 10, 100))):
        mk = (lambda: hip_api.FBstabMpcBatch(*p.sizes(), max_batch=B)) if kind == "mpc" else \
            (lambda: hip_api.FBstabDenseBatch(p.nz, p.nl, p.nv, max_batch=B))
        s = mk()
        data = up(p)
        z = lambda n: torch.zeros((B, n), dtype=torch.float64, device=dev)
        ref = (z(p.nz), z(p.nl), z(p.nv), z(p.nv))
        out_ref = hip_api.out_to_numpy(s.Solve(data, *ref))
        # packed arrays: z, l, v, y and the records travel as five pieces
        x = (z(p.nz), z(p.nl), z(p.nv), z(p.nv))
        root = (z(p.nz) + 7, z(p.nl) + 7, z(p.nv) + 7, z(p.nv) + 7)
        out = torch.zeros((B, 40), dtype=torch.uint8, device=dev)
        root_out = torch.zeros((B, 40), dtype=torch.uint8, device=dev)
        before = g.stats()
        g.Solve([s], [data], [x], [out], 0, root, root_out)
        after = g.stats()
        assert after["gathers"] == before["gathers"] + 1
        assert after["rccl_ops"] - before["rccl_ops"] == (5 if self_send == "1" else 0)
        for a, b in zip(root, ref):
            assert torch.equal(a, b)
        o = hip_api.out_to_numpy(root_out)
        for f in ("eflag", "residual", "newton_iters", "prox_iters", "initial_residual"):
            assert np.array_equal(o[f], out_ref[f]), f
        # one record per QP (the bench's layout): ONE piece for the solution
        nvar = p.nz + p.nl + 2 * p.nv
        cut = lambda r: (r[:, :p.nz], r[:, p.nz:p.nz + p.nl], r[:, p.nz + p.nl:p.nz + p.nl + p.nv], r[:, p.nz + p.nl + p.nv:nvar])
        rec, rroot = z(nvar + 5), z(nvar + 5) + 3
        before = g.stats()
        g.Solve([s], [data], [cut(rec)], [out], 0, cut(rroot), root_out)
        assert g.stats()["rccl_ops"] - before["rccl_ops"] == (2 if self_send == "1" else 0)
        for a, b in zip(cut(rroot), ref):
            assert torch.equal(a, b)
        s.close()
    # configs[4]: the sweep, the applied inputs gathered once at the end
    T, steps = 32, 6
    p = fx.synthetic_mpc_batch(T, first_id=40)
    A, Bm = fx.quadrotor_model()
    z = lambda n: torch.zeros((T, n), dtype=torch.float64, device=dev)
    s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=T)
    r = s.RecedingSweep(up(p), z(p.nz), z(p.nl), z(p.nv), z(p.nv), A, Bm, steps, retire=True, log_inputs=True)
    s.close()
    s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=T)
    u = torch.zeros((steps, T, p.nu), dtype=torch.float64, device=dev)
    ru = torch.full((steps, T, p.nu), 5.0, dtype=torch.float64, device=dev)
    out = torch.zeros((T, 40), dtype=torch.uint8, device=dev)
    before = g.stats()
    st = g.RecedingSweep([s], [up(p)], [(z(p.nz), z(p.nl), z(p.nv), z(p.nv))], [out], A, Bm, steps, True, [u], 0, ru)
    assert g.stats()["gathers"] == before["gathers"] + 1
    assert torch.equal(ru, r["u"]) and torch.equal(u, r["u"])
    assert np.array_equal(st[:, 0].astype(np.int64), r["stats"]["newton_sum"])
    s.close()
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("counts,root", [((40, 24), 1), ((0, 48), 0), ((17, 30, 17), 2)])
def test_several_shards_on_one_physical_device_equal_one_plain_call(monkeypatch, counts, root):
    """The index arithmetic of the sharded entries with MORE THAN ONE shard, on the one GPU a
    test box has: FBSTAB_HIP_SHARD_ALLOW_REPEATED_DEVICE=1 (test-only) lets a group name the
    same device several times; every shard has its own handle, stream and arrays and reaches
    the root through device copies.  Offsets of the shards on the root (`first`), packed and
    one-record-per-QP layouts, an empty shard, a root that is not shard 0, and the log offsets
    of the sharded sweep - all bitwise equal to ONE plain call over the whole batch.  (What
    this cannot cover is ncclCommInitAll over several devices and the sends between them.)"""
    import torch
    from fbstab_amd import hip_api
    from tools import fixtures as fx
    monkeypatch.setenv("FBSTAB_HIP_SHARD_ALLOW_REPEATED_DEVICE", "1")
    dev = torch.device("cuda:0")
    n = len(counts)
    B = sum(counts)
    first = np.concatenate([[0], np.cumsum(counts)]).astype(int)
    up = lambda p: {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
    g = hip_api.ShardGroup([0] * n)
    for kind, p in (("mpc", fx.synthetic_mpc_batch(B, first_id=4100)), ("dense", fx.synthetic_dense_batch(B, 30, 6, 70, first_id=11))):
        mk = (lambda mb: hip_api.FBstabMpcBatch(*p.sizes(), max_batch=mb)) if kind == "mpc" else \
            (lambda mb: hip_api.FBstabDenseBatch(p.nz, p.nl, p.nv, max_batch=mb))
        data = up(p)
        z = lambda rows, w: torch.zeros((rows, w), dtype=torch.float64, device=dev)
        plain = mk(B)
        ref = (z(B, p.nz), z(B, p.nl), z(B, p.nv), z(B, p.nv))
        out_ref = hip_api.out_to_numpy(plain.Solve(data, *ref))
        plain.close()
        solvers = [mk(max(c, 1)) for c in counts]
        shard_data = [{k: a[first[d]:first[d + 1]].contiguous() if a.shape[0] == B else a for k, a in data.items()}
                      for d in range(n)]
        # an empty shard still needs (B_d = 0)-row tensors for the binding; the library skips it
        nvar = p.nz + p.nl + 2 * p.nv
        cut = lambda r: (r[:, :p.nz], r[:, p.nz:p.nz + p.nl], r[:, p.nz + p.nl:p.nz + p.nl + p.nv], r[:, p.nz + p.nl + p.nv:nvar])
        for layout in ("packed", "record"):
            if layout == "packed":
                xs = [(z(c, p.nz), z(c, p.nl), z(c, p.nv), z(c, p.nv)) for c in counts]
                root_x = (z(B, p.nz) + 7, z(B, p.nl) + 7, z(B, p.nv) + 7, z(B, p.nv) + 7)
            else:
                recs = [z(c, nvar + 3) for c in counts]
                xs = [cut(r) for r in recs]
                rroot = z(B, nvar + 3) + 3
                root_x = cut(rroot)
            outs = [torch.zeros((c, 40), dtype=torch.uint8, device=dev) for c in counts]
            root_out = torch.zeros((B, 40), dtype=torch.uint8, device=dev)
            before = g.stats()
            g.Solve(solvers, shard_data, xs, outs, root, root_x, root_out)
            after = g.stats()
            assert after["gathers"] == before["gathers"] + 1 and after["rccl_ops"] == before["rccl_ops"]
            for a, b in zip(root_x, ref):
                assert torch.equal(a, b), (kind, layout)
            o = hip_api.out_to_numpy(root_out)
            for f in ("eflag", "residual", "newton_iters", "prox_iters", "initial_residual"):
                assert np.array_equal(o[f], out_ref[f]), f
            if layout == "record":  # the root's columns behind y of the LAST record are the caller's
                assert float(rroot[B - 1, nvar:].min()) == 3.0
        for s in solvers:
            s.close()
    # configs[4]: shard d's [steps][counts[d]][nu] log at root_u_log + steps * nu * first[d]
    steps = 5
    p = fx.synthetic_mpc_batch(B, first_id=77)
    A, Bm = fx.quadrotor_model()
    z = lambda rows, w: torch.zeros((rows, w), dtype=torch.float64, device=dev)
    s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=B)
    r = s.RecedingSweep(up(p), z(B, p.nz), z(B, p.nl), z(B, p.nv), z(B, p.nv), A, Bm, steps, retire=True, log_inputs=True)
    s.close()
    data = up(p)
    shard_data = [{k: a[first[d]:first[d + 1]].contiguous() for k, a in data.items()} for d in range(n)]
    solvers = [hip_api.FBstabMpcBatch(*p.sizes(), max_batch=max(c, 1)) for c in counts]
    us = [torch.zeros((steps, c, p.nu), dtype=torch.float64, device=dev) for c in counts]
    ru = torch.full((steps * B * p.nu,), 5.0, dtype=torch.float64, device=dev)
    outs = [torch.zeros((c, 40), dtype=torch.uint8, device=dev) for c in counts]
    xs = [(z(c, p.nz), z(c, p.nl), z(c, p.nv), z(c, p.nv)) for c in counts]
    st = g.RecedingSweep(solvers, shard_data, xs, outs, A, Bm, steps, True, us, root, ru)
    for d in range(n):
        blk = ru[steps * p.nu * first[d]: steps * p.nu * first[d + 1]].reshape(steps, counts[d], p.nu)
        assert torch.equal(blk, r["u"][:, first[d]:first[d + 1], :]), d
    assert np.array_equal(st[:, 0].astype(np.int64), r["stats"]["newton_sum"])
    for s in solvers:
        s.close()
    g.close()


@pytest.mark.gpu
def test_sharded_entry_rejects_a_bad_shard_before_anything_is_queued(monkeypatch):
    """a failure of shard d > 0 must not leave earlier shards' kernels in flight.  A null
    data pointer in the SECOND shard is found by the checks that run before the first launch: the
    call returns FBSTAB_HIP_ERR_ARGUMENT, the first shard's arrays are untouched, and the same
    group and handles solve the corrected call afterwards."""
    import ctypes as C
    import torch
    from fbstab_amd import hip_api
    from tools import fixtures as fx
    monkeypatch.setenv("FBSTAB_HIP_SHARD_ALLOW_REPEATED_DEVICE", "1")
    dev = torch.device("cuda:0")
    B = 16
    p = fx.synthetic_mpc_batch(2 * B, first_id=3)
    data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
    shard = [{k: a[d * B:(d + 1) * B].contiguous() for k, a in data.items()} for d in range(2)]
    z = lambda rows, w: torch.zeros((rows, w), dtype=torch.float64, device=dev)
    xs = [(z(B, p.nz) + 9, z(B, p.nl), z(B, p.nv), z(B, p.nv)) for _ in range(2)]
    outs = [torch.zeros((B, 40), dtype=torch.uint8, device=dev) for _ in range(2)]
    root_x = (z(2 * B, p.nz), z(2 * B, p.nl), z(2 * B, p.nv), z(2 * B, p.nv))
    root_out = torch.zeros((2 * B, 40), dtype=torch.uint8, device=dev)
    g = hip_api.ShardGroup([0, 0])
    solvers = [hip_api.FBstabMpcBatch(*p.sizes(), max_batch=B) for _ in range(2)]
    kind, bs, vs, counts, hs, op, var_lens = g._shards(solvers, hip_api.MPC_SEQ, solvers[0].seq_len, shard, xs, outs)
    rv = hip_api._VarBatch()
    hip_api._fill_var(rv, root_x, var_lens)
    lib = hip_api.load_library()
    bs[1].base[3] = None  # shard 1's q sequence
    rc = lib.fbstab_hip_mpc_solve_batch_sharded(g._g, hs, counts, bs, vs, op, 0, C.byref(rv), C.c_void_p(root_out.data_ptr()))
    assert rc == 1 and b"non-empty shard" in lib.fbstab_hip_last_error()
    torch.cuda.synchronize()
    assert float(xs[0][0].min()) == 9.0 and float(root_x[0].abs().max()) == 0.0  # nothing ran, nothing was gathered
    assert g.stats()["gathers"] == 0
    g.Solve(solvers, shard, [(z(B, p.nz), z(B, p.nl), z(B, p.nv), z(B, p.nv)) for _ in range(2)], outs, 0, root_x, root_out)
    o = hip_api.out_to_numpy(root_out)
    assert (o["eflag"] == 0).all() and float(root_x[0].abs().max()) > 0.0
    for s_ in solvers:
        s_.close()
    g.close()
