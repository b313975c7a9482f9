"""Batch sharding across the GPUs of one node (one process per GPU).

The path shards embarrassingly: QPs are independent (the reference solver
objects share nothing between Solve calls, fbstab_mpc.cc:75-86), so rank g of W
owns the contiguous block of global instance ids ``[g*B, (g+1)*B)`` and there is
no collective on the data path.  The only exchange is ONE gather of the
solutions ``(z,l,v,y)`` and ``SolverOut`` records, fused into one record per QP,
to rank 0 at the end of a batch (RCCL over xGMI when the tensors live on GPUs; the same code runs over
gloo on CPU tensors in the tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def shard_range(rank: int, world: int, per_rank_batch: int) -> Tuple[int, int]:
    """Global instance ids ``[first, last)`` owned by ``rank`` (weak scaling:
    every rank owns ``per_rank_batch`` QPs)."""
    if not (0 <= rank < world) or per_rank_batch < 0:
        raise ValueError("bad shard request")
    return rank * per_rank_batch, (rank + 1) * per_rank_batch


OUT_DOUBLES = 5   # one 40-byte SolverOut record (include/fbstab_types.h) as float64 columns


def unpack_out(record):
    """SolverOut records (numpy structured array) from the last ``OUT_DOUBLES``
    float64 columns of gathered solution records."""
    import numpy as np
    from fbstab_amd.hip_api import OUT_DTYPE
    tail = record[:, -OUT_DOUBLES:].contiguous().cpu().numpy()
    return np.frombuffer(tail.tobytes(), dtype=OUT_DTYPE).copy()


def gather_solutions(x, out, dst: int = 0, record=None, gather_list: Optional[List] = None,
                     stack: bool = True):
    """ONE gather of this rank's results to ``dst``: the solutions ``x``
    (``(B, nz+nl+2nv)`` float64) and the ``out`` records (``(B, 40)`` uint8
    SolverOut) travel side by side in one ``(B, nvar + 5)`` float64 record per QP.
    ``record``: optional preallocated record buffer whose first columns ARE ``x``
    (then only the 40 bytes per QP of ``out`` are copied); ``gather_list``:
    optional preallocated receive buffers on ``dst`` - ``gather_list[dst]`` may be ``record`` itself
    (the root's own block is then not copied at all: it is resident where it was computed).  Returns ``(X, O)`` stacked in
    global instance order on ``dst`` (``O`` as uint8 ``(W*B, 40)``), ``(None,
    None)`` elsewhere.  ``stack=False`` (a caller that keeps its own receive
    buffers): no stacked copy is made on ``dst`` - at eight ranks that copy is
    1.1 GB per batch - and ``(gather_list, None)`` is returned there."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    rank = dist.get_rank()
    nvar = x.shape[1]
    if record is None:
        record = torch.empty((x.shape[0], nvar + OUT_DOUBLES), dtype=x.dtype, device=x.device)
        record[:, :nvar] = x
    record[:, nvar:] = out.view(torch.float64).view(-1, OUT_DOUBLES)
    if rank == dst:
        gather_list = gather_list or [torch.empty_like(record) for _ in range(world)]
    else:
        gather_list = None
    dist.gather(record, gather_list, dst=dst)
    if rank != dst:
        return None, None
    if not stack:
        return gather_list, None
    full = torch.cat(gather_list, dim=0)
    return full[:, :nvar], full[:, nvar:].contiguous().view(torch.uint8).view(-1, 40)


def gather_input_log(u_log, dst: int = 0, gather_list: Optional[List] = None):
    """The receding-horizon sweep (BASELINE configs[4]) over several ranks: every rank
    sweeps its own block of trajectories with no exchange at all - a trajectory's next
    QP depends on nothing but its own last solution - and the applied inputs
    ``u_log`` (``(steps, B, nu)`` float64, 32 bytes per trajectory and step) travel to
    ``dst`` in ONE gather when the sweep is over.  Returns the list of per-rank logs
    on ``dst`` (rank g's block = global trajectories ``[g*B, (g+1)*B)``), ``None``
    elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    rank = dist.get_rank()
    if rank == dst:
        gather_list = gather_list or [torch.empty_like(u_log) for _ in range(world)]
    else:
        gather_list = None
    dist.gather(u_log.contiguous(), gather_list, dst=dst)
    return gather_list
