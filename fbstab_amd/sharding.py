"""Batch sharding across the GPUs of one node (one process per GPU).

The path shards embarrassingly: QPs are independent (the reference solver
objects share nothing between Solve calls, fbstab_mpc.cc:75-86), so rank g of W
owns the contiguous block of global instance ids ``[g*B, (g+1)*B)`` and there is
no collective on the data path.  The only exchange is ONE gather of the
solutions ``(z,l,v,y)`` and ``SolverOut`` records to rank 0 at the end of a
batch (RCCL over xGMI when the tensors live on GPUs; the same code runs over
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


def gather_solutions(x, out, dst: int = 0, gather_x: Optional[List] = None,
                     gather_out: Optional[List] = None):
    """One gather of this rank's solution records ``x`` (``(B, nz+nl+2nv)``
    float64) and ``out`` (``(B, 40)`` uint8 SolverOut records) to ``dst``.
    Returns ``(X, O)`` stacked in global instance order on ``dst``, ``(None,
    None)`` elsewhere.  ``gather_x/out`` may supply preallocated receive lists."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    rank = dist.get_rank()
    if rank == dst:
        gather_x = gather_x or [torch.empty_like(x) for _ in range(world)]
        gather_out = gather_out or [torch.empty_like(out) for _ in range(world)]
    else:
        gather_x = gather_out = None
    dist.gather(x, gather_x, dst=dst)
    dist.gather(out, gather_out, dst=dst)
    if rank != dst:
        return None, None
    return torch.cat(gather_x, dim=0), torch.cat(gather_out, dim=0)
