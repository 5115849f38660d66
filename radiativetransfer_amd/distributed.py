"""Multi-GPU layout of the diffuse sweep: one process per GPU, directions sharded, J summed with RCCL.

Each (direction, frequency) sweep is independent and reads the opacities only; the single coupling is
J_nu = sum over directions (transportRoutinesModule.f90:953-955).  Every GPU holds the whole grid (2 GB at
256^3 x 8 groups, of 288 GB), sweeps its share of the direction list into a local J and one all-reduce per
source iteration forms the direction-integrated J on every rank.  The reference has no parallel path of any
kind; this module is new work, exercised on CPU with the gloo backend (tests/test_distributed_gloo.py) and on
GPUs with nccl (= RCCL over xGMI) by bench.py.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_bounds(count: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of `count` items: rank r gets [lo, hi); sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, extra = divmod(count, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_directions(phi, theta, weight, rank: int, world: int):
    """This rank's slice of a direction list (weights are NOT renormalised: the global sum stays sum(weight))."""
    lo, hi = shard_bounds(len(phi), rank, world)
    return np.asarray(phi)[lo:hi].copy(), np.asarray(theta)[lo:hi].copy(), np.asarray(weight)[lo:hi].copy()


def allreduce_J(J_tensor, group=None):
    """In-place sum of a rank-local J (torch tensor, device or host) over the process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(J_tensor, op=dist.ReduceOp.SUM, group=group)
    return J_tensor
