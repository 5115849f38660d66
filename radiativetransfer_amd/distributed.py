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


def shard_groups(nnu: int, rank: int, world: int) -> Tuple[int, int]:
    """Frequency groups [lo, hi) of this rank when the groups rather than the directions are split (with
    as many GPUs as groups every J_nu is complete where it is computed and no reduction is needed; gather_J then only
    assembles the groups)."""
    return shard_bounds(nnu, rank, world)


def gather_J(J_local, nnu: int, group=None):
    """All-gather of per-rank group slices J_local[hi - lo][ncell] into J[nnu][ncell] on every rank (torch tensors)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return J_local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(nnu, r, world) for r in range(world)]
    most = max(hi - lo for lo, hi in sizes)
    # all_gather wants equal shapes: pad every slice to the largest share, cut the padding off afterwards
    mine = torch.zeros((most, J_local.shape[1]), dtype=J_local.dtype, device=J_local.device)
    mine[:J_local.shape[0]] = J_local
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.cat([p[:hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def shard_sources(cells, ndot, rank: int, world: int):
    """This rank's share of the star list.  Point sources are independent; the rates they deposit add up
    (equiSources.f90:3247-3260), so each rank traces its stars into its own rate array and allreduce_rates sums them."""
    lo, hi = shard_bounds(len(cells), rank, world)
    return np.asarray(cells)[lo:hi].copy(), np.asarray(ndot)[lo:hi].copy()


def allreduce_rates(rates_tensor, group=None):
    """In-place sum of rank-local point-source rates [6][ncell] over the process group; hand the result back to the
    library with StellarTransfer.set_rates before the equilibrium update."""
    return allreduce_J(rates_tensor, group)
