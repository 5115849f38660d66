"""Multi-GPU layout of the diffuse sweep: one process per GPU, frequency groups x directions sharded, J combined with RCCL.

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


# ---- frequency groups x directions ------------------------------------------------------------------

def decompose(world: int, nnu: int) -> Tuple[int, int]:
    """(r_nu, r_dir) with r_nu * r_dir == world, frequency groups first: r_nu is the largest divisor of `world` that also
    divides nnu.  With as many GPUs as groups (8 and 8) every rank owns one group for all directions and nothing has to be
    reduced; ranks beyond that split the direction list, and their partial J is summed."""
    import math
    r_nu = math.gcd(world, nnu)
    return r_nu, world // r_nu


class Shard2D:
    """Where one rank sits in the (frequency slice, direction slice) grid and what it has to sweep and exchange.

        sh = Shard2D(rank, world, nnu)
        nu_lo, nu_hi = sh.groups            # this rank's frequency groups
        phi, theta, w = sh.directions(phi, theta, w)
        J_full = sh.combine(J_local)        # J_local[nu_hi - nu_lo][ncell] -> J[nnu][ncell] on every rank
        J_slab = sh.exchange(J_local)       # ... or -> J[nnu][cells lo:hi of sh.slab(ncell)]: all groups, 1/world of the cells

    combine = sum over the ranks that hold the same groups for other directions (all-reduce, only if directions are split),
    then all-gather of the group slices.  Works on any torch.distributed backend (nccl = RCCL on the GPUs, gloo in the
    CPU tests)."""

    def __init__(self, rank: int, world: int, nnu: int):
        self.rank, self.world, self.nnu = rank, world, nnu
        self.r_nu, self.r_dir = decompose(world, nnu)
        self.i_nu, self.i_dir = rank % self.r_nu, rank // self.r_nu
        self.groups = shard_bounds(nnu, self.i_nu, self.r_nu)
        self._dir_group = self._nu_group = None
        self._made = False

    def directions(self, phi, theta, weight):
        return shard_directions(phi, theta, weight, self.i_dir, self.r_dir)

    def describe(self, mode: str = "combine") -> str:
        head = f"{self.r_nu} frequency slice(s) x {self.r_dir} direction slice(s): "
        if mode == "exchange":
            return (head + ("no reduction, " if self.r_dir == 1 else "reduce-scatter over the direction slices, ")
                    + ("nothing to exchange" if self.r_nu == 1 else "all-to-all between the frequency slices")
                    + f" -> every rank holds all groups for 1/{self.world} of the cells")
        return (head + ("no reduction, " if self.r_dir == 1 else "all-reduce over the direction slices, ")
                + ("nothing to gather" if self.r_nu == 1 else "all-gather of the frequency slices"))

    def _groups(self):
        import torch.distributed as dist
        if self._made or self.world == 1:
            return
        # every rank creates every group, in the same order
        for i_nu in range(self.r_nu):
            ranks = [i_nu + self.r_nu * j for j in range(self.r_dir)]
            g = dist.new_group(ranks) if self.r_dir > 1 else None
            if i_nu == self.i_nu:
                self._dir_group = g
        for i_dir in range(self.r_dir):
            ranks = [i + self.r_nu * i_dir for i in range(self.r_nu)]
            g = dist.new_group(ranks) if self.r_nu > 1 else None
            if i_dir == self.i_dir:
                self._nu_group = g
        self._made = True

    def slab(self, ncell: int) -> Tuple[int, int]:
        """Cells [lo, hi) this rank holds after `exchange`: `world` slabs of equal length (the last one shorter), slab
        index = rank."""
        length = -(-ncell // self.world)
        return min(self.rank * length, ncell), min((self.rank + 1) * length, ncell)

    def exchange(self, J_local):
        """J_local[groups of this rank][ncell] -> J[nnu][cells of this rank's slab]: every rank ends up with ALL frequency
        groups, summed over all directions, for 1/world of the cells -- what a per-cell consumer (the equilibrium update,
        equiSources.f90:3459-3677, needs every J_nu of a cell and nothing of other cells) wants, at 1/world of the
        traffic of `combine`: a reduce-scatter over the direction slices (only if directions are split), then an
        all-to-all between the frequency slices, every pair of ranks exchanging one (groups x slab) block directly --
        the pattern xGMI's point-to-point links carry best.  The slab is slab(ncell); nothing is padded in the result."""
        import torch
        import torch.distributed as dist
        nloc, ncell = J_local.shape
        if self.world == 1:
            return J_local
        if self.nnu % self.r_nu:
            raise ValueError("exchange needs equal frequency slices")
        self._groups()
        length = -(-ncell // self.world)
        padded = length * self.world
        src = J_local
        if padded != ncell:
            src = torch.zeros((nloc, padded), dtype=J_local.dtype, device=J_local.device)
            src[:, :ncell] = J_local
        # cells = [direction slice b][frequency slice a][length]: slab index b * r_nu + a = rank of (a, b)
        if self.r_dir > 1:
            # a group's row is [direction slice b][frequency slice a][length] as it lies: one reduce-scatter per row, no staging copy
            mine = torch.empty((nloc, self.r_nu * length), dtype=src.dtype, device=src.device)
            for g in range(nloc):
                dist.reduce_scatter_tensor(mine[g], src[g], op=dist.ReduceOp.SUM, group=self._dir_group)
        else:
            mine = src
        if self.r_nu > 1:
            # all-to-all between the frequency slices, straight out of J's own [group][cell] layout and straight into the rows of
            # the result: row (slice a, group g) of `out` is rank a's row g, cells of MY slab -- one contiguous run on both sides, so
            # every (partner, group) pair is one point-to-point transfer and nothing is permuted or staged on the device
            out = torch.empty((self.nnu, length), dtype=mine.dtype, device=mine.device)
            rows = mine.view(nloc, self.r_nu, length)
            ops = []
            for a in range(self.r_nu):
                peer = a + self.r_nu * self.i_dir           # global rank of frequency slice a in my direction slice
                for g in range(nloc):
                    if a == self.i_nu:
                        out[a * nloc + g].copy_(rows[g, a])
                    else:
                        ops.append(dist.P2POp(dist.isend, rows[g, a], peer))
                        ops.append(dist.P2POp(dist.irecv, out[a * nloc + g], peer))
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        else:
            out = mine.view(self.nnu, length)
        lo, hi = self.slab(ncell)
        return out[:, :hi - lo]

    def sum_directions(self, J_local, stage_on_host: bool = False):
        """In-place sum of J_local[groups of this rank][ncell] over the ranks that sweep the same groups for other directions: what
        a source iteration needs between two sweeps (every rank of a frequency slice needs the complete J of ITS groups for its
        next source function and nothing of the other slices': with as many ranks as groups, nothing is sent at all).
        stage_on_host: the collective runs on a host copy (gloo rehearsals with the ranks sharing one card)."""
        import torch.distributed as dist
        if self.world == 1 or self.r_dir == 1:
            return J_local
        self._groups()
        if stage_on_host and J_local.device.type != "cpu":
            host = J_local.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self._dir_group)
            J_local.copy_(host)
        else:
            dist.all_reduce(J_local, op=dist.ReduceOp.SUM, group=self._dir_group)
        return J_local

    def combine(self, J_local, out=None):
        """J_local: torch tensor [groups of this rank][ncell] (partial over directions if they are split).  Returns
        J[nnu][ncell] (`out` if given)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return J_local
        self._groups()
        if self.r_dir > 1:
            dist.all_reduce(J_local, op=dist.ReduceOp.SUM, group=self._dir_group)
        if self.r_nu == 1:
            return J_local
        if out is None:
            out = torch.empty((self.nnu, J_local.shape[1]), dtype=J_local.dtype, device=J_local.device)
        if self.nnu % self.r_nu == 0:
            dist.all_gather_into_tensor(out, J_local.contiguous(), group=self._nu_group)
            return out
        return gather_J(J_local, self.nnu, group=self._nu_group)
