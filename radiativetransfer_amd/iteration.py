"""Source iteration on top of the sweep (BASELINE configs[4]: strong scattering, many iterations).

The reference's own iteration couples J to the opacities through its chemistry (solveRateEquations, out of scope
here); its transport has the emission term switched off.  For scattering problems the build defines the usual
two-level-atom source function

    S_nu = (1 - eps) J_nu + eps B_nu ,       Iout = Iin exp(-tau) + S (1 - exp(-tau))   (ftte_set_source_function)

and iterates  J^{k+1} = Lambda[S(J^k)]  (Lambda iteration): one sweep per iteration, J and S stay on the device.  Over several
ranks (`shard`: distributed.Shard2D, frequency groups first, then directions) a rank keeps J and S of ITS groups only -- the
source function of a group needs that group's J and nothing else --, so with as many ranks as groups nothing is exchanged at all,
and where the directions are split too one all-reduce of the rank's groups over its direction slices closes an iteration.
Fixed points: J = B wherever the medium is thick; J = inflow in radiative equilibrium with the boundary (tested).
"""
from __future__ import annotations

from typing import Optional

import numpy as np


class SourceIteration:
    def __init__(self, engine, nnu: int, ncell: int, phi, theta, weight, uvb, epsilon: float, planck, device="cuda:0",
                 group=None, shard=None, stage_on_host: bool = False):
        """engine: a DiffuseTransfer with grid and opacities set; planck: B_nu, [nnu] or [nnu][ncell];
        phi/theta/weight: THIS rank's share of the direction list (weights of all ranks sum to the quadrature's total).
        group: all ranks sweep all groups for a share of the directions (one all-reduce of J per iteration over `group`).
        shard: a Shard2D -- nnu, uvb, planck and the engine's opacities are then those of THIS rank's groups (shard.groups), the
        directions its share (shard.directions); stage_on_host as in Shard2D.sum_directions."""
        import torch
        self.torch = torch
        self.engine = engine
        self.phi, self.theta, self.weight = (np.ascontiguousarray(a, dtype=np.float64) for a in (phi, theta, weight))
        self.uvb = np.ascontiguousarray(uvb, dtype=np.float64)
        self.eps = float(epsilon)
        self.group = group
        self.shard, self.stage_on_host = shard, stage_on_host
        dev = torch.device(device)
        B = torch.as_tensor(np.asarray(planck, dtype=np.float64), device=dev)
        self.B = B[:, None].expand(nnu, ncell) if B.dim() == 1 else B
        self.J = torch.zeros((nnu, ncell), dtype=torch.float64, device=dev)
        self.S = torch.empty_like(self.J)
        self.iterations = 0

    def step(self) -> float:
        """One iteration; returns max |J_new - J_old| / max |J_new| (the convergence measure of SURVEY.md section 8(d))."""
        torch = self.torch
        torch.mul(self.J, 1.0 - self.eps, out=self.S)
        self.S.add_(self.B, alpha=self.eps)
        J_old = self.J.clone()
        stream = torch.cuda.current_stream().cuda_stream
        torch.cuda.current_stream().synchronize()  # S is read by the library on the same stream; keep the hand-over simple
        self.engine.set_source_function_device(self.S.data_ptr())
        self.engine.transport_device(self.phi, self.theta, self.weight, self.uvb, self.J.data_ptr(), stream)
        if self.shard is not None:
            self.shard.sum_directions(self.J, self.stage_on_host)
        else:
            from .distributed import allreduce_J
            allreduce_J(self.J, self.group)
        self.iterations += 1
        norms = torch.stack([(self.J - J_old).abs().max(), self.J.abs().max()])
        if self.shard is not None and self.shard.world > 1:  # the measure is over all groups: the largest of every rank's
            import torch.distributed as dist
            if self.stage_on_host:
                norms = norms.cpu()
            dist.all_reduce(norms, op=dist.ReduceOp.MAX)
        return float(norms[0] / norms[1])

    def run(self, iterations: int, tol: Optional[float] = None):
        history = []
        for _ in range(iterations):
            history.append(self.step())
            if tol is not None and history[-1] < tol:
                break
        return history
