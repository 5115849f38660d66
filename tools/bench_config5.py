#!/usr/bin/env python3
"""BASELINE configs[4]: 256^3, 96 directions, 8 frequency groups, source iterations S = (1-eps) J + eps B with eps = 1e-2 on a
plane-parallel opacity stratification; per-iteration rate and convergence.  One GPU as it stands; over N GPUs launched like bench.py,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_config5.py [n [iterations]]

one rank per GPU, frequency groups first, then directions (distributed.Shard2D): a rank keeps J and S of its groups, and with
N = 8 = the groups nothing is exchanged between the sweeps.  --rehearse-on-one-gpu: the ranks share GPU 0 and the collectives run
on host copies over gloo (what a one-GPU box can check of the branch)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic
from radiativetransfer_amd.iteration import SourceIteration

import torch.distributed as dist
from radiativetransfer_amd.distributed import Shard2D

rehearse = "--rehearse-on-one-gpu" in sys.argv
dump = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--dump=")]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 256
iters = int(args[1]) if len(args) > 1 else 50
world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
if rehearse:
    local = 0
torch.cuda.set_device(local)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if rehearse:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
nnu, ndir, eps = 8, 96, 1e-2
_, s_nu, uvb = synthetic.frequency_groups(nnu)
z = (np.arange(n) + 0.5) / n
tau_cell = 10.0 ** (-2.0 + 3.0 * z)                      # plane-parallel stratification: tau per cell 0.01 ... 10 along storage-i
kappa_host = (tau_cell * n)[None, :, None, None] * s_nu[:, None, None, None] * np.ones((1, 1, n, n))
kappa_host = np.ascontiguousarray(kappa_host.reshape(nnu, n ** 3))
ang = np.array([rt.pix2ang_nest(4, i) for i in range(ndir)])
phi, theta, w = ang[:, 0].copy(), ang[:, 1].copy(), np.full(ndir, 1.0 / ndir)
dev = torch.device("cuda", local)
sh = Shard2D(rank, world, nnu)
lo, hi = sh.groups                                       # this rank's frequency groups, its share of the directions
phi, theta, w = sh.directions(phi, theta, w)
kappa = torch.from_numpy(np.ascontiguousarray(kappa_host[lo:hi])).to(dev)
eng = rt.DiffuseTransfer(device=local)
for a in sys.argv[1:]:                                   # library options: --team=2 ...
    if a.startswith("--") and "=" in a and not a.startswith("--dump="):
        eng.set_option(a[2:].split("=")[0], int(a.split("=")[1]))
eng.set_uniform_grid(n, 1.0)
eng.set_opacity_device(hi - lo, kappa.data_ptr())
it = SourceIteration(eng, hi - lo, n ** 3, phi, theta, w, (uvb * 0.0 + 1e-30)[lo:hi], eps, (1e-21 * s_nu ** 0.5)[lo:hi], device=dev,
                     shard=sh, stage_on_host=rehearse)
if rank == 0 and world > 1:
    print(f"{world} ranks: {sh.describe()[:sh.describe().index(':')]}; rank 0 sweeps groups {lo}..{hi - 1} in {len(phi)} directions", flush=True)
upd = n ** 3 * nnu * ndir
for k in range(iters):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    change = it.step()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if rehearse or world == 1 else dev)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)       # an iteration lasts as long as its slowest rank
    dt = float(dt[0])
    if rank == 0 and (k < 3 or k % 10 == 9 or k == iters - 1):
        print(f"iteration {k + 1:3d}: {dt * 1e3:7.1f} ms  {upd / dt:.3e} updates/s  |dJ|/|J| = {change:.3e}", flush=True)
J = it.J
if dump:                                                 # (tests: this rank's groups after the last iteration)
    np.save(os.path.join(dump[0], f"J{rank}.npy"), J.cpu().numpy())
if rank == 0:
    print("J finite:", bool(torch.isfinite(J).all()), "min", float(J.min()), "max", float(J.max()), "S max", float(it.S.max()))
    Jc = J.reshape(hi - lo, n, n, n)
    print("J(nu=0) along the stratification, centre column:", [f"{float(Jc[0, i, n // 2, n // 2]):.3e}" for i in range(0, n, max(n // 8, 1))])
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
