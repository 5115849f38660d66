#!/usr/bin/env python3
"""BASELINE configs[4] on ONE GPU: 256^3, 96 directions, 8 frequency groups, source iterations
S = (1-eps) J + eps B with eps = 1e-2 on a plane-parallel opacity stratification; per-iteration rate and convergence."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic
from radiativetransfer_amd.iteration import SourceIteration

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
nnu, ndir, eps = 8, 96, 1e-2
_, s_nu, uvb = synthetic.frequency_groups(nnu)
z = (np.arange(n) + 0.5) / n
tau_cell = 10.0 ** (-2.0 + 3.0 * z)                      # plane-parallel stratification: tau per cell 0.01 ... 10 along storage-i
kappa_host = (tau_cell * n)[None, :, None, None] * s_nu[:, None, None, None] * np.ones((1, 1, n, n))
kappa_host = np.ascontiguousarray(kappa_host.reshape(nnu, n ** 3))
ang = np.array([rt.pix2ang_nest(4, i) for i in range(ndir)])
phi, theta, w = ang[:, 0].copy(), ang[:, 1].copy(), np.full(ndir, 1.0 / ndir)
dev = torch.device("cuda", 0)
kappa = torch.from_numpy(kappa_host).to(dev)
eng = rt.DiffuseTransfer(device=0)
for a in sys.argv[3:]:                                   # library options: --team=2 ...
    if a.startswith("--") and "=" in a:
        eng.set_option(a[2:].split("=")[0], int(a.split("=")[1]))
eng.set_uniform_grid(n, 1.0)
eng.set_opacity_device(nnu, kappa.data_ptr())
it = SourceIteration(eng, nnu, n ** 3, phi, theta, w, uvb * 0.0 + 1e-30, eps, 1e-21 * s_nu ** 0.5)
upd = n ** 3 * nnu * ndir
for k in range(iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    change = it.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if k < 3 or k % 10 == 9 or k == iters - 1:
        print(f"iteration {k + 1:3d}: {dt * 1e3:7.1f} ms  {upd / dt:.3e} updates/s  |dJ|/|J| = {change:.3e}", flush=True)
J = it.J
print("J finite:", bool(torch.isfinite(J).all()), "min", float(J.min()), "max", float(J.max()), "S max", float(it.S.max()))
Jc = J.reshape(nnu, n, n, n)
print("J(nu=0) along the stratification, centre column:", [f"{float(Jc[0, i, n // 2, n // 2]):.3e}" for i in range(0, n, max(n // 8, 1))])
