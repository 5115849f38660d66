#!/usr/bin/env python3
"""Whole-iteration rate when the boundary is used with HOST arrays (ftte_set_opacity + ftte_diffuse_sweep):
opacities go up and J comes back over PCIe every iteration.  For DESIGN.md only; never bench.py's value."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic
n, nnu, ndir = 256, 8, 96
kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=12345, tau_median=0.1)
ang = np.array([rt.pix2ang_nest(4, i) for i in range(ndir)])
phi, theta, w = ang[:, 0].copy(), ang[:, 1].copy(), np.full(ndir, 1.0 / ndir)
eng = rt.DiffuseTransfer()
eng.set_uniform_grid(n, box)
for rep in range(3):
    t0 = time.perf_counter()
    eng.set_opacity(kappa)
    J = eng.transport(phi, theta, w, uvb)
    dt = time.perf_counter() - t0
    print(f"host-array iteration {rep}: {dt * 1e3:.1f} ms -> {n ** 3 * nnu * ndir / dt:.3e} updates/s (PCIe-inclusive, pageable host memory)")
