#!/usr/bin/env python3
"""Whole-iteration rate when the boundary is used with HOST arrays (ftte_set_grid + ftte_set_opacity + ftte_diffuse_sweep, the call
sequence of the Fortran drop-ins): opacities go up and J comes back over PCIe every iteration.  Pageable arrays (staged through
the library's pinned blocks) and arrays registered with ftte_host_register (DMA in place).  Also the refined config-4 tree: the first
call builds tree + forests, the later ones must not.  For DESIGN.md only; never bench.py's value."""
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
level = np.zeros(n ** 3, np.int32)
eng = rt.DiffuseTransfer()
J = np.empty((nnu, n ** 3))
for mode in ("pageable", "registered"):
    if mode == "registered":
        t0 = time.perf_counter(); eng.host_register(kappa); eng.host_register(J)
        print(f"registering 2 x {kappa.nbytes / 1e9:.2f} GB: {(time.perf_counter() - t0) * 1e3:.0f} ms (once)")
    for rep in range(4):
        t0 = time.perf_counter()
        eng.set_grid(n, level, box)          # the drop-in hands the level list over every time; unchanged: kept
        t1 = time.perf_counter()
        eng.set_opacity(kappa)
        t2 = time.perf_counter()
        eng.transport_into(phi, theta, w, uvb, J)
        t3 = time.perf_counter()
        print(f"{mode:10s} iteration {rep}: {(t3 - t0) * 1e3:7.1f} ms = set_grid {(t1 - t0) * 1e3:6.1f} + set_opacity {(t2 - t1) * 1e3:6.1f} + sweep and J back "
              f"{(t3 - t2) * 1e3:6.1f}  -> {n ** 3 * nnu * ndir / (t3 - t0):.3e} updates/s PCIe-inclusive", flush=True)
# the same as ONE call (ftte_diffuse_iteration): the frequency groups cross PCIe and are swept in overlapping lanes
for rep in range(4):
    t0 = time.perf_counter()
    eng.set_grid(n, level, box)
    eng.iterate_into(kappa, phi, theta, w, uvb, J)
    t3 = time.perf_counter()
    print(f"registered, one call, iteration {rep}: {(t3 - t0) * 1e3:7.1f} ms -> {n ** 3 * nnu * ndir / (t3 - t0):.3e} updates/s PCIe-inclusive", flush=True)
eng.host_unregister(kappa); eng.host_unregister(J)
for rep in range(3):
    t0 = time.perf_counter()
    eng.set_grid(n, level, box)
    eng.iterate_into(kappa, phi, theta, w, uvb, J)
    t3 = time.perf_counter()
    print(f"pageable,   one call, iteration {rep}: {(t3 - t0) * 1e3:7.1f} ms -> {n ** 3 * nnu * ndir / (t3 - t0):.3e} updates/s PCIe-inclusive", flush=True)
eng.close()

# BASELINE configs[3] tree through the same call sequence: set_grid + set_opacity + sweep per outer iteration
n = 128
q = n // 4
lo = n // 2 - q // 2
level = synthetic.refine_levels(n, [(lo + a, lo + b, lo + c) for a in range(q) for b in range(q) for c in range(q)], depth=1)
ncell = len(level)
rho = synthetic.lognormal_density(ncell, seed=4)
_, s_nu, uvb3 = synthetic.frequency_groups(3)
kappa = (0.1 * n / 3.0e22) * s_nu[:, None] * rho[None, :] * (2.0 ** level)[None, :]
phi, theta, w = (np.concatenate([a, a]) for a in (phi, theta, w))   # 192 entries as the reference's level-3 set has; reuse the 96
w = w / 2
J = np.empty((3, ncell))
eng = rt.DiffuseTransfer()
eng.host_register(kappa); eng.host_register(J)
for rep in range(3):
    t0 = time.perf_counter()
    eng.set_grid(n, level, 3.0e22)
    eng.set_opacity(kappa)
    eng.transport_into(phi, theta, w, uvb3, J)
    print(f"config-4 tree, 3 groups x 192 directions, call {rep}: {(time.perf_counter() - t0) * 1e3:8.1f} ms  (tree builds {eng.counter('grid_builds')}, "
          f"forest builds {eng.counter('forest_builds')})", flush=True)
