#!/usr/bin/env python3
"""Refined cells scattered in clusters over the grid (what a cosmological cell array looks like, rather than BASELINE configs[3]'s one
central block): 128^3 base grid, `k` clusters of 4^3 refined base cells at fixed pseudo-random places, 8 groups, 96 directions.
The hybrid sweep (a box per cluster, forests in passes) against the forest path for the whole tree.
usage: bench_clusters.py [n] [clusters]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 128
k = int(args[1]) if len(args) > 1 else 8
rng = np.random.default_rng(5)
blocks = []
for _ in range(k):
    c = rng.integers(2, n - 6, 3)
    blocks += [(int(c[0]) + a, int(c[1]) + b, int(c[2]) + d) for a in range(4) for b in range(4) for d in range(4)]
level = synthetic.refine_levels(n, list(dict.fromkeys(blocks)), depth=1)
ncell = len(level)
nnu, ndir = 8, 96
rho = synthetic.lognormal_density(ncell, seed=4)
_, s_nu, uvb = synthetic.frequency_groups(nnu)
kappa = torch.from_numpy((0.1 * n / 3.0e22) * s_nu[:, None] * rho[None, :] * (2.0 ** level)[None, :]).to("cuda:0")
ang = np.array([rt.pix2ang_nest(4, i) for i in range(ndir)])
phi, theta, w = ang[:, 0].copy(), ang[:, 1].copy(), np.full(ndir, 1.0 / ndir)
J = torch.empty((nnu, ncell), dtype=torch.float64, device="cuda:0")
print(f"{ncell} leaves: {n}^3 base grid, {k} clusters of 4^3 refined base cells", flush=True)
ref = None
for hybrid in ((1,) if "--hybrid-only" in sys.argv else (1, 0)):
    eng = rt.DiffuseTransfer(device=0)
    eng.set_grid(n, level, 3.0e22)
    eng.set_option("hybrid", hybrid)
    if "--no-graph" in sys.argv:
        eng.set_option("graph", 0)
    for opt in sys.argv[1:]:
        if opt.startswith("--") and "=" in opt:      # e.g. --chunk=8 --pipelines=2
            eng.set_option(opt[2:].split("=")[0], int(opt.split("=")[1]))
    for it in range(4):
        t0 = time.perf_counter()
        eng.set_opacity_device(nnu, kappa.data_ptr())
        eng.transport_device(phi, theta, w, uvb, J.data_ptr(), 0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if it in (0, 3):
            print(f"hybrid {hybrid} iteration {it}: {dt * 1e3:8.1f} ms -> {ncell * nnu * ndir / dt:.3e} updates/s"
                  f"   (boxes {eng.counter('hybrid_boxes')}, passes {eng.counter('hybrid_passes')})", flush=True)
    if ref is None:
        ref = J.clone()
    else:
        print("max relative difference between the two:", float(((J - ref).abs() / ref.abs()).max()))
    eng.close()
