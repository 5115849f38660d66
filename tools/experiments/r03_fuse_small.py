"""Round 3: thin forest levels in one launch (option forest_fuse) on SMALL refined trees, where every level is a launch of a few
hundred threads: the forest path of the whole tree, sweep time per call."""
import sys, time
import numpy as np
import torch
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic

for n, blocks, depth in ((16, [(7, 7, 7), (8, 8, 8)], 2), (32, [(14 + a, 15 + b, 16) for a in range(3) for b in range(3)], 2), (64, [(30 + a, 31 + b, 33 + c) for a in range(3) for b in range(2) for c in range(4)], 1)):
    level = synthetic.refine_levels(n, blocks, depth=depth)
    nnu = 3
    rho = synthetic.lognormal_density(len(level), seed=2)
    _, s_nu, uvb = synthetic.frequency_groups(nnu)
    kappa = (0.15 * n) * s_nu[:, None] * rho[None, :] * (2.0 ** level)[None, :]
    ang = np.array([rt.pix2ang_nest(2, i) for i in range(48)])
    phi, theta, w = ang[:, 0].copy(), ang[:, 1].copy(), np.full(48, 1.0 / 48)
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        e.set_option("hybrid", 0)
        for fuse in (0, 4096, 1 << 24):
            e.set_option("forest_fuse", fuse)
            J = e.transport(phi, theta, w, uvb)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                J = e.transport(phi, theta, w, uvb)
            torch.cuda.synchronize()
            print(f"n={n:3d} leaves={len(level):7d} forest_fuse={fuse:9d}: {(time.perf_counter() - t0) / 5 * 1e3:8.2f} ms per call (host arrays in and out)", flush=True)
