#!/usr/bin/env python3
"""How long the host takes to ENQUEUE one sweep (ftte_diffuse_sweep_device returns before the GPU is done) against how long the GPU
takes to run it, for the per-rank shapes of a frequency-sharded run: is a rank with one group waiting for its own launches?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic

n, ndir = 256, 96
ang = np.array([rt.pix2ang_nest(4, i) for i in range(ndir)])
phi, theta, w = ang[:, 0].copy(), ang[:, 1].copy(), np.full(ndir, 1.0 / ndir)
dev = torch.device("cuda", 0)
for nnu in (8, 4, 2, 1):
    kappa_host, uvb, box = synthetic.uniform_workload(n, nnu, seed=12345, tau_median=0.1)
    kappa = torch.from_numpy(kappa_host).to(dev)
    J = torch.empty((nnu, n ** 3), dtype=torch.float64, device=dev)
    eng = rt.DiffuseTransfer(device=0)
    for a in sys.argv[1:]:
        if a.startswith("--") and "=" in a:
            eng.set_option(a[2:].split("=")[0], int(a.split("=")[1]))
    eng.set_uniform_grid(n, box)
    eng.set_opacity_device(nnu, kappa.data_ptr())
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.transport_device(phi, theta, w, uvb, J.data_ptr())
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"nnu {nnu}: enqueue {1e3 * (t1 - t0):6.2f} ms, until done {1e3 * (t2 - t0):6.2f} ms, launches {sum(1 for _ in eng.launch_records())}", flush=True)
    eng.close()
