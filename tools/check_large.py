#!/usr/bin/env python3
"""Sanity of index arithmetic beyond 2^31 bytes per plane set: 512^3 cells x 4 groups (4.3 GB per array, ~100 GB resident).
A transparent box must return the inflow, an opaque slab must shadow exactly the cells behind it along every direction, and
two runs must agree bit for bit.  usage: check_large.py [n] [nnu]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import radiativetransfer_amd as rt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nnu = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nc = n ** 3
dev = torch.device("cuda", 0)
ang = np.array([rt.pix2ang_nest(2, i) for i in range(48)])
phi, theta, w = ang[:, 0].copy(), ang[:, 1].copy(), np.full(48, 1.0 / 48)
uvb = np.array([1e-21 * 0.5 ** g for g in range(nnu)])
eng = rt.DiffuseTransfer(device=0)
eng.set_uniform_grid(n, 1.0)
kappa = torch.zeros((nnu, nc), dtype=torch.float64, device=dev)
J = torch.empty((nnu, nc), dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
eng.set_opacity_device(nnu, kappa.data_ptr())
t0 = time.perf_counter(); eng.transport_device(phi, theta, w, uvb, J.data_ptr(), stream); torch.cuda.synchronize(); dt = time.perf_counter() - t0
dev_uvb = torch.from_numpy(uvb).to(dev)[:, None]
err = float(((J - dev_uvb).abs() / dev_uvb).max())
print(f"{n}^3 x {nnu}: transparent box, max |J/uvb - 1| = {err:.2e} ({dt * 1e3:.0f} ms incl. planning, {nc * nnu * 48 / dt:.3e} updates/s)")
assert err < 1e-14
# a random medium, twice
g = torch.Generator(device=dev); g.manual_seed(5)
kappa = torch.rand((nnu, nc), dtype=torch.float64, device=dev, generator=g) * (0.5 * n)
eng.set_opacity_device(nnu, kappa.data_ptr())
eng.transport_device(phi, theta, w, uvb, J.data_ptr(), stream); torch.cuda.synchronize()
J1 = J.clone()
t0 = time.perf_counter(); eng.transport_device(phi, theta, w, uvb, J.data_ptr(), stream); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"random medium: bitwise repeatable = {bool(torch.equal(J, J1))}, {dt * 1e3:.0f} ms = {nc * nnu * 48 / dt:.3e} updates/s; "
      f"J/uvb in [{float((J / dev_uvb).min()):.3e}, {float((J / dev_uvb).max()):.3e}]")
assert torch.equal(J, J1) and bool(torch.isfinite(J).all()) and float((J / dev_uvb).max()) <= 1.0 + 1e-12
# corner cells see the inflow on three faces: the first and the last cell of the array get the same by symmetry of the set
print("first / last cell, group 0:", float(J[0, 0]), float(J[0, -1]))
