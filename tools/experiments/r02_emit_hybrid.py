import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic
n = 128; q = n // 4; lo = n // 2 - q // 2
level = synthetic.refine_levels(n, [(lo + a, lo + b, lo + c) for a in range(q) for b in range(q) for c in range(q)], depth=1)
ncell = len(level); nnu, ndir = 8, 96
rho = synthetic.lognormal_density(ncell, seed=4)
_, s_nu, uvb = synthetic.frequency_groups(nnu)
kappa = torch.from_numpy((0.1 * n / 3.0e22) * s_nu[:, None] * rho[None, :] * (2.0 ** level)[None, :]).to("cuda:0")
S = torch.full((nnu, ncell), 1e-22, dtype=torch.float64, device="cuda:0")
ang = np.array([rt.pix2ang_nest(4, i) for i in range(ndir)])
phi, theta, w = ang[:, 0].copy(), ang[:, 1].copy(), np.full(ndir, 1.0 / ndir)
J = torch.empty((nnu, ncell), dtype=torch.float64, device="cuda:0")
for hybrid in (1, 0):
    eng = rt.DiffuseTransfer(device=0); eng.set_grid(n, level, 3.0e22); eng.set_option("hybrid", hybrid)
    for it in range(4):
        t0 = time.perf_counter(); eng.set_opacity_device(nnu, kappa.data_ptr()); eng.set_source_function_device(S.data_ptr())
        eng.transport_device(phi, theta, w, uvb, J.data_ptr(), 0); torch.cuda.synchronize()
        if it == 3: print("source iteration on the configs[3] tree, hybrid", hybrid, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), "boxes", eng.counter("hybrid_boxes"), flush=True)
    eng.close()
