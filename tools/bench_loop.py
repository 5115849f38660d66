#!/usr/bin/env python3
"""The reference's outer iteration (equiSources.f90:1230-1843) for its three frequency groups with everything resident on the
device: computeOpacities -> diffuse sweep (96 directions) -> solveRateEquations, 256^3 cells.  Prints the time of each stage.
(The CPU rate of the same update is measured by tests/compare_with_reference.py.)  usage: bench_loop.py [n] [--save case.npz]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith('--') else 256
nc = n ** 3
g = np.load(os.path.join(ROOT, "tests", "golden", "chem_uvb_refined.npz"))   # the reference's rate-coefficient tables
box = 2.5e23
rho = 3.0e-26 * synthetic.lognormal_density(nc, seed=7, sigma_ln=0.7)
mp = float(np.float32(1.6726231e-24)); mn = float(np.float32(1.67492728e-24)); psi = float(np.float32(0.76))
nh, nhe = psi * rho / mp, (1 - psi) * rho / (2 * (mp + mn))
HI, HeI, HeII = 1e-3 * nh, 1e-2 * nhe, 0.3 * nhe
tgas = np.full(nc, 1.5e4)
beta = np.array([[6.3e-18, 1.2e-18, 2.0e-19], [0.0, 7.4e-18, 1.5e-18], [0.0, 0.0, 1.6e-18]])
ksi = g["ksi"]
ang = np.array([rt.pix2ang_nest(4, i) for i in range(96)])
phi, theta, w = ang[:, 0].copy(), ang[:, 1].copy(), np.full(96, 1.0 / 96)
uvb = np.array([2e-22, 1e-22, 3e-23])

st = rt.StellarTransfer()
st.set_uniform_grid(n, box)
st.set_rate_coefficients(float(g["logtem0"]), float(g["logtem9"]), float(g["dlogtem"]), g["k"])
st.set_medium(HI, HeI, HeII, rho, None, 0)
st.set_temperature(tgas)
J = torch.empty((3, nc), dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for it in range(5):
    t0 = time.perf_counter(); st.compute_opacities_from_medium(beta); torch.cuda.synchronize(); t1 = time.perf_counter()
    st.transport_device(phi, theta, w, uvb, J.data_ptr(), stream); torch.cuda.synchronize(); t2 = time.perf_counter()
    change = st.solve_rate_equations_device(J.data_ptr(), ksi); t3 = time.perf_counter()
    print(f"iteration {it}: opacities {1e3 * (t1 - t0):6.2f} ms, sweep {1e3 * (t2 - t1):7.2f} ms ({nc * 3 * 96 / (t2 - t1):.3e} updates/s), "
          f"equilibrium {1e3 * (t3 - t2):6.2f} ms ({nc / (t3 - t2):.3e} cells/s, {st.rate_equation_steps() / nc:.1f} bisection steps per cell), "
          f"largest change of a species fraction {change:.3e}", flush=True)
if "--save" in sys.argv:
    m = min(nc, 200000)
    np.savez(sys.argv[sys.argv.index("--save") + 1], n=n, box=box, rho=rho[:m], tgas=tgas[:m], HI=HI[:m], HeI=HeI[:m], HeII=HeII[:m],
             J=J[:, :m].cpu().numpy(), ksi=ksi, cells_per_s=nc / (t3 - t2))
