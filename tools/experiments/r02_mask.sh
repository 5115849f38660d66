#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mask
timeout -k 10 900 python -m pytest tests/test_hybrid_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -15 || exit 1
for lanes in 1 16 64; do
timeout -k 10 300 python3 - <<PY 2>&1 | grep -v amdgpu.ids
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
ang = np.array([rt.pix2ang_nest(4, i) for i in range(ndir)])
phi, theta, w = ang[:, 0].copy(), ang[:, 1].copy(), np.full(ndir, 1.0 / ndir)
eng = rt.DiffuseTransfer(device=0); eng.set_grid(n, level, 3.0e22); eng.set_option("box_lanes", $lanes)
J = torch.empty((nnu, ncell), dtype=torch.float64, device="cuda:0")
for it in range(4):
    t0 = time.perf_counter(); eng.set_opacity_device(nnu, kappa.data_ptr()); eng.transport_device(phi, theta, w, uvb, J.data_ptr(), 0); torch.cuda.synchronize()
    print("box_lanes $lanes iteration", it, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), "-> %.3e updates/s" % (ncell * nnu * ndir / (time.perf_counter() - t0)), flush=True)
np.save("gpurun_out/mask/J$lanes.npy", J.cpu().numpy()[:, ::97])
PY
done
python3 -c "
import numpy as np; a=np.load('gpurun_out/mask/J1.npy'); b=np.load('gpurun_out/mask/J64.npy'); print('max rel diff 1 vs 64 lanes', np.abs(a-b).max()/np.abs(b).max(), np.abs(a/b-1).max())"
