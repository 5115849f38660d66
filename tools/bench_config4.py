#!/usr/bin/env python3
"""BASELINE configs[3]: 128^3 base grid with the central 32^3 block refined once (2 326 528 leaves), one GPU.
  * diffuse part: 8 frequency groups, 96 directions, the segment-forest path; plan-build time (host, once per tree +
    direction list) and the per-iteration rate;
  * point source (the Stromgren-sphere set-up): one star in the centre of the refined patch, homogeneous hydrogen;
    stellarBetaTable on the device and the splitting tracer.  (The reference's own tracer is timed on the same case by
    tests/compare_with_reference.py, which may use the checker; this tool does not.)
usage: bench_config4.py [n] [--no-diffuse] [--save case.npz]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic

def flag(name, default=None):
    if name in sys.argv:
        return int(sys.argv[sys.argv.index(name) + 1])
    return default


skip = {sys.argv.index(k) + 1 for k in ("--chunk", "--group", "--hybrid", "--pipelines", "--graph", "--hybrid_slots", "--fine_bricks", "--fine_chunk", "--forest_fuse", "--save") if k in sys.argv}
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and i not in skip]
n = int(args[0]) if args else 128
q = n // 4
blocks = [(n // 2 - q // 2 + a, n // 2 - q // 2 + b, n // 2 - q // 2 + c) for a in range(q) for b in range(q) for c in range(q)]
t0 = time.perf_counter()
level = synthetic.refine_levels(n, blocks, depth=1)
ncell = len(level)
print(f"{ncell} leaves ({n}^3 base, central {q}^3 block refined once); levels built in {time.perf_counter() - t0:.1f} s", flush=True)
eng = rt.StellarTransfer(device=0)
t0 = time.perf_counter(); eng.set_grid(n, level, 3.0e22); print(f"set_grid (tree rebuild): {time.perf_counter() - t0:.2f} s", flush=True)
for key in ("chunk", "group", "hybrid", "pipelines", "graph", "hybrid_slots", "fine_bricks", "fine_chunk", "forest_fuse"):
    if flag("--" + key) is not None:
        eng.set_option(key, flag("--" + key))

if "--no-diffuse" not in sys.argv:
    nnu, ndir = 8, 96
    rho = synthetic.lognormal_density(ncell, seed=4)
    _, s_nu, uvb = synthetic.frequency_groups(nnu)
    kappa_host = (0.1 * n / 3.0e22) * s_nu[:, None] * rho[None, :] * (2.0 ** level)[None, :]
    ang = np.array([rt.pix2ang_nest(4, i) for i in range(ndir)])
    phi, theta, w = ang[:, 0].copy(), ang[:, 1].copy(), np.full(ndir, 1.0 / ndir)
    dev = torch.device("cuda", 0)
    kappa = torch.from_numpy(kappa_host).to(dev)
    J = torch.empty((nnu, ncell), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for it in range(4):
        t0 = time.perf_counter()
        eng.set_opacity_device(nnu, kappa.data_ptr())
        eng.transport_device(phi, theta, w, uvb, J.data_ptr(), stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        upd = ncell * nnu * ndir
        kms = sum(ms for ms, _ in eng.launch_records())
        print(f"diffuse iteration {it}: {dt * 1e3:9.1f} ms ({'includes building and uploading 96 forests' if it == 0 else 'plan cached'}); "
              f"device {kms:8.1f} ms -> {upd / dt:.3e} updates/s", flush=True)
    print("J range", float(J.min()), float(J.max()), "uvb", uvb[0], uvb[-1])
    del kappa, J

if "--no-point" in sys.argv:
    sys.exit(0)
# ---- the point source
box = 3.0e22
pop = synthetic.stellar_population()
isp, csp, im, cm = 3, 0.4, 2, 0.3
t0 = time.perf_counter(); total = eng.stellar_beta_table(*pop, isp, csp, im, cm); t_table = time.perf_counter() - t0
tau_box = 6.0  # hydrogen optical depth across the box at threshold: the front sits well inside
HI = np.full(ncell, tau_box / (6.3e-18 * box))
HeI, HeII = 0.08 * HI, 1e-3 * HI
rho, abun2 = HI * 1.67e-24 / 0.76, np.full(ncell, 0.02)
eng.set_medium(HI, HeI, HeII, rho, abun2, 0)
centre = [n // 2, n // 2, n // 2, 2, 2, 2]  # the fine cell just past the centre of the box
src = eng.locate_cell(centre)
weight = 1000
for rep in range(3):
    eng.set_zero_rates()
    t0 = time.perf_counter(); hp = eng.point_sources([src], [float(weight)]); t_trace = time.perf_counter() - t0
k = eng.rates()
tab = eng.rate_tables()
emitted = tab[0, 0, 0, 0, 0] * weight
print(f"point source: stellarBetaTable {t_table * 1e3:.1f} ms (device), tracer {t_trace * 1e3:.2f} ms, highestPixelLevel {hp}, "
      f"absorbed/emitted (HI) = {k[0].sum() / emitted:.6f}", flush=True)

if "--save" in sys.argv:   # for tests/compare_with_reference.py, which times the reference on the same case
    np.savez(sys.argv[sys.argv.index("--save") + 1], n=n, level=level, HI=HI, HeI=HeI, HeII=HeII, rho=rho, abun2=abun2, box=box, src=src,
             weight=weight, isp=isp, csp=csp, im=im, cm=cm, rates=k, tables=tab, trace_ms=t_trace * 1e3)
