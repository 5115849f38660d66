#!/usr/bin/env python3
"""Throughput of the point-source tracer with many stars: 256^3 cells, log-normal hydrogen and helium, S stars at random
cells; without dust and with the dust approximation `completeSublimation`.  Prints ms per trace, stars/s, cell crossings/s.
usage: bench_point.py [n] [stars]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nc = n ** 3
box = 3.0e22
st = rt.StellarTransfer()
st.set_uniform_grid(n, box)
st.stellar_beta_table(*synthetic.stellar_population(), 3, 0.4, 2, 0.3)
rho = synthetic.lognormal_density(nc, seed=3, sigma_ln=1.0)
HI = rho * 2.0 / (6.3e-18 * box)          # hydrogen optical depth 2 across the box at mean density
HeI, HeII = 0.08 * HI, 0.01 * HI
rng = np.random.default_rng(5)
src = rng.choice(nc, S, replace=False)
ndot = rng.uniform(1, 3, S)
for dust, label in ((0, "no dust"), (1, "dust ~ HI")):
    st.set_medium(HI, HeI, HeII, rho * 1e-24, np.full(nc, 0.02), dust)
    for rep in range(3):
        st.set_zero_rates()
        t0 = time.perf_counter(); st.point_sources(src, ndot); dt = time.perf_counter() - t0
    steps = st.ray_steps()
    k = st.rates()
    print(f"{label:10s}: {S} stars in {n}^3: {dt * 1e3:8.2f} ms = {S / dt:9.1f} stars/s, {steps / dt:.3e} cell crossings/s "
          f"({steps / S:.0f} per star); absorbed HI fraction {k[0].sum() / (st.rate_tables()[0, 0, 0, 0, 0] * ndot.sum()):.4f}", flush=True)
