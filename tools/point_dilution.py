#!/usr/bin/env python3
"""Geometric dilution of a point source in a thin homogeneous box: the photons absorbed inside radius r grow like r
(absorbed per shell = Ndot n sigma dr), whatever the angular sampling.  Prints A(r)/r and the tracer's timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
st = rt.StellarTransfer()
box = 3.0e22
st.set_grid(n, np.zeros(n ** 3, np.int32), box)
pop = synthetic.stellar_population()
t0 = time.perf_counter(); total = st.stellar_beta_table(*pop, 3, 0.4, 2, 0.3); print(f"stellarBetaTable: {time.perf_counter() - t0:.3f} s, totalIntegral {total:.4e}")
tab = st.rate_tables()
tau_box = 0.01
HI = np.full(n ** 3, tau_box / (6.3e-18 * box))
z = np.zeros(n ** 3)
st.set_medium(HI, z, z, None, None, 0)
c = n // 2
src = (c * n + c) * n + c
for rep in range(5):
    t0 = time.perf_counter(); st.set_zero_rates(); t1 = time.perf_counter()
    hp = st.point_sources([src], [1.0]); t2 = time.perf_counter()
    k = st.rates(); t3 = time.perf_counter()
    print(f"zero {1e3 * (t1 - t0):.2f} ms, trace {1e3 * (t2 - t1):.2f} ms, download {1e3 * (t3 - t2):.2f} ms, highestPixelLevel {hp}")
rng = np.random.default_rng(3)
many = rng.choice(n ** 3, 256, replace=False)
for rep in range(3):
    st.set_zero_rates()
    t1 = time.perf_counter(); st.point_sources(many, np.ones(256)); t2 = time.perf_counter()
    print(f"256 sources: trace {1e3 * (t2 - t1):.2f} ms")
st.set_zero_rates(); st.point_sources([src], [1.0]); k = st.rates()
i, j, kk = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
r = np.sqrt((i - c) ** 2 + (j - c) ** 2 + (kk - c) ** 2).ravel()
for rr in (4, 8, 12, 16, 20, 24, 28, 31):
    if rr < n // 2:
        A = k[0][r <= rr].sum()
        print(f"r = {rr:3d} cells: absorbed(<r)/r = {A / rr:.6e}   per-photon {A / tab[0, 0, 0, 0, 0]:.4e}")
