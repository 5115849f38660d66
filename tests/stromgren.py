#!/usr/bin/env python3
"""Stromgren-sphere test (SURVEY.md 8(c) known-answer test 6, BASELINE configs[3] set-up): a star in the centre of a nested
refined patch in homogeneous hydrogen; the ionisation front  dx/dt = (1 - x) Gamma - alpha_B n_H x^2  is followed for ~15 recombination times with the
device tracer supplying the absorbed photons per cell at every step.  In equilibrium the recombinations in the box balance the star's
photons, so the ionised volume sum(x^2 V) equals (4 pi / 3) R_S^3 with R_S = (3 Ndot / 4 pi alpha_B n_H^2)^(1/3), and the
front sits at R_S.
usage: stromgren.py [n] [steps] [R_S in base cells]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ALPHA_B = 2.59e-13  # cm^3/s, case B at 1e4 K


def nested_patch_levels(n):
    from radiativetransfer_amd import synthetic
    q = n // 4
    lo = n // 2 - q // 2
    return synthetic.refine_levels(n, [(lo + a, lo + b, lo + c) for a in range(q) for b in range(q) for c in range(q)], depth=1)


def cell_geometry(n, level):
    """centres (box units) and volumes (box units^3) of the leaves of a one-level nested patch, cell-array order"""
    q = n // 4
    lo = n // 2 - q // 2
    pos, vol = [], []
    off = (np.indices((2, 2, 2)).reshape(3, -1).T + 0.5) / 2.0
    for i in range(n):
        for j in range(n):
            inside_ij = lo <= i < lo + q and lo <= j < lo + q
            for k in range(n):
                if inside_ij and lo <= k < lo + q:
                    for o in off:
                        pos.append(((i + o[0]) / n, (j + o[1]) / n, (k + o[2]) / n))
                        vol.append(1.0 / (8 * n ** 3))
                else:
                    pos.append(((i + 0.5) / n, (j + 0.5) / n, (k + 0.5) / n))
                    vol.append(1.0 / n ** 3)
    pos, vol = np.array(pos), np.array(vol)
    assert len(vol) == len(level)
    return pos, vol


def run(st, n, tables, iterations=60, rs_cells=None, box=8.0e22, weight=1.0, verbose=False, dt_rec=0.25):
    level = nested_patch_levels(n)
    pos, vol = cell_geometry(n, level)
    vol_cm3 = vol * box ** 3
    ncell = len(level)
    st.set_grid(n, level, box)
    st.set_rate_tables(tables)
    src = st.locate_cell([n // 2, n // 2, n // 2, 2, 2, 2])
    centre = pos[src]
    r = np.sqrt(((pos - centre) ** 2).sum(axis=1)) * box
    ndot = float(np.asarray(tables).reshape(6, -1)[0, 0]) * weight  # photons/s above 13.6 eV
    # density that puts the Stromgren radius at rs_cells base cells.  The reference's tracer stops splitting at pixel level 6
    # (12 288 rays): beyond sqrt(12288 / 4 pi) = 31 cells its rays are sparser than the cells and the sphere is no longer filled
    r_s = (min(0.3 * n, 24.0) if rs_cells is None else rs_cells) * box / n
    n_h = np.sqrt(3.0 * ndot / (4.0 * np.pi * ALPHA_B * r_s ** 3))
    zeros = np.zeros(ncell)
    n_hi = np.full(ncell, n_h)
    t_trace = 0.0
    for it in range(iterations):
        st.set_medium(n_hi, zeros, zeros, None, None, 0)
        st.set_zero_rates()
        t0 = time.perf_counter()
        st.point_sources([src], [weight])
        t_trace += time.perf_counter() - t0
        absorbed = st.rates()[0]                        # photons/s absorbed by HI in each cell
        gamma = absorbed / (n_hi * vol_cm3)             # per neutral atom
        # one step of  dx/dt = (1 - x) Gamma - alpha n x^2  over dt = dt_rec recombination times: implicit in x with Gamma
        # frozen (unconditionally stable in the thin, highly ionised interior), and never more ionisations than photons
        # absorbed (the limit that holds in a front cell, which absorbs what enters it whatever its neutral fraction)
        x_old = 1.0 - n_hi / n_h
        a, dt = ALPHA_B * n_h, dt_rec / (ALPHA_B * n_h)
        bq = 1.0 + gamma * dt
        x_imp = (-bq + np.sqrt(bq * bq + 4.0 * a * dt * (x_old + gamma * dt))) / (2.0 * a * dt)
        x_cap = x_old + dt * (absorbed / (n_h * vol_cm3) - a * x_old * x_old)
        x = np.clip(np.minimum(x_imp, x_cap), 0.0, 1.0 - 1e-9)
        n_hi = (1.0 - x) * n_h
        if verbose and (it % 10 >= 8 or it == iterations - 1):
            xi = 1.0 - n_hi / n_h
            print(f"iteration {it + 1:3d}: ionised volume / Stromgren volume = {(xi ** 2 * vol_cm3).sum() / (4 * np.pi / 3 * r_s ** 3):.4f}, "
                  f"absorbed / emitted = {absorbed.sum() / ndot:.4f}", flush=True)
    xi = 1.0 - n_hi / n_h
    if os.environ.get("STROMGREN_DEBUG"):
        recomb = ALPHA_B * n_h ** 2 * xi ** 2 * vol_cm3
        res = absorbed - recomb
        worst = np.argsort(-np.abs(res))[:8]
        print("sum absorbed", absorbed.sum() / ndot, "sum recomb", recomb.sum() / ndot)
        for c in worst:
            print(c, "r/R_S", r[c] / r_s, "x", xi[c], "absorbed", absorbed[c] / ndot, "recomb", recomb[c] / ndot, "n_hi/n", n_hi[c] / n_h)
    # radius of the half-ionised surface from the shell-averaged profile
    edges = np.linspace(0, 0.5 * box, n // 2 + 1)
    which = np.digitize(r, edges) - 1
    prof = np.array([(xi[which == b] * vol[which == b]).sum() / max(vol[which == b].sum(), 1e-300) for b in range(len(edges) - 1)])
    mid = 0.5 * (edges[1:] + edges[:-1])
    below = np.where(prof < 0.5)[0]
    b = below[0] if len(below) else len(prof) - 1
    r_half = mid[b - 1] + (prof[b - 1] - 0.5) / max(prof[b - 1] - prof[b], 1e-300) * (mid[b] - mid[b - 1]) if b > 0 else mid[0]
    return {"n_h": n_h, "r_s": r_s, "r_half": r_half, "volume_ratio": (xi ** 2 * vol_cm3).sum() / (4 * np.pi / 3 * r_s ** 3),
            "absorbed_fraction": absorbed.sum() / ndot, "cell": box / n, "trace_ms": 1e3 * t_trace / iterations, "ncell": ncell}


if __name__ == "__main__":
    import radiativetransfer_amd as rt
    from radiativetransfer_amd import synthetic
    args = [a for a in sys.argv[1:]]
    n = int(args[0]) if args else 128
    iters = int(args[1]) if len(args) > 1 else 60
    rs = float(args[2]) if len(args) > 2 else None
    st = rt.StellarTransfer()
    st.set_uniform_grid(4, 1.0)
    st.stellar_beta_table(*synthetic.stellar_population(), 3, 0.4, 2, 0.3)
    out = run(st, n, st.rate_tables(), iters, rs_cells=rs, verbose=True)
    print(f"{out['ncell']} cells; n_H = {out['n_h']:.3e} cm^-3; R_S = {out['r_s']:.4e} cm = {out['r_s'] / out['cell']:.2f} base cells; half-ionised radius / R_S = "
          f"{out['r_half'] / out['r_s']:.4f}; ionised volume / Stromgren volume = {out['volume_ratio']:.4f}; absorbed / emitted = "
          f"{out['absorbed_fraction']:.4f}; tracer {out['trace_ms']:.2f} ms per iteration")
