"""Synthetic inputs for the diffuse sweep.

The reference ships no sample data (its grid, star list and spectra files are not
in the repository, inputParameters:3-4), so every workload here is generated:
deterministically, from a seed, in cell-array order (definitionsModule.f90:323-326).
Pure numpy; nothing here touches the GPU.
"""
from __future__ import annotations

import numpy as np

_MASK = (1 << 64) - 1


def splitmix64(seed: int, count: int) -> np.ndarray:
    """`count` successive outputs of the splitmix64 generator seeded with `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & _MASK) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def standard_normal(seed: int, count: int) -> np.ndarray:
    """Box-Muller on splitmix64 uniforms (two uniforms per pair of normals)."""
    m = (count + 1) // 2
    bits = splitmix64(seed, 2 * m)
    u = ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    r = np.sqrt(-2.0 * np.log(u[0::2]))
    a = 2.0 * np.pi * u[1::2]
    out = np.empty(2 * m)
    out[0::2] = r * np.cos(a)
    out[1::2] = r * np.sin(a)
    return out[:count]


def lognormal_density(ncell: int, seed: int = 12345, sigma_ln: float = 1.0) -> np.ndarray:
    """rho(cell), median 1, log-normal with the given ln-sigma, cell-array order."""
    return np.exp(sigma_ln * standard_normal(seed, ncell))


def frequency_groups(nnu: int):
    """Group frequencies log-spaced 13.6-136 eV; returns (nu_ratio, s_nu, uvb_nu).

    s_nu = (nu/nu1)^-3 scales the opacity, uvb_nu = 1e-21 (nu/nu1)^-1.8 is the inflow
    (kept at the physical 1e-21 scale: the reference `transport` aborts unless the summed
    outgoing intensity stays below 1e-20, transportRoutinesModule.f90:680-688).
    """
    ratio = 10.0 ** (np.arange(nnu) / max(nnu - 1, 1)) if nnu > 1 else np.ones(1)
    return ratio, ratio ** -3.0, 1.0e-21 * ratio ** -1.8


def uniform_workload(n: int, nnu: int, seed: int = 12345, tau_median: float = 0.1, box: float = 1.0):
    """kappa[nnu][n^3], uvb[nnu], box for a uniform n^3 grid.

    kappa_nu(cell) = kappa0 * rho(cell) * s_nu with kappa0 chosen so that the median
    optical depth of one cell at the lowest frequency is `tau_median`.
    """
    ncell = n ** 3
    rho = lognormal_density(ncell, seed)
    _, s_nu, uvb = frequency_groups(nnu)
    kappa0 = tau_median / (box / n)
    kappa = (kappa0 * s_nu)[:, None] * rho[None, :]
    return np.ascontiguousarray(kappa), uvb, box


def refine_levels(n: int, blocks, depth: int = 1) -> np.ndarray:
    """Depth-first leaf `level` list (readCellArray.f90:154-187) for an n^3 base grid in
    which the base cells listed in `blocks` (0-based (i,j,k) tuples) are refined `depth`
    times (every child refined again until `depth`)."""
    marked = set(map(tuple, blocks))
    out = []
    def leaves(level):
        if level == depth:
            out.append(level)
        else:
            for _ in range(8):
                leaves(level + 1)
    for i in range(n):
        for j in range(n):
            for k in range(n):
                if (i, j, k) in marked:
                    for _ in range(8):
                        leaves(1)
                else:
                    out.append(0)
    return np.asarray(out, dtype=np.int32)


def stellar_population():
    """A stand-in for the reference's stellar library and dust-curve data files (neither ships with it): black-body
    spectra on the library's grid.  Returns a_smc[7][5] (rows: lambda_i [micron], a_i, b_i, p_i, q_i of the SMC fit the
    reference reads, dustModule.f90:16-21), wavelength[1221] (cm, ascending) and
    specificLuminosity[5][37][1221] (log10 erg/s/Angstrom; metallicity, age, wavelength)."""
    lam_um = np.array([0.042, 0.08, 0.22, 9.7, 18.0, 25.0, 0.067])
    a_smc = np.stack([lam_um, np.array([185.0, 27.0, 0.005, 0.010, 0.012, 0.030, 10.0]),
                      np.array([90.0, 5.5, -1.95, -1.95, -1.8, 0.0, 1.9]), np.array([2.0, 4.0, 2.0, 2.0, 2.0, 2.0, 4.0]),
                      np.array([2.0, 4.0, 2.0, 2.0, 2.0, 2.0, 15.0])], axis=1)
    lam_A = np.logspace(np.log10(91.0), np.log10(1.6e6), 1221)
    hc_k = 1.43877688e8  # h c / k in Angstrom Kelvin
    spec = np.empty((5, 37, 1221))
    for im in range(5):
        for isp in range(37):
            T = 5.0e4 * (1.0 - 0.015 * isp) * (1.0 + 0.03 * im)
            x = hc_k / (lam_A * T)
            planck = lam_A ** -5.0 / np.expm1(np.minimum(x, 600.0))
            spec[im, isp] = 36.0 + np.log10(planck / planck.max() + 1e-30) + 0.01 * isp
    return a_smc, lam_A * 1e-8, spec
