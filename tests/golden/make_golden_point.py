#!/usr/bin/env python3
"""Golden vectors for the point-source rows (P1-P3) and pix2ang_nest (A2), from the reference's own code:
oracle/_ref/point_harness (see oracle/point_harness.f90 and oracle/Makefile for how the contained procedures of the
reference's main program get compiled).  Run in the build container:

    make -C oracle ref && python tests/golden/make_golden_point.py

The reference's data files (Starburst99 spectra, dust parameter tables) are not in its repository: the spectrum and the
dust fit below are synthetic, chosen only to be smooth and positive.  DATA only is stored.
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from radiativetransfer_amd import synthetic  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "point_harness")
NT = 11 ** 4


def synthetic_population():
    """a_smc[7][5], wavelength[1221] (cm, ascending), specificLuminosity[5][37][1221] (log10 erg/s/A)."""
    lam_um = np.array([0.042, 0.08, 0.22, 9.7, 18.0, 25.0, 0.067])
    a_smc = np.stack([lam_um, np.array([185.0, 27.0, 0.005, 0.010, 0.012, 0.030, 10.0]),
                      np.array([90.0, 5.5, -1.95, -1.95, -1.8, 0.0, 1.9]), np.array([2.0, 4.0, 2.0, 2.0, 2.0, 2.0, 4.0]),
                      np.array([2.0, 4.0, 2.0, 2.0, 2.0, 2.0, 15.0])], axis=1)  # (7,5)
    lam_A = np.logspace(np.log10(91.0), np.log10(1.6e6), 1221)
    wavelength = lam_A * 1e-8
    hc_k = 1.43877688e8  # h c / k in Angstrom Kelvin
    spec = np.empty((5, 37, 1221))
    for im in range(5):
        for isp in range(37):
            T = 5.0e4 * (1.0 - 0.015 * isp) * (1.0 + 0.03 * im)
            x = hc_k / (lam_A * T)
            planck = lam_A ** -5.0 / np.expm1(np.minimum(x, 600.0))
            spec[im, isp] = 36.0 + np.log10(planck / planck.max() + 1e-30) + 0.01 * isp
    return a_smc, wavelength, spec


def run_reference(n, level, HI, HeI, HeII, rho, abun2, box, dust, src_leaf, src_weight, pop, isp, im, csp, cm, samples,
                  npixlevel=3):
    a_smc, wavelength, spec = pop
    ncell, nsrc, nsample = len(level), len(src_leaf), samples.shape[0]
    with tempfile.TemporaryDirectory() as tmp:
        case, out = os.path.join(tmp, "case.bin"), os.path.join(tmp, "out.bin")
        with open(case, "wb") as f:
            f.write(struct.pack("<6i", n, ncell, nsrc, dust, nsample, npixlevel))
            f.write(struct.pack("<d", box))
            f.write(np.asarray(level, "<i4").tobytes())
            for a in (HI, HeI, HeII, rho, abun2):
                f.write(np.asarray(a, "<f8").tobytes())
            f.write((np.asarray(src_leaf, "<i4") + 1).tobytes())
            f.write(np.asarray(src_weight, "<i4").tobytes())
            f.write(np.asfortranarray(a_smc).tobytes(order="F"))                  # a_smc(7,5)
            f.write(np.asarray(wavelength, "<f8").tobytes())
            f.write(np.asfortranarray(spec).tobytes(order="F"))                   # (5,37,1221)
            f.write(struct.pack("<2i", isp, im))
            f.write(struct.pack("<2d", csp, cm))
            f.write(np.ascontiguousarray(samples, "<f8").tobytes())               # sample(4,nsample) Fortran == [nsample][4] C
        res = subprocess.run([HARNESS, case, out], capture_output=True, text=True)
        if res.returncode != 0 or not os.path.exists(out):
            raise RuntimeError(f"point_harness failed: {res.stdout[-500:]} {res.stderr[-500:]}")
        raw = np.fromfile(out, dtype=np.uint8)
    off = 0

    def take(count, dtype="<f8"):
        nonlocal off
        a = np.frombuffer(raw, dtype, count, off)
        off += count * np.dtype(dtype).itemsize
        return a.copy()
    o = {"totalIntegral": take(1)[0]}
    for line in res.stdout.splitlines():
        if "TRACE_SECONDS" in line:
            o["trace_seconds"] = float(line.split()[-1])  # the star loop alone, as the harness timed it
    o["tables"] = take(6 * NT).reshape(6, 11, 11, 11, 11)  # [table][idust][i3][i2][i1]  (Fortran order reversed)
    o["outputSigma"] = take(4 * 300).reshape(4, 300)
    o["rates"] = take(nsample * 6).reshape(nsample, 3, 2)   # Fortran rates(2,3,nsample)
    pix = []
    for L in range(1, npixlevel + 1):
        pix.append(take(2 * 12 * 4 ** (L - 1)).reshape(-1, 2))
    o["pix"] = pix
    o["rmax"] = take(30)
    o["krate"] = take(6 * ncell).reshape(6, ncell)          # Fortran kout(ncell,6)
    o["highestPixelLevel"] = take(1, "<i4")[0]
    # per star: the tracer's escape bookkeeping (equiSources.f90:3198-3233, 3336-3345) and the src: line's fraction (:1342-1348)
    o["ndotRemaining"] = take(7 * nsrc).reshape(nsrc, 7)
    o["ndotBoundary"] = take(7 * nsrc).reshape(nsrc, 7)
    o["ndotDust"] = take(nsrc)
    o["ndotSpectrum"] = take(300 * nsrc).reshape(nsrc, 300)
    o["fraction"] = take(7 * nsrc).reshape(nsrc, 7)
    assert off == len(raw), (off, len(raw))
    return o


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build oracle/_ref/point_harness first: make -C oracle ref")
    pop = synthetic_population()
    rng = np.random.default_rng(123)
    samples = np.concatenate([rng.uniform(0, 10, (40, 4)), rng.uniform(0, 0.5, (20, 4)), np.zeros((1, 4)),
                              np.array([[10.5, 1, 1, 1], [9.99, 9.99, 9.99, 9.99]])])
    sigma = 6.3e-18

    # (1) homogeneous 16^3 box, one source in the central cell, no dust: tables, look-ups, pixels, rates
    n = 16
    ncell = n ** 3
    level = np.zeros(ncell, np.int32)
    HI = np.full(ncell, 0.25 / sigma)
    HeI = np.full(ncell, 0.02 / 7.42e-18)
    HeII = np.full(ncell, 0.002 / 1.58e-18)
    rho = np.full(ncell, 1e-24)
    abun2 = np.full(ncell, 0.01)
    src = [((8 - 1) * n + (8 - 1)) * n + (8 - 1)]
    o = run_reference(n, level, HI, HeI, HeII, rho, abun2, float(n), 0, src, [1], pop, 3, 2, 0.4, 0.7, samples, npixlevel=6)
    save("point16_homogeneous", n=n, level=level, HI=HI, HeI=HeI, HeII=HeII, rho=rho, abun2=abun2, box=float(n), dust=0,
         src_leaf=np.array(src), src_weight=np.array([1]), a_smc=pop[0], wavelength=pop[1],
         iSpectrum=3, iMetal=2, coefSpectrum=0.4, coefMetal=0.7, samples=samples, totalIntegral=o["totalIntegral"],
         tables=o["tables"], outputSigma=o["outputSigma"], rates=o["rates"], pix1=o["pix"][0], pix2=o["pix"][1],
         pix3=o["pix"][2], pix4=o["pix"][3], pix5=o["pix"][4], pix6=o["pix"][5], rmax=o["rmax"], krate=o["krate"], highestPixelLevel=o["highestPixelLevel"],
         **escape(o))
    print("  highest pixel level", o["highestPixelLevel"], " sum krate24", o["krate"][0].sum(), " totalIntegral", o["totalIntegral"])

    # (2) refined: 10^3 base, a 2x2x2 block refined once, log-normal densities, dust ~ HI, two sources (one in a fine leaf)
    n = 10
    blocks = [(4 + a, 4 + b, 5 + c) for a in range(2) for b in range(2) for c in range(2)]
    level = synthetic.refine_levels(n, blocks, depth=1)
    ncell = len(level)
    dens = synthetic.lognormal_density(ncell, seed=77, sigma_ln=0.7)
    HI = 0.3 / sigma * dens * (2.0 ** level)
    HeI = 0.03 / 7.42e-18 * dens
    HeII = 0.003 / 1.58e-18 * dens
    rho = 1e-24 * dens
    abun2 = 0.2 * np.ones(ncell)
    fine = int(np.nonzero(level == 1)[0][5])
    coarse = int(np.nonzero(level == 0)[0][123])
    o = run_reference(n, level, HI, HeI, HeII, rho, abun2, float(n), 1, [fine, coarse], [2, 1], pop, 10, 1, 0.25, 0.1,
                      samples[:8], npixlevel=6)
    save("point10_refined_dust", n=n, level=level, HI=HI, HeI=HeI, HeII=HeII, rho=rho, abun2=abun2, box=float(n), dust=1,
         src_leaf=np.array([fine, coarse]), src_weight=np.array([2, 1]), a_smc=pop[0], wavelength=pop[1],
         iSpectrum=10, iMetal=1, coefSpectrum=0.25, coefMetal=0.1, tables=o["tables"], krate=o["krate"],
         highestPixelLevel=o["highestPixelLevel"], **escape(o))
    print("  highest pixel level", o["highestPixelLevel"], " sum krate24", o["krate"][0].sum())

    # (3) escape fractions: a box of physical size (80 kpc: the output radii 0.1 ... 30 kpc lie inside it, 100 kpc outside),
    #     12^3 base with a refined 2x2x2 block around the first star, total optical depth of a few across the box, dust ~ total
    #     hydrogen, three stars (one in a fine leaf, one near a face so that part of its light leaves early, one with weight 3)
    n = 12
    kpc = 1.0e3 * 3.08568025e18
    box = 80.0 * kpc
    blocks = [(5 + a, 5 + b, 6 + c) for a in range(2) for b in range(2) for c in range(2)]
    level = synthetic.refine_levels(n, blocks, depth=1)
    ncell = len(level)
    dens = synthetic.lognormal_density(ncell, seed=31, sigma_ln=0.5)
    HI = 2.5 / (sigma * box) * dens
    HeI = 0.3 / (7.42e-18 * box) * dens
    HeII = 0.05 / (1.58e-18 * box) * dens
    rho = HI * 1.6726231e-24 / 0.76 * 50.0
    abun2 = 0.05 * np.ones(ncell)
    fine = int(np.nonzero(level == 1)[0][11])
    near_face = int(np.nonzero(level == 0)[0][7])
    inner = int(np.nonzero(level == 0)[0][900])
    o = run_reference(n, level, HI, HeI, HeII, rho, abun2, box, 2, [fine, near_face, inner], [1, 2, 3], pop, 7, 2, 0.6, 0.35,
                      samples[:4], npixlevel=6)
    save("point12_escape", n=n, level=level, HI=HI, HeI=HeI, HeII=HeII, rho=rho, abun2=abun2, box=box, dust=2,
         src_leaf=np.array([fine, near_face, inner]), src_weight=np.array([1, 2, 3]), a_smc=pop[0], wavelength=pop[1],
         iSpectrum=7, iMetal=2, coefSpectrum=0.6, coefMetal=0.35, tables=o["tables"], outputSigma=o["outputSigma"], krate=o["krate"],
         highestPixelLevel=o["highestPixelLevel"], **escape(o))
    print("  highest pixel level", o["highestPixelLevel"], " fraction", np.round(o["fraction"], 4).tolist())


def escape(o):
    return {k: o[k] for k in ("ndotRemaining", "ndotBoundary", "ndotDust", "ndotSpectrum", "fraction")}


if __name__ == "__main__":
    main()
