#!/usr/bin/env python3
"""Golden vectors for the ionisation-equilibrium update (solveRateEquations, equiSources.f90:3459-3677), produced by the
reference's own compiled code through oracle/_ref/chem_harness (build it first: make -C oracle ref).

    python tests/golden/make_golden_chem.py

Writes tests/golden/chem_uvb_refined.npz (J-driven rates + point-source rates on a refined cell array) and
chem_uniform_background.npz (uniform background with self-shielding).  The rate-coefficient tables k1a..k6a of the
reference's calc_rates are stored with the first case (float64, 5000 entries each).
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "chem_harness")
NRATEC = 5000


def run_reference(n, level, box, rho, tgas, HI, HeI, HeII, krate, J, run_uvb, ksi, uniform, threshold):
    ncell = len(level)
    with tempfile.TemporaryDirectory() as tmp:
        case, out = os.path.join(tmp, "case.bin"), os.path.join(tmp, "out.bin")
        with open(case, "wb") as f:
            f.write(struct.pack("<3i", n, ncell, int(run_uvb)))
            f.write(struct.pack("<d", box))
            f.write(np.asarray(level, "<i4").tobytes())
            for a in (rho, tgas, HI, HeI, HeII, krate[0], krate[1], krate[2], J[0], J[1], J[2]):   # f(ncell,11), column by column
                f.write(np.asarray(a, "<f8").tobytes())
            f.write(np.asarray(ksi, "<f8").reshape(3, 3).tobytes())   # [group][reaction] == Fortran ksiIn(reaction, group)
            f.write(np.asarray(uniform, "<f8").tobytes())
            f.write(struct.pack("<d", threshold))
        res = subprocess.run([HARNESS, case, out], capture_output=True, text=True)
        if res.returncode != 0 or not os.path.exists(out):
            raise RuntimeError(f"chem_harness failed: {res.stdout[-800:]} {res.stderr[-500:]}")
        raw = np.fromfile(out, dtype="<f8")
    o = {"logtem0": raw[0], "logtem9": raw[1], "dlogtem": raw[2], "k": raw[3:3 + 6 * NRATEC].reshape(6, NRATEC)}
    rest = raw[3 + 6 * NRATEC:]
    assert rest.size == 3 * ncell
    o["HI"], o["HeI"], o["HeII"] = rest[:ncell].copy(), rest[ncell:2 * ncell].copy(), rest[2 * ncell:].copy()
    return o


def synthetic_gas(rng, level, n, box):
    ncell = len(level)
    mp, mn = float(np.float32(1.6726231e-24)), float(np.float32(1.67492728e-24))
    psi = float(np.float32(0.76))
    rho = 10 ** rng.uniform(-27.0, -24.0, ncell)
    nh, nhe = psi * rho / mp, (1 - psi) * rho / (2 * (mp + mn))
    tgas = 10 ** rng.uniform(1.5, 6.5, ncell)
    tgas[::17] = 0.3        # below the table
    tgas[5::23] = 3.0e8     # above it
    HI = nh * 10 ** rng.uniform(-6, 0, ncell)
    HI[::11] = nh[::11] * 1.5          # more neutral hydrogen than hydrogen: the reference takes min(HI, nh)
    HeI = nhe * rng.uniform(0, 0.6, ncell)
    HeII = nhe * rng.uniform(0, 0.4, ncell)
    HeI[3::29] = nhe[3::29] * 0.9      # HeI + HeII > nhe: the reference's HeIII < 0 branch
    HeII[3::29] = nhe[3::29] * 0.5
    size = box / (2.0 ** level * n)
    krate = np.stack([HI, HeII, HeI]) * size ** 3 * 10 ** rng.uniform(-16, -11, (3, ncell))
    krate[:, ::7] = 0.0
    J = 10 ** rng.uniform(-24, -21.3, (3, ncell))   # stronger fields at these densities ionise helium so completely that
    # the reference's HeI comes out of its formula as -1e-16 and it stops (equiSources.f90:3643-3654)
    return rho, tgas, HI, HeI, HeII, krate, J


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build oracle/_ref/chem_harness first: make -C oracle ref")
    rng = np.random.default_rng(2024)
    amr = np.load(os.path.join(HERE, "amr6_scattered_level2.npz"))
    n, level, box = int(amr["n"]), amr["level"].astype(np.int32), 2.5e23
    rho, tgas, HI, HeI, HeII, krate, J = synthetic_gas(rng, level, n, box)
    ksi = np.array([[2.1e8, 0.0, 0.0], [6.0e7, 0.0, 1.4e8], [9.0e6, 3.5e7, 2.2e7]])   # [group][ksi24, ksi25, ksi26]
    uniform = np.array([3.0e-14, 1.0e-16, 2.0e-14])
    o = run_reference(n, level, box, rho, tgas, HI, HeI, HeII, krate, J, True, ksi, uniform, 0.0)
    save("chem_uvb_refined", n=n, level=level, box=box, rho=rho, tgas=tgas, HI=HI, HeI=HeI, HeII=HeII, krate=krate, J=J, ksi=ksi,
         logtem0=o["logtem0"], logtem9=o["logtem9"], dlogtem=o["dlogtem"], k=o["k"], HI_out=o["HI"], HeI_out=o["HeI"],
         HeII_out=o["HeII"])
    # uniform background, no point sources; the threshold shields roughly half of the cells
    n2 = 8
    level2 = np.zeros(n2 ** 3, np.int32)
    rho, tgas, HI, HeI, HeII, krate, J = synthetic_gas(rng, level2, n2, box)
    mfp = 1.0 / (np.minimum(HI, 0.76 * rho / 1.6726231e-24) * 6.3e-18 + HeI * 7.42e-18 + HeII * 1.58e-18)
    threshold = float(np.median(mfp))
    o = run_reference(n2, level2, box, rho, tgas, HI, HeI, HeII, np.zeros((3, n2 ** 3)), J, False, ksi, uniform, threshold)
    save("chem_uniform_background", n=n2, level=level2, box=box, rho=rho, tgas=tgas, HI=HI, HeI=HeI, HeII=HeII, uniform=uniform,
         threshold=threshold, HI_out=o["HI"], HeI_out=o["HeI"], HeII_out=o["HeII"])
    # assignUvbRadiation (transportRoutinesModule.f90:1056-1093): the optically thin alternative to the sweep, same cells
    uvb = np.array([1.0e-21, 4.0e-22, 1.0e-22])
    o = run_reference(n2, level2, box, rho, tgas, HI, HeI, HeII, np.zeros((3, n2 ** 3)), J, 2, ksi, uvb, threshold)
    save("thin_limit_uvb", n=n2, level=level2, box=box, rho=rho, HI=HI, HeI=HeI, HeII=HeII, uvb=uvb, threshold=threshold,
         J=np.stack([o["HI"], o["HeI"], o["HeII"]]))


if __name__ == "__main__":
    main()
