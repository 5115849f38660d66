#!/usr/bin/env python3
"""Golden vector for the Fortran drop-in check: the reference's own sweep (oracle/_ref/ref_harness) on the refined cell array of
amr6_scattered_level2 with the direction list the reference driver itself uses -- all 12*4**(nAngularLevel-1) = 192 pixels
of nAngularLevel = 3 (definitionsModule.f90:41), weight 1./float(ndir) in single precision (equiSources.f90:1386).
Writes tests/golden/dropin_uvb_192dir.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as M  # noqa: E402
import _oracle  # noqa: E402


def main():
    g = np.load(os.path.join(HERE, "amr6_scattered_level2.npz"))
    phi, theta, _ = _oracle.healpix_directions(3)
    w = np.full(192, float(np.float32(1.0) / np.float32(192)))
    J, _ = M.run_reference(int(g["n"]), g["level"], g["kappa"], float(g["box"]), g["uvb"], phi, theta, w)
    path = os.path.join(HERE, "dropin_uvb_192dir.npz")
    np.savez_compressed(path, n=g["n"], level=g["level"], kappa=g["kappa"], box=g["box"], uvb=g["uvb"], phi=phi, theta=theta, w=w, J=J)
    print(f"dropin_uvb_192dir: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
