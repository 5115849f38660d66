#!/usr/bin/env python3
"""GPU-side diagnostic: single-direction sweeps for every izone against the oracle evaluated with the
device arithmetic (bitwise expected).  Prints one line per case; used while bringing kernels up."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
import radiativetransfer_amd as rt  # noqa: E402
from radiativetransfer_amd import synthetic  # noqa: E402


def one_per_izone():
    phi, theta, _ = O.healpix_directions(3)
    pick = {}
    for p, t in zip(phi, theta):
        pick.setdefault(O.fold_direction(p, t)[2], (p, t))
    return pick


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [16, 70]
    pick = one_per_izone()
    eng = rt.DiffuseTransfer()
    nbad = 0
    for n in sizes:
        kappa, uvb, box = synthetic.uniform_workload(n, 2, seed=n, tau_median=0.3)
        eng.set_uniform_grid(n, box)
        eng.set_opacity(kappa)
        for rows, stack in ((4, 1), (8, 1), (16, 1), (8, 4), (8, 2), (4, 8), (4, 4)):
            eng.set_option("rows", rows)
            eng.set_option("stack", stack)
            for z in range(1, 25):
                p, t = pick[z]
                phi, theta, w = np.array([p]), np.array([t]), np.array([0.37])
                J = eng.transport(phi, theta, w, uvb)
                ref = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
                same = np.array_equal(J, ref)
                if not same:
                    nbad += 1
                    diff = np.abs(J - ref) / np.abs(ref)
                    bad = np.argwhere(J != ref)
                    cells = bad[:4].tolist()
                    idx = [np.unravel_index(c[1], (n, n, n)) for c in cells]
                    print(f"n={n} rows={rows} stack={stack} izone={z:2d} MISMATCH count={len(bad)}/{J.size} maxrel={diff.max():.3e} "
                          f"first={[(int(a), int(b), int(c)) for a, b, c in idx]} nan={np.isnan(J).sum()}")
                else:
                    print(f"n={n} rows={rows} stack={stack} izone={z:2d} bitwise ok")
        # all 24 at once (three layouts, slots, merge)
        eng.set_option("rows", 8)
        eng.set_option("stack", 4)
        for slots in (1, 4):
            eng.set_option("slots", slots)
            ps = np.array([pick[z][0] for z in range(1, 25)]); ts = np.array([pick[z][1] for z in range(1, 25)])
            ws = np.full(24, 1 / 24)
            t0 = time.time()
            J = eng.transport(ps, ts, ws, uvb)
            dt = time.time() - t0
            ref = O.sweep_uniform(n, kappa, box, ps, ts, ws, uvb, arith=O.ARITH_DEVICE)
            print(f"n={n} 24 dirs slots={slots}: max rel diff vs serial-order oracle {np.max(np.abs(J - ref) / np.abs(ref)):.3e} "
                  f"({dt * 1e3:.1f} ms)  launches={eng.launch_records()}")
    print("MISMATCHING CASES:", nbad)
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
