#!/usr/bin/env python3
"""Randomised cross-check of the hybrid sweep of refined cell arrays (bricks outside boxes around the refined cells, segment
forests inside, csrc/ftte_hybrid.cpp) against the forest path of the whole tree, which the parity tests pin to the oracle
(run by hand on a GPU box: python tests/fuzz_hybrid_gpu.py [cases] [seed]).  Grid sizes with and without ragged bricks, patches
anywhere including the domain boundary, one or two levels, direction subsets with unequal weights, and the options that shape
the sweep (chunk, group, share, pipelines, box_lanes, forest_batch).  J to the rounding of the sum over directions."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402
import radiativetransfer_amd as rt  # noqa: E402
from radiativetransfer_amd import synthetic  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    worst, hybrid_taken = 0.0, 0
    for case in range(cases):
        n = int(rng.choice([64, 72, 96, 100, 128, 136]))
        nnu = int(rng.integers(1, 4))
        blocks = []
        for _ in range(int(rng.integers(1, 6))):      # one to five clusters: boxes of their own where they lie apart
            size = rng.integers(1, 5, 3)
            corner = [int(rng.choice([0, n - s, rng.integers(0, n - s + 1)], p=[0.1, 0.1, 0.8])) for s in size]
            blocks += [(corner[0] + a, corner[1] + b, corner[2] + c) for a in range(size[0]) for b in range(size[1]) for c in range(size[2])]
        depth = int(rng.integers(1, 3))
        level = synthetic.refine_levels(n, list(dict.fromkeys(blocks)), depth=depth)
        nc = level.size
        kappa = rng.lognormal(0, 1.0, (nnu, nc)) * n * 10 ** rng.uniform(-2, 0) * (2.0 ** level)[None, :]
        uvb = 10 ** rng.uniform(-23, -20, nnu)
        phi, theta, _ = O.healpix_directions(int(rng.integers(1, 3)))
        pick = np.sort(rng.choice(phi.size, int(rng.integers(1, min(phi.size, 24) + 1)), replace=False))
        phi, theta = phi[pick], theta[pick]
        w = rng.uniform(0.1, 1.0, pick.size) / pick.size
        opts = dict(chunk=int(rng.choice([0, 2, 4, 8])), group=int(rng.choice([0, 1, 2, 4])), share=int(rng.integers(0, 3)),
                    pipelines=int(rng.integers(1, 5)), box_lanes=int(rng.choice([1, 2, 4, 16, 64])),
                    forest_batch=int(rng.choice([0, 0, 5])), hybrid_slots=int(rng.integers(0, 3)))
        t0 = time.perf_counter()
        with rt.DiffuseTransfer() as eng:
            eng.set_grid(n, level, 1.0)
            eng.set_opacity(kappa)
            for k, v in opts.items():
                eng.set_option(k, v)
            J = eng.transport(phi, theta, w, uvb)
            again = eng.transport(phi, theta, w, uvb)
            shape = (eng.counter("hybrid_boxes"), eng.counter("hybrid_passes"))
            eng.set_option("hybrid", 0)
            ref = eng.transport(phi, theta, w, uvb)
        err = float(np.max(np.abs(J - ref) / np.abs(ref)))
        same = bool(np.array_equal(J, again))
        worst = max(worst, err)
        ok = err < 1e-13 and same and np.all(np.isfinite(J))
        print(f"case {case:3d}: n={n:3d} nnu={nnu} ndir={pick.size:2d} depth={depth} refined base cells={len(set(blocks)):3d} {opts}: "
              f"boxes {shape[0]} passes {shape[1]}: max rel diff {err:.2e}{'' if same else ' NOT REPRODUCIBLE'} ({time.perf_counter() - t0:.1f} s){'' if ok else '   <-- FAIL'}", flush=True)
        hybrid_taken += 1
        if not ok:
            sys.exit(1)
    print("worst", worst, "cases", hybrid_taken)


if __name__ == "__main__":
    main()
