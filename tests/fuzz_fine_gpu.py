#!/usr/bin/env python3
"""Randomised cross-check of the fine-level bricks of fully refined blocks (csrc/ftte_hybrid.cpp, option fine_bricks) against the
same sweep with the block left to the segment forests and against the forest path of the whole tree, which the parity tests pin
to the oracle (run by hand on a GPU box: python tests/fuzz_fine_gpu.py [cases] [seed]).  A cube of 32 (sometimes 64) base cells
refined once, anywhere in the grid including at the domain boundary, sometimes with other refined cells elsewhere (the block is
then not eligible and the case checks the fall-back); no emission, a source function or the reference's emissivity term; the
options that shape the sweep.  J to the rounding of the sum over directions."""
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
    worst, taken = 0.0, 0
    for case in range(cases):
        n = int(rng.choice([64, 72, 96, 128]))
        q = 64 if n == 128 and rng.random() < 0.3 else 32
        nnu = int(rng.integers(1, 4))
        corner = [int(rng.choice([0, n - q, rng.integers(0, n - q + 1)], p=[0.15, 0.15, 0.7])) for _ in range(3)]
        blocks = [(corner[0] + a, corner[1] + b, corner[2] + c) for a in range(q) for b in range(q) for c in range(q)]
        extra = rng.random() < 0.2
        if extra:   # a second patch somewhere: more than one cluster, or a block that is no longer a cube
            blocks.append(tuple(int(x) for x in rng.integers(0, n, 3)))
        level = synthetic.refine_levels(n, list(dict.fromkeys(blocks)), depth=1)
        nc = level.size
        kappa = rng.lognormal(0, 1.0, (nnu, nc)) * n * 10 ** rng.uniform(-2, 0) * (2.0 ** level)[None, :]
        uvb = 10 ** rng.uniform(-23, -20, nnu)
        phi, theta, _ = O.healpix_directions(int(rng.integers(1, 3)))
        pick = np.sort(rng.choice(phi.size, int(rng.integers(1, min(phi.size, 20) + 1)), replace=False))
        phi, theta = phi[pick], theta[pick]
        w = rng.uniform(0.1, 1.0, pick.size) / pick.size
        emission = str(rng.choice(["none", "source", "eta"], p=[0.5, 0.3, 0.2]))
        X = rng.random((nnu, nc)) * (uvb[:, None] if emission == "source" else uvb[:, None] * kappa.mean())
        opts = dict(chunk=int(rng.choice([0, 4, 8, 16])), fine_chunk=int(rng.choice([0, 4, 8, 16, 32])), group=int(rng.choice([0, 1, 2, 4])),
                    share=int(rng.integers(0, 3)), pipelines=int(rng.integers(1, 5)), box_lanes=int(rng.choice([1, 4, 64])))
        t0 = time.perf_counter()
        with rt.DiffuseTransfer() as eng:
            eng.set_grid(n, level, 1.0)
            eng.set_opacity(kappa)
            if emission == "source":
                eng.set_source_function(X)
            elif emission == "eta":
                eng.set_emissivity(X)
            for k, v in opts.items():
                eng.set_option(k, v)
            J = eng.transport(phi, theta, w, uvb)
            again = eng.transport(phi, theta, w, uvb)
            fine = eng.counter("fine_block")
            eng.set_option("fine_bricks", 0)
            forest_block = eng.transport(phi, theta, w, uvb)
            eng.set_option("hybrid", 0)
            ref = eng.transport(phi, theta, w, uvb)
        err = max(float(np.max(np.abs(J - ref) / np.abs(ref))), float(np.max(np.abs(forest_block - ref) / np.abs(ref))))
        same = bool(np.array_equal(J, again))
        worst = max(worst, err)
        taken += 1 if fine else 0
        ok = err < 1e-13 and same and np.all(np.isfinite(J))
        print(f"case {case:3d}: n={n:3d} q={q} at {corner} nnu={nnu} ndir={pick.size:2d} {emission:6s} extra={int(extra)} {opts}: fine block {fine}: "
              f"max rel diff {err:.2e}{'' if same else ' NOT REPRODUCIBLE'} ({time.perf_counter() - t0:.1f} s){'' if ok else '   <-- FAIL'}", flush=True)
        if not ok:
            sys.exit(1)
    print("worst", worst, "cases", cases, "through fine bricks", taken)


if __name__ == "__main__":
    main()
