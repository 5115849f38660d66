#!/usr/bin/env python3
"""Randomised cross-check of the library against the oracle (run by hand on a GPU box: python tests/fuzz_gpu.py [cases] [seed]).
Grid sizes, group counts, direction subsets with unequal weights, directions per launch, refined trees: J against the oracle
with the device arithmetic (summation order differs: relative 1e-13)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402
import radiativetransfer_amd as rt  # noqa: E402
from radiativetransfer_amd import synthetic  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    eng = rt.DiffuseTransfer()
    worst = 0.0
    for case in range(cases):
        n = int(rng.integers(3, 41)) if rng.random() < 0.8 else int(rng.choice([64, 70, 128]))
        nnu = int(rng.integers(1, 10))
        level_dirs = int(rng.integers(1, 4))
        phi, theta, _ = O.healpix_directions(level_dirs)
        pick = np.sort(rng.choice(phi.size, int(rng.integers(1, min(phi.size, 60) + 1)), replace=False))
        phi, theta = phi[pick], theta[pick]
        w = rng.uniform(0.1, 1.0, pick.size) / pick.size
        slots = int(rng.integers(1, 17))
        refined = rng.random() < 0.35 and n <= 12
        if refined:
            blocks = [tuple(int(x) for x in rng.integers(0, n, 3)) for _ in range(int(rng.integers(1, 5)))]
            level = synthetic.refine_levels(n, list(dict.fromkeys(blocks)), depth=int(rng.integers(1, 3)))
        else:
            level = np.zeros(n ** 3, np.int32)
        nc = level.size
        kappa = rng.lognormal(0, 1.2, (nnu, nc)) * n * 10 ** rng.uniform(-2, 0.5) * (2.0 ** level)[None, :]
        uvb = 10 ** rng.uniform(-23, -20, nnu)
        eng.set_option("slots", slots)
        # the organisation of the uniform-grid sweep: bricks with random shapes and sharing, now and then one launch with flags,
        # now and then the ray-following tiles; every form of the brick kernel
        opts = dict(engine=int(rng.choice([0, 0, 0, 1])), chunk=int(rng.choice([0, 1, 3, 4, 16, 32])), group=int(rng.integers(0, 6)),
                    share=int(rng.integers(0, 3)), lanes=int(rng.integers(1, 5)), dataflow=int(rng.choice([0, 0, 0, 2, 3, 3])),
                    team=int(rng.choice([-1, -1, 0, 2, 2])))
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.set_grid(n, level, 1.0)
        eng.set_opacity(kappa)
        J = eng.transport(phi, theta, w, uvb)
        if refined:
            ref = O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
        else:
            ref = O.sweep_uniform(n, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
        ref = ref[0] if isinstance(ref, tuple) else ref
        err = float(np.max(np.abs(J - ref) / np.abs(ref)))
        worst = max(worst, err)
        flag = "" if err < 1e-13 else "   <-- FAIL"
        print(f"case {case:3d}: n={n:3d} nnu={nnu} ndir={pick.size:3d} slots={slots:2d} {opts} {'refined' if refined else 'uniform'} "
              f"cells={nc:6d}: max rel diff {err:.2e}{flag}", flush=True)
        if err >= 1e-13:
            sys.exit(1)
    print("worst", worst)


if __name__ == "__main__":
    main()
