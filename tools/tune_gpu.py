#!/usr/bin/env python3
"""Timing sweep over the kernel's tuning knobs on the BASELINE workload (one process, one grid upload)."""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import radiativetransfer_amd as rt  # noqa: E402
from radiativetransfer_amd import synthetic  # noqa: E402


def main():
    n, nnu, ndir = 256, 8, 96
    combos = []
    for arg in sys.argv[1:]:
        f = [int(x) for x in arg.split(",")]
        combos.append(tuple(f + [1] * (4 - len(f))) if len(f) < 5 else tuple(f))
    if not combos:
        combos = [(8, 4, 6, 1), (8, 4, 6, 4), (8, 4, 6, 2), (4, 6, 6, 8), (4, 6, 6, 4), (4, 4, 6, 8)]
    kappa_host, uvb, box = synthetic.uniform_workload(n, nnu, seed=12345, tau_median=0.1)
    dev = torch.device("cuda", 0)
    kappa = torch.from_numpy(kappa_host).to(dev)
    J = torch.empty((nnu, n ** 3), dtype=torch.float64, device=dev)
    ang = np.array([rt.pix2ang_nest(4, i) for i in range(ndir)])
    phi, theta, w = ang[:, 0].copy(), ang[:, 1].copy(), np.full(ndir, 1.0 / ndir)
    eng = rt.DiffuseTransfer(device=0)
    eng.set_uniform_grid(n, box)
    stream = torch.cuda.current_stream().cuda_stream
    ref = None
    for combo in combos:
        rows, waves, slots, stack = combo[:4]
        eng.set_option("ldspad", combo[4] if len(combo) > 4 else 0)
        eng.set_option("rows", rows); eng.set_option("waves", waves); eng.set_option("slots", slots)
        eng.set_option("stack", stack)
        def step():
            eng.set_opacity_device(nnu, kappa.data_ptr())
            eng.transport_device(phi, theta, w, uvb, J.data_ptr(), stream)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 2
        kms = 0.0
        for _ in range(reps):
            step(); torch.cuda.synchronize()
            kms += sum(ms for ms, _ in eng.launch_records())
        dt = (time.perf_counter() - t0) / reps
        upd = n ** 3 * nnu * ndir
        chk = float(J.sum().item())
        if ref is None:
            ref = chk
        print(f"rows={rows:2d} stack={stack} waves={waves} slots={slots} ldspad={combo[4] if len(combo) > 4 else 0}: {dt * 1e3:7.2f} ms/step  {upd / dt:.3e} upd/s  "
              f"sweep kernels {kms / reps:7.2f} ms -> {upd * 24 / (kms / reps * 1e-3) / 1e9:6.0f} GB/s algorithmic "
              f"({upd * 24 / (kms / reps * 1e-3) / 8e12:.3f} of 8 TB/s)  checksum rel {abs(chk / ref - 1):.1e}", flush=True)


if __name__ == "__main__":
    main()
