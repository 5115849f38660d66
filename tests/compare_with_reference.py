#!/usr/bin/env python3
"""CPU legs of the side measurements (test infrastructure: this file may use oracle/, the tools may not).

  config4 : runs tools/bench_config4.py (device) and then the reference's own star loop (oracle/_ref/point_harness) on the
            same 128^3 + refined-patch case on this box's host; prints both times and the worst deviation of the rates.
  loop    : runs tools/bench_loop.py (device) and times the C restatement of solveRateEquations on a 200 000-cell sample.

usage: python tests/compare_with_reference.py config4|loop [n]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))


def config4(n):
    import make_golden_point as M
    from radiativetransfer_amd import synthetic
    with tempfile.TemporaryDirectory() as tmp:
        case = os.path.join(tmp, "case.npz")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_config4.py"), str(n), "--no-diffuse", "--save", case], check=True)
        c = np.load(case)
    if not os.path.exists(M.HARNESS):
        sys.exit("oracle/_ref/point_harness not built (make -C oracle ref): no reference timing")
    pop = synthetic.stellar_population()
    t0 = time.perf_counter()
    ref = M.run_reference(int(c["n"]), c["level"], c["HI"], c["HeI"], c["HeII"], c["rho"], c["abun2"], float(c["box"]), 0,
                          np.array([int(c["src"])]), np.array([int(c["weight"])]), pop, int(c["isp"]), int(c["im"]), float(c["csp"]),
                          float(c["cm"]), np.zeros((1, 4)), npixlevel=1)
    t_ref = time.perf_counter() - t0
    k, tab = c["rates"], c["tables"]
    scale = np.abs(ref["krate"]).max(axis=1, keepdims=True)
    err = np.abs(k - ref["krate"]) / (np.abs(ref["krate"]) + 1e-4 * scale)
    secs = ref.get("trace_seconds", float("nan"))
    print(f"reference (1 host core): star loop {secs:.2f} s (whole harness run {t_ref:.1f} s); device tracer {float(c['trace_ms']):.2f} ms: "
          f"{secs / (float(c['trace_ms']) * 1e-3):.0f} x; worst |device - reference| / (|reference| + 1e-4 max) = {err.max():.2e}; "
          f"tables worst rel {np.abs(tab / ref['tables'] - 1).max():.1e}", flush=True)


def loop(n):
    import _oracle as O
    with tempfile.TemporaryDirectory() as tmp:
        case = os.path.join(tmp, "case.npz")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_loop.py"), str(n), "--save", case], check=True)
        c = np.load(case)
    g = np.load(os.path.join(HERE, "golden", "chem_uvb_refined.npz"))
    m = c["rho"].size
    t0 = time.perf_counter()
    O.solve_rate_equations(int(c["n"]), np.zeros(m, np.int32), float(c["box"]), c["rho"], c["tgas"], c["HI"], c["HeI"], c["HeII"], None, True,
                           c["J"], c["ksi"], None, 0.0, float(g["logtem0"]), float(g["logtem9"]), float(g["dlogtem"]), g["k"])
    dt = time.perf_counter() - t0
    print(f"CPU (C restatement of solveRateEquations, one core, {m} cells): {m / dt:.3e} cells/s; device {float(c['cells_per_s']):.3e} cells/s")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "config4"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else (128 if which == "config4" else 256)
    {"config4": config4, "loop": loop}[which](size)
