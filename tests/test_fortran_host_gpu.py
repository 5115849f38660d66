"""The Fortran host (fortran/ftte_demo_driver: ISO_C_BINDING module + libftte.so) on a real GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fortran", "ftte_demo_driver")


def test_fortran_demo_driver():
    if not os.path.exists(EXE):
        pytest.skip("fortran/ftte_demo_driver not built (no Fortran compiler at build time)")
    out = subprocess.run([EXE, "40", "2"], capture_output=True, text=True, timeout=300, cwd=os.path.join(ROOT, "fortran"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ftte_demo_driver OK" in out.stdout


def test_fortran_point_source_host(golden, tmp_path):
    """The Fortran star-loop host (fortran/ftte_demo_point) on the refined, dusty case the reference itself traced."""
    import numpy as np
    exe = os.path.join(ROOT, "fortran", "ftte_demo_point")
    if not os.path.exists(exe):
        pytest.skip("fortran/ftte_demo_point not built (no Fortran compiler at build time)")
    g = golden("point10_refined_dust")
    case, out = tmp_path / "case.bin", tmp_path / "rates.bin"
    with open(case, "wb") as f:
        f.write(np.array([int(g["n"]), g["level"].size, g["src_leaf"].size, int(g["dust"])], "<i4").tobytes())
        f.write(np.array([float(g["box"])], "<f8").tobytes())
        f.write(g["level"].astype("<i4").tobytes())
        for k in ("HI", "HeI", "HeII", "rho", "abun2"):
            f.write(g[k].astype("<f8").tobytes())
        f.write(g["src_leaf"].astype("<i8").tobytes())
        f.write(g["src_weight"].astype("<f8").tobytes())
        f.write(g["tables"].astype("<f8").tobytes())
    res = subprocess.run([exe, str(case), str(out)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "ftte_demo_point OK" in res.stdout, res.stdout + res.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    nc = g["level"].size
    rates = raw[:8 * 6 * nc].view("<f8").reshape(6, nc)
    highest = int(raw[8 * 6 * nc:].view("<i4")[0])
    assert highest == int(g["highestPixelLevel"])
    ref = g["krate"]
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert np.all(np.abs(rates - ref) <= 1e-9 * np.abs(ref) + 1e-13 * scale)
