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
