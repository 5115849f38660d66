"""The library's host planners under AddressSanitizer and UndefinedBehaviorSanitizer (host code only: GPU sanitizers are not
available): tests/host_sanitize_harness.cpp drives the tree rebuild, the per-direction segment forests -- whole tree and
restricted to a box with brick face buffers around it --, the ray geometry and the grid ingest over random refined cell arrays
and checks every index they hand to the device kernels."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "radiativetransfer_amd", "csrc")


def test_host_planners_under_asan_and_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "harness")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_sanitize_harness.cpp"),
           os.path.join(CSRC, "ftte_amr.cpp"), os.path.join(CSRC, "ftte_geometry.cpp"), os.path.join(CSRC, "ftte_ingest.cpp"),
           "-lpthread", "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and ("asan" in build.stderr or "ubsan" in build.stderr) and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtimes not installed")
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-4000:]
    assert "host planners under the sanitizers" in run.stdout and "ERROR" not in run.stderr
