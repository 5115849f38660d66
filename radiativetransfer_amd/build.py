"""Build radiativetransfer_amd/libftte.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m radiativetransfer_amd.build [--force]

hipcc cross-compiles without a GPU.  The shared object is git-ignored but travels to the GPU box
with the repository snapshot; the Python side never builds at import time and never falls back to
anything else if the library is missing.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libftte.so")
SOURCES = ["ftte_kernels.hip", "ftte_brick.hip", "ftte_api.cpp", "ftte_plan.cpp", "ftte_sweeps.cpp", "ftte_hybrid.cpp", "ftte_multi.cpp", "ftte_host_arrays.cpp",
           "ftte_geometry.cpp", "ftte_amr.cpp", "ftte_point.cpp", "ftte_ingest.cpp"]
HEADERS = ["ftte.map", "ftte_context.h", "ftte_internal.h", "ftte_kernels.h", "ftte_geometry.h", "ftte_math.h", "ftte_amr.h", "ftte_point.h", os.path.join("..", "..", "include", "ftte.h")]
# -ffp-contract=off: the sweep arithmetic spells out its fused multiply-adds (ftte_math.h); nothing else may be fused,
# so that the device rounds exactly like the host evaluation the parity tests compare against.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-rdc", "-Wall"]
FLAGS += os.environ.get("FTTE_CXXFLAGS", "").split()  # experiments only (e.g. -DFTTE_TOUCH=1)


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def includes(path: str, seen=None) -> set:
    """Quoted includes of a source file, followed recursively."""
    import re
    seen = set() if seen is None else seen
    for m in re.finditer(r'^\s*#include\s+"([^"]+)"', open(path).read(), re.M):
        h = os.path.normpath(os.path.join(os.path.dirname(path), m.group(1)))
        if h not in seen and os.path.exists(h):
            seen.add(h)
            includes(h, seen)
    return seen


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    # per-file: an object is rebuilt when its source, a header it includes, or this script is newer; files compile side by side
    from concurrent.futures import ThreadPoolExecutor
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        deps = sorted(includes(src)) + [src, os.path.abspath(__file__)]
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps):
            jobs.append([hipcc(), *FLAGS, "-x", "hip", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(run, jobs))
    # the version script keeps the C++ behind the ABI out of the dynamic symbol table: only ftte_* is exported
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "ftte.map"), "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
