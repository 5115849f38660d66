#!/usr/bin/env python3
"""Golden vectors for the grid ingest from the reference's own lines: oracle/_ref/ingest_harness runs
equiSources.f90:427-618 + placeCellProjectWithVelocity + writeCell (lifted by oracle/Makefile) on synthetic per-level cell lists.

    make -C oracle ref && python tests/golden/make_golden_ingest.py

DATA only is stored: the lists and the cell array the reference built from them.
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ingest_harness")


def synthetic_levels(n, seed, kinematics, metals, depth=3):
    """Level 1: the n^3 base cells (centres on a lattice of 7.5 kpc, jittered by less than the float32 spacing would move a cell);
    level 2: the children of some base cells, not all eight of every refined cell listed (the missing ones keep their parent's
    state); level 3: children of some level-2 cells, one of them under a level-2 cell that is itself not listed."""
    rng = np.random.default_rng(seed)
    h = 7.5
    centres = (np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3) + 0.5) * h - 0.5 * n * h
    order = rng.permutation(len(centres))            # the lists are not sorted
    levels = [dict(pos=centres[order].astype(np.float32))]
    refined = rng.choice(len(centres), size=max(3, len(centres) // 9), replace=False)
    kids, deeper = [], []
    for b in refined:
        offs = np.array([[a, c, d] for a in (-1, 1) for c in (-1, 1) for d in (-1, 1)]) * 0.25 * h
        keep = rng.random(8) < 0.8
        keep[rng.integers(8)] = True
        for q in range(8):
            p = centres[b] + offs[q]
            if keep[q]:
                kids.append(p)
            if rng.random() < 0.25:                   # grandchildren, listed or not their parent is
                o2 = np.array([[a, c, d] for a in (-1, 1) for c in (-1, 1) for d in (-1, 1)]) * 0.125 * h
                for g in range(8):
                    if rng.random() < 0.7:
                        deeper.append(p + o2[g])
    levels.append(dict(pos=np.array(kids, np.float32)))
    if depth >= 3:
        levels.append(dict(pos=np.array(deeper, np.float32)))
    for L, lv in enumerate(levels):
        nc = len(lv["pos"])
        lv["lT"] = rng.uniform(3.5, 5.5, nc).astype(np.float32)
        lv["lnH"] = (rng.normal(-3.0 + 0.6 * L, 0.5, nc)).astype(np.float32)
        lv["lx"] = rng.uniform(-5.0, -0.01, nc).astype(np.float32)
        lv["vel"] = rng.normal(0, 80.0, (nc, 3)).astype(np.float32) if kinematics else None
        lv["abun"] = (10 ** rng.uniform(-4, -1.5, (nc, 4))).astype(np.float32) if metals else None
    return levels


def run_reference(levels):
    kin, met = levels[0]["vel"] is not None, levels[0]["abun"] is not None
    with tempfile.TemporaryDirectory() as tmp:
        case, out = os.path.join(tmp, "case.bin"), os.path.join(tmp, "out.bin")
        with open(case, "wb") as f:
            f.write(struct.pack("<3i", len(levels), int(kin), int(met)))
            for lv in levels:
                f.write(struct.pack("<i", len(lv["pos"])))
                f.write(np.asfortranarray(lv["pos"]).tobytes(order="F"))
                for k in ("lT", "lnH", "lx"):
                    f.write(lv[k].tobytes())
                if kin:
                    f.write(np.asfortranarray(lv["vel"]).tobytes(order="F"))
                if met:
                    f.write(np.asfortranarray(lv["abun"]).tobytes(order="F"))
        res = subprocess.run([HARNESS, case, out], capture_output=True, text=True)
        if res.returncode != 0 or not os.path.exists(out):
            raise RuntimeError(f"ingest_harness failed: {res.stdout[-400:]} {res.stderr[-400:]}")
        raw = open(out, "rb").read()
    off = 0

    def take(count, dtype):
        nonlocal off
        a = np.frombuffer(raw, dtype, count, off).copy()
        off += count * np.dtype(dtype).itemsize
        return a
    nx, ncell = take(2, "<i4")
    o = {"n": int(nx), "box": float(take(1, "<f8")[0]), "level": take(ncell, "<i4")}
    for k in ("HI", "HeI", "HeII", "temperature", "density"):
        o["f32_" + k] = take(ncell, "<f4")
    if kin:
        for k in ("velx", "vely", "velz"):
            o["f32_" + k] = take(ncell, "<f4")
    if met:
        o["f32_abun2"] = take(ncell, "<f4")
    f64 = take(9 * ncell, "<f8").reshape(9, ncell)   # Fortran f64(ncell,9)
    for q, k in enumerate(("HI", "HeI", "HeII", "tgas", "rho", "velx", "vely", "velz", "abun2")):
        o[k] = f64[q]
    assert off == len(raw)
    return o


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build oracle/_ref/ingest_harness first: make -C oracle ref")
    for name, n, seed, kin, met in (("ingest6_three_levels_metals_velocities", 6, 3, True, True), ("ingest5_metals", 5, 11, False, True)):
        levels = synthetic_levels(n, seed, kin, met)
        o = run_reference(levels)
        store = {}
        for L, lv in enumerate(levels):
            for k, a in lv.items():
                if a is not None:
                    store[f"in{L + 1}_{k}"] = a
        store.update({"out_" + k: v for k, v in o.items()})
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, nlevels=len(levels), **store)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, n = {o['n']}, {len(o['level'])} leaves, levels {np.bincount(o['level']).tolist()}, "
              f"box {o['box']:.4e} cm")


if __name__ == "__main__":
    main()
