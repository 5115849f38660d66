"""The reference's cell-array file without HDF4.

`hdf42bin.f90:208-218` writes the cell array as a Fortran sequential unformatted file `<name>.dat` of nine records,

    level (int32) | x | y | z | HI | HeI | HeII | temperature | density      (float32, ncell values each)

each framed by 4-byte length markers; the positions are the cell centres in kpc, box centred on the origin
(`hdf42bin.f90:162-192`, `computeCellCoordinates:222-269`).  The HDF4 files of the main program hold the same arrays in the
same order as datasets 1..6 (`equiSources.f90:4738-4795`, `readCellArray.f90:18-98`), preceded by the base grid size; HDF4
itself is not available here.  This module reads and writes the `.dat` form and computes the centres; it is host-side I/O
around the device path (`DiffuseTransfer.set_grid` takes `level`, `StellarTransfer.set_medium` the species).
"""
from __future__ import annotations

import struct
from typing import Dict

import numpy as np

KPC = float(np.float32(1.0e3)) * float(np.float32(3.08568025e18))  # kpc = 1.e3*pc, default-real literals widened
FIELDS = ("level", "x", "y", "z", "HI", "HeI", "HeII", "temperature", "density")


def cell_centres(n: int, level, physical_box_size_cm: float) -> np.ndarray:
    """float32 [3][ncell]: cell centres in kpc as hdf42bin writes them.  The arithmetic is the reference's: base centres
    (float(i)-0.5)/float(nx) in single precision, children at +-0.25 of the parent's size in double, the result narrowed to
    float32, then scaled to [-L/2, L/2] kpc in double and narrowed again."""
    level = np.asarray(level, dtype=np.int64)
    ncell = level.size
    out = np.empty((3, ncell), dtype=np.float32)
    xa = -0.5 * physical_box_size_cm / KPC
    xb = 0.5 * physical_box_size_cm / KPC
    cursor = 0

    def place(x0, y0, z0, lvl, size):
        nonlocal cursor
        if cursor >= ncell:
            raise ValueError("error in levels")
        if level[cursor] == lvl:
            for q, v in enumerate((x0, y0, z0)):
                unit = np.float32(v)                              # cellArrayXpos(icosmic) = x0
                out[q, cursor] = np.float32(float(unit) * (xb - xa) + xa)
            cursor += 1
        elif level[cursor] > lvl:
            for a in (-1, 1):
                for b in (-1, 1):
                    for c in (-1, 1):
                        place(x0 + a * 0.25 * size, y0 + b * 0.25 * size, z0 + c * 0.25 * size, lvl + 1, size / 2.0)
        else:
            raise ValueError("error in levels")

    fn = np.float32(n)
    for i in range(1, n + 1):
        xpos = float((np.float32(i) - np.float32(0.5)) / fn)
        for j in range(1, n + 1):
            ypos = float((np.float32(j) - np.float32(0.5)) / fn)
            for k in range(1, n + 1):
                zpos = float((np.float32(k) - np.float32(0.5)) / fn)
                place(xpos, ypos, zpos, 0, 1.0 / float(n))
    if cursor != ncell:
        raise ValueError("error in levels")
    return out


def write_dat(path: str, level, centres_kpc, HI, HeI, HeII, temperature, density) -> None:
    """Write `<name>.dat` exactly as hdf42bin.f90:208-218 does (little-endian, 4-byte record markers)."""
    level = np.ascontiguousarray(level, dtype="<i4")
    ncell = level.size
    recs = [level] + [np.ascontiguousarray(a, dtype="<f4").reshape(-1) for a in
                      (centres_kpc[0], centres_kpc[1], centres_kpc[2], HI, HeI, HeII, temperature, density)]
    for r in recs:
        if r.size != ncell:
            raise ValueError("all records have ncell entries")
    if 4 * ncell >= 2 ** 31:
        raise ValueError("records beyond 2 GiB need sub-records, which this writer does not produce")
    with open(path, "wb") as f:
        for r in recs:
            mark = struct.pack("<i", 4 * ncell)
            f.write(mark)
            f.write(r.tobytes())
            f.write(mark)


def read_dat(path: str) -> Dict[str, np.ndarray]:
    """Read a cell-array `.dat` file: dict with the nine records under FIELDS' names."""
    raw = np.fromfile(path, dtype=np.uint8)
    out, off = {}, 0
    for name in FIELDS:
        if off + 4 > raw.size:
            raise ValueError(f"{path}: file ends before record '{name}'")
        nbytes = int(raw[off:off + 4].view("<i4")[0])
        if nbytes < 0 or off + 8 + nbytes > raw.size or int(raw[off + 4 + nbytes:off + 8 + nbytes].view("<i4")[0]) != nbytes:
            raise ValueError(f"{path}: bad record markers around '{name}'")
        body = raw[off + 4:off + 4 + nbytes]
        out[name] = body.view("<i4" if name == "level" else "<f4").copy()
        off += 8 + nbytes
    if off != raw.size:
        raise ValueError(f"{path}: {raw.size - off} bytes after the ninth record")
    n0 = out["level"].size
    if any(v.size != n0 for v in out.values()):
        raise ValueError(f"{path}: records of different lengths")
    return out


def base_grid_size(level) -> int:
    """n of the n^3 base grid that a depth-first level list describes: a leaf at level l covers 8^-l of a base cell."""
    level = np.asarray(level, dtype=np.int64)
    cells = float(np.sum(8.0 ** (-level)))
    n = round(cells ** (1.0 / 3.0))
    if n < 1 or abs(n ** 3 - cells) > 1e-6 * max(cells, 1.0):
        raise ValueError("the level list does not fill a cubic base grid")
    return int(n)
