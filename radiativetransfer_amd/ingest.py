"""Grid ingest: per-level lists of SPH-projected cells -> the cell array.

The reference reads these lists from an HDF4 file (`equiSources.f90:316-423`: per level the datasets position (ncell,3) [kpc],
log10 temperature, log10 n_H, log10 neutral fraction, optionally abundances (ncell,4) and velocities (ncell,3)), builds its
octree from them (`:427-618`, `placeCellProjectWithVelocity:1870-1974`) and keeps it for the whole run.  `ingest_levels` is the
host-only entry point of the library (`ftte_ingest_levels`, csrc/ftte_ingest.cpp) for that step; HDF4 itself is not available
here, the lists are handed over as arrays.  The result feeds `DiffuseTransfer.set_grid` (level), `StellarTransfer.set_medium`
(HI, HeI, HeII, rho, abun2), `set_temperature` (tgas) and `cellarray.write_dat`.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib


class _LevelList(C.Structure):
    _fields_ = [("ncell", C.c_int64)] + [(n, C.POINTER(C.c_float)) for n in ("pos", "lT", "lnH", "lx", "vel", "abun")]


def ingest_levels(levels: Sequence[Dict[str, Optional[np.ndarray]]]) -> Dict[str, object]:
    """levels[L-1] = dict(pos=(ncell,3) kpc, lT, lnH, lx (ncell,), vel=(ncell,3) or None, abun=(ncell,4) or None), float32.
    Returns dict(n, box [cm], level int32[ncell], HI, HeI, HeII, tgas, rho, velx, vely, velz, abun2 float64[ncell])."""
    lib = _lib.load()
    fp = C.POINTER(C.c_float)
    keep, recs = [], (_LevelList * len(levels))()
    for rec, lv in zip(recs, levels):
        pos = np.asfortranarray(np.asarray(lv["pos"], dtype=np.float32))   # Fortran (ncell,3): all x, all y, all z
        if pos.ndim != 2 or pos.shape[1] != 3:
            raise ValueError("pos must have shape (ncell, 3)")
        nc = pos.shape[0]
        rec.ncell = nc
        arrays = {"pos": pos}
        for k in ("lT", "lnH", "lx"):
            a = np.ascontiguousarray(lv[k], dtype=np.float32)
            if a.shape != (nc,):
                raise ValueError(f"{k} must have shape (ncell,)")
            arrays[k] = a
        for k, width in (("vel", 3), ("abun", 4)):
            a = lv.get(k)
            if a is not None:
                a = np.asfortranarray(np.asarray(a, dtype=np.float32))
                if a.shape != (nc, width):
                    raise ValueError(f"{k} must have shape (ncell, {width})")
            arrays[k] = a
        for k, a in arrays.items():
            setattr(rec, k, a.ctypes.data_as(fp) if a is not None else None)
        keep.append(arrays)
    handle = C.c_void_p()
    lib.ftte_ingest_levels.restype = C.c_int
    lib.ftte_ingest_levels.argtypes = [C.c_int, C.POINTER(_LevelList), C.POINTER(C.c_void_p)]
    code = lib.ftte_ingest_levels(len(levels), recs, C.byref(handle))
    if code:
        raise _lib.FtteError(code, "ftte_ingest_levels: " + {-5: "level 1 does not fill an n^3 base grid (equiSources.f90:436-439)",
                                                            -1: "bad argument, or a cell outside the level-1 bounding box"}.get(code, ""))
    try:
        nx, ncell, box = C.c_int(), C.c_int64(), C.c_double()
        kin, met = C.c_int(), C.c_int()
        lib.ftte_cellarray_info(handle, C.byref(nx), C.byref(ncell), C.byref(box), C.byref(kin), C.byref(met))
        out: Dict[str, object] = {"n": nx.value, "box": box.value, "has_velocity": bool(kin.value), "has_metals": bool(met.value)}
        level = np.empty(ncell.value, np.int32)
        names = ("HI", "HeI", "HeII", "tgas", "rho", "velx", "vely", "velz", "abun2")
        fields = [np.empty(ncell.value) for _ in names]
        dp = C.POINTER(C.c_double)
        lib.ftte_cellarray_fields(handle, level.ctypes.data_as(C.POINTER(C.c_int32)), *[f.ctypes.data_as(dp) for f in fields])
        out["level"] = level
        out.update(dict(zip(names, fields)))
        return out
    finally:
        lib.ftte_cellarray_free(handle)
