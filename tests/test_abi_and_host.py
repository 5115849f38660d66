"""CPU-side checks of the product: the C-ABI library loads and exports exactly what include/ftte.h declares,
its host geometry reproduces the reference's vectors, and it fails loudly (never falls back) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rt():
    import radiativetransfer_amd
    from radiativetransfer_amd import build
    build.build_library()  # hipcc cross-compiles without a GPU
    return radiativetransfer_amd


def header_functions():
    text = open(os.path.join(ROOT, "include", "ftte.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ftte_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rt):
    from radiativetransfer_amd import _lib
    names = header_functions()
    assert len(names) >= 19
    lib = C.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f"libftte.so does not export {name}"
    # and the Python binding table covers exactly the header
    assert sorted(_lib.SIGNATURES) == names


def test_no_device_is_an_error_not_a_fallback(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.FtteError) as e:
        rt.DiffuseTransfer()
    assert e.value.status == "FTTE_ERR_NO_DEVICE"


def test_product_never_references_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "radiativetransfer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".f90")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.replace("oracle evaluated", ""), f"{f} mentions the oracle"


def test_host_geometry_matches_reference_vectors(rt, golden):
    g = golden("rotate_indices")
    nx, ny, nz = map(int, g["extents"])
    for z in range(24):
        for i in range(3):
            for j in range(4):
                for k in range(5):
                    assert rt.rotate_indices(i + 1, j + 1, k + 1, nx, ny, nz, z + 1) == tuple(g["table"][z, i, j, k])
    for name in ("geometry_192dir_16layers", "geometry_12dir_256layers"):
        g = golden(name)
        n = int(g["n"])
        for d in range(len(g["phi_in"])):
            p, t, z = rt.fold_direction(g["phi_in"][d], g["theta_in"][d])
            assert (p, t, z) == (g["phi"][d], g["theta"][d], g["izone"][d])
            L = rt.layer_patterns(n, p, t)
            for i in range(n):
                r, P = g["layers"][d][i], L[i]
                assert (P.xz_active, P.yz_active, P.xy_top, P.xz_top, P.yz_top) == tuple(r["flags"])
                assert (P.xy_x0, P.xy_y0, P.xy_len) == tuple(r["xy"])
                if P.xz_active:
                    assert (P.xz_x0, P.xz_z0, P.xz_len) == tuple(r["xz"])
                if P.yz_active:
                    assert (P.yz_y0, P.yz_z0, P.yz_len) == tuple(r["yz"])


def test_healpix_directions_match_oracle(rt):
    for level in (1, 2, 3):
        a, b = rt.healpix_directions(level), O.healpix_directions(level)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_error_codes_where_the_reference_stops(rt):
    pi = O.lib().fo_pi()
    for args, status in (((0.5 * pi, 0.3), "FTTE_ERR_PHI"), ((0.0, 0.3), "FTTE_ERR_PHI"),
                         ((0.3, 0.0), "FTTE_ERR_THETA"), ((0.3, 2.0), "FTTE_ERR_THETA"),
                         ((0.25 * np.pi, np.arctan(np.sqrt(0.5))), None)):  # near-diagonal: must fold or tie
        try:
            rt.fold_direction(*args)
            assert status is None
        except rt.FtteError as e:
            assert status is None or e.status == status
    with pytest.raises(rt.FtteError) as e:
        rt.rotate_indices(1, 1, 1, 4, 4, 4, 25)
    assert e.value.status == "FTTE_ERR_IZONE"
    with pytest.raises(rt.FtteError) as e:
        rt.pix2ang_nest(2, 48)
    assert e.value.status == "FTTE_ERR_PIXEL"


def test_compute_cell_intensity_is_the_reference_formula(rt):
    assert rt.compute_cell_intensity(1.0, 2.0, 1.0) == 1.0 + (2.0 - 1.0) / np.log(2.0)
    assert rt.compute_cell_intensity(0.0, 1.0, 1.0) == 1.0
    assert rt.compute_cell_intensity(0.0, 3e-21, 0.0) == 0.0  # Iout underflowed: (Iin-0)/log(inf) = 0


def test_shard_bounds():
    from radiativetransfer_amd.distributed import shard_bounds
    for count in (0, 1, 7, 96, 768):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_bounds(count, r, world)
                cover.extend(range(lo, hi))
                assert 0 <= hi - lo - count // world <= 1
            assert cover == list(range(count))
