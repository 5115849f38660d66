"""Grid ingest: ftte_ingest_levels (csrc/ftte_ingest.cpp, host only) against the cell arrays the reference's own
lines built from the same per-level lists (tests/golden/ingest*.npz: equiSources.f90:427-618 + placeCellProjectWithVelocity +
writeCell, lifted into oracle/_ref/ingest_harness), then on through the .dat file and -- on the GPU -- into the library."""
import numpy as np
import pytest

import radiativetransfer_amd as rt
from radiativetransfer_amd import cellarray, ingest

NAMES = ["ingest6_three_levels_metals_velocities", "ingest5_metals"]
FIELDS = ("HI", "HeI", "HeII", "tgas", "rho", "velx", "vely", "velz", "abun2")


def lists_of(g):
    out = []
    for L in range(1, int(g["nlevels"]) + 1):
        out.append({k: (g[f"in{L}_{k}"] if f"in{L}_{k}" in g.files else None) for k in ("pos", "lT", "lnH", "lx", "vel", "abun")})
    return out


@pytest.mark.parametrize("name", NAMES)
def test_ingest_equals_the_reference_tree(golden, name):
    g = golden(name)
    a = ingest.ingest_levels(lists_of(g))
    assert a["n"] == int(g["out_n"]) and a["box"] == float(g["out_box"])      # physicalBoxSize, bit for bit
    assert np.array_equal(a["level"], g["out_level"])                          # the same tree, the same leaf order
    # 10.**x is real*4 ** real*4 in the reference: its compiler's powf and this one's may differ in the last bit of the float
    for k in FIELDS:
        want = g["out_" + k]
        if k in ("velx", "vely", "velz", "abun2"):
            assert np.array_equal(a[k], want), k                               # copied or smoothed in exact arithmetic
        else:
            assert np.allclose(a[k], want, rtol=2.0 ** -22, atol=0), k
    # and what writeCell stores (sngl)
    for mine, ref in (("HI", "HI"), ("HeI", "HeI"), ("HeII", "HeII"), ("tgas", "temperature"), ("rho", "density")):
        assert np.allclose(a[mine].astype(np.float32), g["out_f32_" + ref], rtol=2.0 ** -21, atol=0)
    assert np.array_equal(a["abun2"].astype(np.float32), g["out_f32_abun2"])
    exact = sum(np.array_equal(a[k], g["out_" + k]) for k in FIELDS)
    assert exact >= 4


def test_ingest_properties(golden):
    g = golden(NAMES[0])
    lists = lists_of(g)
    a = ingest.ingest_levels(lists)
    level = a["level"]
    # every listed cell of the deepest level became a leaf of that depth (no two of them share one here)
    assert np.count_nonzero(level == len(lists) - 1) >= len(lists[-1]["lT"])
    # a refined cell's unlisted children carry their parent's state: fewer distinct temperatures than leaves
    assert len(np.unique(a["tgas"])) < len(level)
    # volume: the leaves tile the base grid
    assert np.isclose(np.sum(8.0 ** -level.astype(float)), a["n"] ** 3)
    # helium as the reference sets it: HeI = (1 - psi) rho / mhe, HeII = 0
    assert not a["HeII"].any() and np.all(a["HeI"] > 0)
    # without metals a placed cell gets abun2 = 0.02 (the real*4 literal); without velocities zeros; a list that is no cube is refused
    plain = [{k: (v if k not in ("vel", "abun") else None) for k, v in lv.items()} for lv in lists]
    b = ingest.ingest_levels(plain)
    # (children nobody lists keep the 0 they were created with, :1904)
    assert np.array_equal(b["level"], level) and set(np.unique(b["abun2"])) <= {0.0, float(np.float32(0.02))} and not b["velx"].any()
    assert np.count_nonzero(b["abun2"]) > 0.6 * len(level)
    bad = [dict(lists[0])]
    for k in ("pos", "lT", "lnH", "lx", "vel", "abun"):
        bad[0][k] = bad[0][k][:-1]
    with pytest.raises(rt.FtteError) as err:
        ingest.ingest_levels(bad)
    assert err.value.status == "FTTE_ERR_NOT_CUBIC"


def test_ingest_to_dat_round_trip(golden, tmp_path):
    """lists -> cell array -> the reference's .dat form (row F3) -> back: the level list and the float32 fields survive."""
    g = golden(NAMES[1])
    a = ingest.ingest_levels(lists_of(g))
    path = tmp_path / "cells.dat"
    centres = cellarray.cell_centres(a["n"], a["level"], a["box"])
    cellarray.write_dat(str(path), a["level"], centres, a["HI"], a["HeI"], a["HeII"], a["tgas"], a["rho"])
    back = cellarray.read_dat(str(path))
    assert np.array_equal(back["level"], a["level"]) and cellarray.base_grid_size(back["level"]) == a["n"]
    for mine, theirs in (("HI", "HI"), ("HeI", "HeI"), ("HeII", "HeII"), ("tgas", "temperature"), ("rho", "density")):
        assert np.array_equal(back[theirs], a[mine].astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_ingested_cell_array_drives_the_library(golden, name):
    """The round trip the verdict asks for: per-level lists -> ingest -> ftte_set_grid / ftte_set_medium -> opacities -> sweep,
    equal to the oracle's tree sweep on the reference-built cell array of the golden file."""
    import _oracle as O
    g = golden(name)
    a = ingest.ingest_levels(lists_of(g))
    phi, theta, w = O.healpix_directions(1)
    beta = np.array([[6.3e-18, 1e-18, 2e-19], [0.0, 7.4e-18, 1e-18], [0.0, 0.0, 1.6e-18]])
    uvb = np.array([1e-21, 3e-22, 1e-22])
    with rt.StellarTransfer() as st:
        st.set_grid(a["n"], a["level"], a["box"])
        st.set_medium(a["HI"], a["HeI"], a["HeII"], a["rho"], a["abun2"], 0)
        st.compute_opacities_from_medium(beta)
        J = st.transport(phi, theta, w, uvb)
    kappa = O.compute_opacities(g["out_HI"], g["out_HeI"], g["out_HeII"], beta)
    ref = O.sweep_tree(int(g["out_n"]), g["out_level"], kappa, float(g["out_box"]), phi, theta, w, uvb, arith=O.ARITH_DEVICE)
    # the ingested fields are the reference's own, bit for bit (test_ingest_equals_the_reference_tree checks all nine): so are
    # the opacities, and J agrees to the rounding of the sum over directions
    for mine, theirs in (("HI", "out_HI"), ("HeI", "out_HeI"), ("HeII", "out_HeII")):
        assert np.array_equal(np.asarray(a[mine], dtype=np.float64), np.asarray(g[theirs], dtype=np.float64)), mine
    assert np.array_equal(a["level"], g["out_level"])
    assert J.shape == ref.shape
    assert np.allclose(J, ref, rtol=64 * np.finfo(np.float64).eps, atol=0)
