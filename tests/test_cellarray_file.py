"""The cell-array `.dat` reader/writer (radiativetransfer_amd/cellarray.py) against a file written by the
reference's own converter statements (tests/golden/make_golden_dat.py)."""
import numpy as np
import pytest

from radiativetransfer_amd import cellarray


def test_reads_the_reference_file_and_writes_it_back_byte_for_byte(golden, tmp_path):
    g = golden("cellarray_dat")
    ref = tmp_path / "reference.dat"
    g["dat_bytes"].tofile(ref)
    rec = cellarray.read_dat(str(ref))
    assert np.array_equal(rec["level"], g["level"])
    for name in ("HI", "HeI", "HeII", "temperature", "density"):
        assert np.array_equal(rec[name], g[name])
    # cell centres as the reference computes them, to the bit
    centres = cellarray.cell_centres(int(g["n"]), g["level"], float(g["box"]))
    assert np.array_equal(centres[0], rec["x"]) and np.array_equal(centres[1], rec["y"]) and np.array_equal(centres[2], rec["z"])
    mine = tmp_path / "mine.dat"
    cellarray.write_dat(str(mine), g["level"], centres, g["HI"], g["HeI"], g["HeII"], g["temperature"], g["density"])
    assert np.array_equal(np.fromfile(mine, dtype=np.uint8), g["dat_bytes"])
    assert cellarray.base_grid_size(g["level"]) == int(g["n"])


def test_refuses_damaged_files(golden, tmp_path):
    g = golden("cellarray_dat")
    raw = g["dat_bytes"].copy()
    p = tmp_path / "short.dat"
    raw[:-10].tofile(p)
    with pytest.raises(ValueError):
        cellarray.read_dat(str(p))
    raw2 = raw.copy()
    raw2[0] ^= 1
    raw2.tofile(p)
    with pytest.raises(ValueError):
        cellarray.read_dat(str(p))
    with pytest.raises(ValueError):
        cellarray.cell_centres(2, np.array([0, 0, 0], np.int32), 1.0)          # too few leaves
    with pytest.raises(ValueError):
        cellarray.cell_centres(1, np.array([1] * 7, np.int32), 1.0)            # a refined cell has eight children
    with pytest.raises(ValueError):
        cellarray.base_grid_size(np.array([0, 0, 0], np.int32))
