"""radiativetransfer_amd -- MI355X-native diffuse radiative-transfer sweep (FTTE hot path).

The package is a thin host layer over libftte.so (HIP kernels for gfx950 + the C ABI of
include/ftte.h).  Importing it never compiles anything and never substitutes a CPU path.
"""
from . import cellarray, ingest, synthetic  # noqa: F401
from .api import (DiffuseTransfer, StellarTransfer, FtteError, Pattern, compute_cell_intensity, fold_direction,  # noqa: F401
                  healpix_directions, layer_patterns, pix2ang_nest, rotate_indices, set_pattern, rmax, dust_cross_section, uvb_beta_table, uniform_table, coll_rates, rate_coefficient_tables)
