"""ctypes binding of libftte.so (include/ftte.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libftte.so")

STATUS = {
    0: "FTTE_OK", -1: "FTTE_ERR_ARG", -2: "FTTE_ERR_STATE", -3: "FTTE_ERR_NO_DEVICE", -4: "FTTE_ERR_UNSUPPORTED",
    -5: "FTTE_ERR_NOT_CUBIC", -6: "FTTE_ERR_LEVELS", -7: "FTTE_ERR_PHI", -8: "FTTE_ERR_THETA",
    -9: "FTTE_ERR_DOMINANT_AXIS", -10: "FTTE_ERR_PATTERN", -11: "FTTE_ERR_IZONE", -12: "FTTE_ERR_PIXEL", -13: "FTTE_ERR_RATES", -14: "FTTE_ERR_MEMORY", -15: "FTTE_ERR_STALLED",
}


class FtteError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{STATUS.get(code, code)}: {message}")
        self.code = code
        self.status = STATUS.get(code, str(code))


class Pattern(C.Structure):
    """ftte_pattern == patternType of definitionsModule.f90:141-152 (without the tree link)."""
    _fields_ = [(n, C.c_double) for n in
                ("xy_x0", "xy_y0", "xy_len", "xz_x0", "xz_z0", "xz_len", "yz_y0", "yz_z0", "yz_len")] + \
               [(n, C.c_int32) for n in ("xz_active", "yz_active", "xy_top", "xz_top", "yz_top", "reserved_")]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

# name -> (restype, argtypes): every symbol include/ftte.h declares
SIGNATURES = {
    "ftte_create": (C.c_int, [C.POINTER(_vp), C.c_int, _ip]),
    "ftte_destroy": (C.c_int, [_vp]),
    "ftte_last_error": (C.c_char_p, [_vp]),
    "ftte_multi_info": (C.c_char_p, [_vp]),
    "ftte_set_grid": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int32), C.c_double]),
    "ftte_set_opacity": (C.c_int, [_vp, C.c_int, _dp]),
    "ftte_set_opacity_device": (C.c_int, [_vp, C.c_int, _vp]),
    "ftte_set_species": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp]),
    "ftte_set_emissivity": (C.c_int, [_vp, _dp]),
    "ftte_set_emissivity_device": (C.c_int, [_vp, _vp]),
    "ftte_set_source_function": (C.c_int, [_vp, _dp]),
    "ftte_set_source_function_device": (C.c_int, [_vp, _vp]),
    "ftte_diffuse_sweep": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "ftte_diffuse_iteration": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "ftte_diffuse_sweep_device": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, _vp, _vp]),
    "ftte_stellar_beta_table": (C.c_int, [_vp, _dp, C.c_int, _dp, C.c_int, C.c_int, _dp, C.c_int, C.c_double, C.c_int,
                                          C.c_double, _dp]),
    "ftte_set_rate_tables": (C.c_int, [_vp, _dp]),
    "ftte_get_rate_tables": (C.c_int, [_vp, _dp]),
    "ftte_get_rates_hydrogen_helium": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp]),
    "ftte_set_medium": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, C.c_int]),
    "ftte_set_medium_device": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "ftte_set_zero_rates": (C.c_int, [_vp]),
    "ftte_locate_cell": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "ftte_point_sources": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64), _dp, _ip]),
    "ftte_point_escape": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "ftte_set_output_sigma": (C.c_int, [_vp, _dp]),
    "ftte_get_point_rates": (C.c_int, [_vp, _dp]),
    "ftte_point_rates_device": (C.c_int, [_vp, C.POINTER(_vp)]),
    "ftte_set_point_rates": (C.c_int, [_vp, _dp]),
    "ftte_set_rate_coefficients": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp]),
    "ftte_set_temperature": (C.c_int, [_vp, _dp]),
    "ftte_solve_rate_equations": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, C.c_double, C.c_int, _dp]),
    "ftte_solve_rate_equations_device": (C.c_int, [_vp, C.c_int, _vp, _dp, _dp, C.c_double, C.c_int, _dp]),
    "ftte_get_medium": (C.c_int, [_vp, _dp, _dp, _dp]),
    "ftte_compute_opacities": (C.c_int, [_vp, C.c_int, _dp]),
    "ftte_assign_uvb_radiation": (C.c_int, [_vp, C.c_int, _dp, C.c_double, _dp]),
    "ftte_assign_uvb_radiation_device": (C.c_int, [_vp, C.c_int, _dp, C.c_double, _vp]),
    "ftte_rate_equation_steps": (C.c_longlong, [_vp]),
    "ftte_point_ray_steps": (C.c_longlong, [_vp]),
    "ftte_rmax": (C.c_int, [_dp]),
    "ftte_uvb_beta_table": (C.c_int, [C.c_int, C.c_double, _dp, _dp, _dp, _dp]),
    "ftte_coll_rates": (C.c_int, [C.c_double, C.c_int, _dp]),
    "ftte_rate_coefficient_tables": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_int, _dp, _dp, _dp, _dp]),
    "ftte_uniform_table": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp]),
    "ftte_dust_cross_section": (C.c_double, [C.c_double, _dp]),
    "ftte_ingest_levels": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "ftte_cellarray_info": (C.c_int, [_vp, _ip, C.POINTER(C.c_int64), _dp, _ip, _ip]),
    "ftte_cellarray_fields": (C.c_int, [_vp, C.POINTER(C.c_int32), _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "ftte_cellarray_free": (None, [_vp]),
    "ftte_host_register": (C.c_int, [_vp, _vp, C.c_size_t]),
    "ftte_host_unregister": (C.c_int, [_vp, _vp]),
    "ftte_counter": (C.c_longlong, [_vp, C.c_char_p]),
    "ftte_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "ftte_launch_count": (C.c_int, [_vp]),
    "ftte_launch_info": (C.c_int, [_vp, C.c_int, _dp, C.POINTER(C.c_int64)]),
    "ftte_rotate_indices": (C.c_int, [C.c_int] * 7 + [_ip, _ip, _ip]),
    "ftte_pix2ang_nest": (C.c_int, [C.c_int, C.c_int64, _dp, _dp]),
    "ftte_fold_direction": (C.c_int, [C.c_double, C.c_double, _dp, _dp, _ip]),
    "ftte_set_pattern": (C.c_int, [C.POINTER(Pattern), C.c_double, C.c_double]),
    "ftte_layer_patterns": (C.c_int, [C.c_int, C.c_double, C.c_double, C.POINTER(Pattern)]),
    "ftte_compute_cell_intensity": (None, [_dp, C.c_double, C.c_double]),
}

_lib = None


def load() -> C.CDLL:
    """Load libftte.so; raises if it has not been built (python -m radiativetransfer_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m radiativetransfer_amd.build`); there is no CPU fallback")
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 and resolve it by file
        # name, so if the system copy were mapped first (through libftte.so) a later `import torch` would map a second
        # runtime and find no devices.  Mapping torch's first makes libftte.so bind to that same instance by soname.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
