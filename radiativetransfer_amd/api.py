"""Host-side mirror of the reference's diffuse-transfer interface, on top of the C ABI.

The reference has no API object: its driver (equiSources.f90:1372-1808) calls
`computeOpacities`, then per direction `pix2ang_nest`, the folding block, `setPattern`,
`localizeCellFindNeighbours` and `transport`.  The names below follow that vocabulary.
Everything that touches cells runs in libftte.so on the GPU; there is no Python or CPU
implementation of the sweep in this package.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import FtteError, Pattern

__all__ = ["DiffuseTransfer", "StellarTransfer", "rmax", "dust_cross_section", "uvb_beta_table", "uniform_table", "coll_rates", "rate_coefficient_tables", "FtteError", "Pattern", "pix2ang_nest", "healpix_directions", "fold_direction",
           "rotate_indices", "set_pattern", "layer_patterns", "compute_cell_intensity"]


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _check(code: int, what: str):
    if code:
        raise FtteError(code, what)


# ---- host geometry: the reference's callable surface ----------------------------------------------------

def rotate_indices(i: int, j: int, k: int, nx: int, ny: int, nz: int, izone: int) -> Tuple[int, int, int]:
    """rotateIndices (rotateIndicesModule.f90:7): sweep indices -> storage indices, 1-based."""
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    _check(_lib.load().ftte_rotate_indices(i, j, k, nx, ny, nz, izone, C.byref(a), C.byref(b), C.byref(c)),
           f"izone {izone} outside 1..24")
    return a.value, b.value, c.value


def pix2ang_nest(nside: int, ipix: int) -> Tuple[float, float]:
    """pix2ang_nest + rotateAngles (equiSources.f90:2118, 2297): rotated centre of a NESTED pixel."""
    p, t = C.c_double(), C.c_double()
    _check(_lib.load().ftte_pix2ang_nest(nside, ipix, C.byref(p), C.byref(t)), "nside/ipix out of range")
    return p.value, t.value


def healpix_directions(angular_level: int, count: Optional[int] = None):
    """(phi, theta, weight) of the first `count` pixels of angular level L (12*4^(L-1) pixels),
    equal weights 1/count: the direction loop header of equiSources.f90:1385-1391."""
    nside = 2 ** (angular_level - 1)
    npix = 12 * nside * nside
    count = npix if count is None else count
    if not 0 < count <= npix:
        raise ValueError("count out of range")
    ang = np.array([pix2ang_nest(nside, i) for i in range(count)])
    return ang[:, 0].copy(), ang[:, 1].copy(), np.full(count, 1.0 / count)


def fold_direction(phi: float, theta: float) -> Tuple[float, float, int]:
    """The folding block equiSources.f90:1395-1454: canonical (phi, theta) and izone."""
    p, t, z = C.c_double(), C.c_double(), C.c_int()
    _check(_lib.load().ftte_fold_direction(phi, theta, C.byref(p), C.byref(t), C.byref(z)),
           f"direction ({phi}, {theta}) cannot be folded")
    return p.value, t.value, z.value


def set_pattern(x0: float, y0: float, phi: float, theta: float) -> Pattern:
    """setPattern (transportRoutinesModule.f90:7) for a ray entering the unit cell at (x0, y0)."""
    p = Pattern()
    p.xy_x0, p.xy_y0 = x0, y0
    _check(_lib.load().ftte_set_pattern(C.byref(p), phi, theta), "ray pattern left the unit cell")
    return p


def layer_patterns(n: int, phi: float, theta: float):
    """Patterns of layers 1..n of a folded direction (equiSources.f90:1495-1534)."""
    arr = (Pattern * n)()
    _check(_lib.load().ftte_layer_patterns(n, phi, theta, arr), "ray pattern left the unit cell")
    return arr


def compute_cell_intensity(jmean: float, i_in: float, i_out: float) -> float:
    """computeCellIntensity (transportRoutinesModule.f90:1036): returns the updated Jmean."""
    j = C.c_double(jmean)
    _lib.load().ftte_compute_cell_intensity(C.byref(j), i_in, i_out)
    return j.value


# ---- the sweep ---------------------------------------------------------------------------------------------

class DiffuseTransfer:
    """One GPU's diffuse-transfer engine: the `if (runUVBTransfer)` block of the reference driver.

        rt = DiffuseTransfer(device=0)
        rt.set_grid(n, level, box_cm)            # the cell array (definitionsModule.f90:323-326)
        rt.compute_opacities(HI, HeI, HeII, beta)  # or rt.set_opacity(kappa)
        J = rt.transport(phi, theta, weight, uvb)  # J[nnu][ncell], cell-array order
    """

    def __init__(self, device: Optional[int] = None, devices: Optional[Sequence[int]] = None):
        """device: one HIP ordinal (default: the current device); devices: several ordinals -> ONE context that splits the sweep
        over them (frequency groups first, then directions) and takes host arrays only (include/ftte.h: ftte_create)."""
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        if devices is not None and len(devices) > 1:
            ids = (C.c_int * len(devices))(*devices)
            code = self._lib.ftte_create(C.byref(self._ctx), len(devices), ids)
        else:
            if devices is not None:
                device = devices[0]
            ids = (C.c_int * 1)(device) if device is not None else None
            code = self._lib.ftte_create(C.byref(self._ctx), 1, ids)
        if code:
            raise FtteError(code, self._lib.ftte_last_error(None).decode())
        self.n = 0
        self.ncell = 0
        self.nnu = 0

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.ftte_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ok(self, code: int):
        if code:
            raise FtteError(code, self._lib.ftte_last_error(self._ctx).decode())

    # -- inputs
    def set_grid(self, n, level: Sequence[int], box_cm: float):
        """n: base grid size (int, or (nx, ny, nz)); level: depth-first leaf levels; box_cm: physicalBoxSize."""
        nx, ny, nz = (n, n, n) if np.isscalar(n) else n
        level = np.ascontiguousarray(level, dtype=np.int32)
        self._ok(self._lib.ftte_set_grid(self._ctx, nx, ny, nz, level.size, level.ctypes.data_as(C.POINTER(C.c_int32)),
                                         float(box_cm)))
        self.n, self.ncell = int(nx), int(level.size)

    def set_uniform_grid(self, n: int, box_cm: float):
        self.set_grid(n, np.zeros(n ** 3, np.int32), box_cm)

    def set_opacity(self, kappa):
        """kappa[nnu][ncell] (numpy, host)."""
        kappa = _f64(kappa)
        if kappa.ndim != 2 or (self.ncell and kappa.shape[1] != self.ncell):
            raise ValueError("kappa must have shape [nnu][ncell]")
        self._ok(self._lib.ftte_set_opacity(self._ctx, kappa.shape[0], _dp(kappa)))
        self.nnu = kappa.shape[0]

    def set_opacity_device(self, nnu: int, device_ptr: int):
        """kappa[nnu][ncell] already in device memory (e.g. torch.Tensor.data_ptr())."""
        self._ok(self._lib.ftte_set_opacity_device(self._ctx, nnu, C.c_void_p(device_ptr)))
        self.nnu = nnu

    def compute_opacities(self, HI, HeI, HeII, beta):
        """computeOpacities (equiSources.f90:4956): beta[3][nnu], rows HI, HeI, HeII."""
        HI, HeI, HeII, beta = map(_f64, (HI, HeI, HeII, beta))
        if beta.ndim != 2 or beta.shape[0] != 3:
            raise ValueError("beta must have shape [3][nnu]")
        self._ok(self._lib.ftte_set_species(self._ctx, beta.shape[1], _dp(HI), _dp(HeI), _dp(HeII), _dp(beta)))
        self.nnu = beta.shape[1]

    def set_emissivity(self, eta=None):
        """The reference's emission term (transportRoutinesModule.f90:676) with eta[nnu][ncell]; None switches it off."""
        eta = None if eta is None else _f64(eta)
        self._ok(self._lib.ftte_set_emissivity(self._ctx, None if eta is None else _dp(eta)))

    def set_source_function(self, S=None):
        """Source function S[nnu][ncell]: Iout = Iin exp(-tau) + S (1 - exp(-tau)) (not in the reference); None: off."""
        S = None if S is None else _f64(S)
        self._ok(self._lib.ftte_set_source_function(self._ctx, None if S is None else _dp(S)))

    def set_emissivity_device(self, device_ptr: int):
        self._ok(self._lib.ftte_set_emissivity_device(self._ctx, C.c_void_p(device_ptr)))

    def set_source_function_device(self, device_ptr: int):
        self._ok(self._lib.ftte_set_source_function_device(self._ctx, C.c_void_p(device_ptr)))

    def set_option(self, key: str, value: int):
        self._ok(self._lib.ftte_set_option(self._ctx, key.encode(), int(value)))

    # -- the sweep
    def transport(self, phi, theta, weight, uvb) -> np.ndarray:
        """One diffuse-transfer iteration; returns J[nnu][ncell] (host)."""
        phi, theta, weight, uvb = map(_f64, (phi, theta, weight, uvb))
        if not (len(phi) == len(theta) == len(weight)) or (self.nnu and len(uvb) != self.nnu):
            raise ValueError("direction arrays must have equal length and uvb one value per frequency group")
        J = np.empty((max(self.nnu, 1), max(self.ncell, 1)))
        self._ok(self._lib.ftte_diffuse_sweep(self._ctx, len(phi), _dp(phi), _dp(theta), _dp(weight), _dp(uvb), _dp(J)))
        return J

    def transport_device(self, phi, theta, weight, uvb, j_device_ptr: int, stream: int = 0):
        """Same with J[nnu][ncell] in device memory; asynchronous on `stream` (a hipStream_t handle, 0 = own)."""
        phi, theta, weight, uvb = map(_f64, (phi, theta, weight, uvb))
        self._ok(self._lib.ftte_diffuse_sweep_device(self._ctx, len(phi), _dp(phi), _dp(theta), _dp(weight), _dp(uvb),
                                                     C.c_void_p(j_device_ptr), C.c_void_p(stream)))

    def transport_into(self, phi, theta, weight, uvb, J: np.ndarray) -> np.ndarray:
        """Same as transport() into a caller-owned J[nnu][ncell] (e.g. one registered with host_register)."""
        phi, theta, weight, uvb = map(_f64, (phi, theta, weight, uvb))
        if J.dtype != np.float64 or not J.flags.c_contiguous or J.shape != (self.nnu, self.ncell):
            raise ValueError("J must be a C-contiguous float64 array of shape [nnu][ncell]")
        self._ok(self._lib.ftte_diffuse_sweep(self._ctx, len(phi), _dp(phi), _dp(theta), _dp(weight), _dp(uvb), _dp(J)))
        return J

    def iterate_into(self, kappa: np.ndarray, phi, theta, weight, uvb, J: np.ndarray) -> np.ndarray:
        """set_opacity(kappa) + transport_into(..., J) as one call (ftte_diffuse_iteration): on a uniform grid the frequency
        groups cross PCIe and are swept in overlapping lanes."""
        phi, theta, weight, uvb = map(_f64, (phi, theta, weight, uvb))
        kappa = _f64(kappa)
        if kappa.ndim != 2 or kappa.shape[1] != self.ncell:
            raise ValueError("kappa must have shape [nnu][ncell]")
        if J.dtype != np.float64 or not J.flags.c_contiguous or J.shape != kappa.shape:
            raise ValueError("J must be a C-contiguous float64 array of kappa's shape")
        self._ok(self._lib.ftte_diffuse_iteration(self._ctx, kappa.shape[0], _dp(kappa), len(phi), _dp(phi), _dp(theta), _dp(weight),
                                                  _dp(uvb), _dp(J)))
        self.nnu = kappa.shape[0]
        return J

    def host_register(self, a: np.ndarray):
        """Pin a host array the caller keeps (kappa, J): the library then moves it by DMA without a staging copy."""
        self._ok(self._lib.ftte_host_register(self._ctx, C.c_void_p(a.ctypes.data), a.nbytes))

    def host_unregister(self, a: np.ndarray):
        self._ok(self._lib.ftte_host_unregister(self._ctx, C.c_void_p(a.ctypes.data)))

    def multi_info(self) -> str:
        """How the last sweep of a multi-device context combined its devices' J ("" for one device)."""
        return self._lib.ftte_multi_info(self._ctx).decode()

    def counter(self, name: str) -> int:
        """grid_builds / plan_builds / forest_builds: how often the expensive host-side builds ran."""
        return int(self._lib.ftte_counter(self._ctx, name.encode()))

    def launch_records(self):
        """[(ms, updates)] of the sweep-kernel launches of the last sweep (synchronise first)."""
        out = []
        for i in range(self._lib.ftte_launch_count(self._ctx)):
            ms, upd = C.c_double(), C.c_int64()
            self._ok(self._lib.ftte_launch_info(self._ctx, i, C.byref(ms), C.byref(upd)))
            out.append((ms.value, upd.value))
        return out


# ---- point sources ---------------------------------------------------------------------------------------

TABLE_SHAPE = (6, 11, 11, 11, 11)  # reactionRate1..3, energyRate1..3; then [tauDust][tau3][tau2][tau1]
RATE_NAMES = ("krate24", "krate25", "krate26", "crate24", "crate25", "crate26")


def rmax() -> np.ndarray:
    """rmax(1:30) of equiSources.f90:296-309: path length (cells) after which a ray of pixel level L splits."""
    out = np.empty(30)
    _check(_lib.load().ftte_rmax(_dp(out)), "ftte_rmax")
    return out


def uvb_beta_table(alpha, nfreq: int = 400, freqdel: Optional[float] = None):
    """uvbBetaTable (uvbBetaTable.f90): returns (beta[3][3] = [species HI, HeI, HeII][group], ksi[3][3] = [group][24, 25, 26],
    gamma[3][3] = [group][HI, HeI, HeII]) for the three groups' power-law slopes alpha[3].  Defaults: the reference's nfbins
    and frequencyBinWidth (the default-real literal 0.02 widened, definitionsModule.f90:239-241)."""
    alpha = _f64(alpha).reshape(3)
    beta, ksi, gamma = np.empty((3, 3)), np.empty((3, 3)), np.empty((3, 3))
    _check(_lib.load().ftte_uvb_beta_table(int(nfreq), float(np.float32(0.02)) if freqdel is None else float(freqdel), _dp(alpha),
                                           _dp(beta), _dp(ksi), _dp(gamma)), "ftte_uvb_beta_table")
    return beta, ksi, gamma


def coll_rates(T: float, recombination_type: int = 2) -> np.ndarray:
    """coll_rates (coll_rates.f): k1..k6 at temperature T; recombination_type 1 = case A, 2 = case B."""
    k = np.empty(6)
    _check(_lib.load().ftte_coll_rates(float(T), int(recombination_type), _dp(k)), "ftte_coll_rates")
    return k


def rate_coefficient_tables(nratec: int = 5000, temstart: float = 1.0, temend: Optional[float] = None, recombination_type: int = 2):
    """k1a..k6a as the reference's driver tabulates them (calc_rates.f:324-337): returns (k[6][nratec], logtem0, logtem9, dlogtem),
    the arguments of StellarTransfer.set_rate_coefficients.  Defaults: the reference's nratec, temstart, temend = 1.e8 (a default-real
    literal), case B."""
    temend = float(np.float32(1.0e8)) if temend is None else float(temend)
    k = np.empty((6, nratec))
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    _check(_lib.load().ftte_rate_coefficient_tables(int(nratec), float(temstart), temend, int(recombination_type), _dp(k), C.byref(a),
                                                    C.byref(b), C.byref(c)), "ftte_rate_coefficient_tables")
    return k, a.value, b.value, c.value


def uniform_table(alpha_quasar: float, alpha_stellar: float, nfreq: int = 400, freqdel: Optional[float] = None):
    """uniformTable (uniformTable.f90): (ksi[2][3], gamma[2][3]) of the quasar and the stellar power-law component."""
    ksi, gamma = np.empty((2, 3)), np.empty((2, 3))
    _check(_lib.load().ftte_uniform_table(int(nfreq), float(np.float32(0.02)) if freqdel is None else float(freqdel), float(alpha_quasar),
                                          float(alpha_stellar), _dp(ksi), _dp(gamma)), "ftte_uniform_table")
    return ksi, gamma


def dust_cross_section(lambda_micron: float, a_smc) -> float:
    """dustCrossSection (dustModule.f90:30-73), SMC curve; a_smc[7][5] as the rows of the reference's data file."""
    a = np.asfortranarray(_f64(a_smc).reshape(7, 5))
    return float(_lib.load().ftte_dust_cross_section(float(lambda_micron), a.ctypes.data_as(C.POINTER(C.c_double))))


class StellarTransfer(DiffuseTransfer):
    """The `if (runStellarTransfer)` block of the reference driver (equiSources.f90:1256-1370) on one GPU.

        st = StellarTransfer(device=0)
        st.set_grid(n, level, box_cm)
        st.set_medium(HI, HeI, HeII, rho, abun2, dust_approximation)
        st.set_zero_rates()
        st.stellar_beta_table(a_smc, wavelength_cm, specific_luminosity, iSpectrum, cS, iMetal, cM)
        st.point_sources(cells, ndot)            # the stars of that population
        rates = st.rates()                       # [6][ncell]: krate24, 25, 26, crate24, 25, 26
    """

    def stellar_beta_table(self, a_smc, wavelength_cm, specific_luminosity, i_spectrum: int, coef_spectrum: float,
                           i_metal: int, coef_metal: float) -> float:
        """stellarBetaTable (stellarBetaTable.f90).  specific_luminosity[nmetal][nspectrum][nwave] (log10, as the
        reference's library); i_spectrum, i_metal 1-based.  Returns totalIntegral; the tables stay on the device."""
        a = np.asfortranarray(_f64(a_smc).reshape(7, 5))
        wl = _f64(wavelength_cm)
        sl = _f64(specific_luminosity)
        if sl.ndim != 3 or sl.shape[2] != wl.size:
            raise ValueError("specific_luminosity must have shape [nmetal][nspectrum][nwave]")
        slf = np.asfortranarray(sl)  # the Fortran array specificLuminosity(nmetal, nspectrum, nwave)
        total = C.c_double()
        self._ok(self._lib.ftte_stellar_beta_table(
            self._ctx, a.ctypes.data_as(C.POINTER(C.c_double)), wl.size, _dp(wl), sl.shape[1], sl.shape[0],
            slf.ctypes.data_as(C.POINTER(C.c_double)), int(i_spectrum), float(coef_spectrum), int(i_metal),
            float(coef_metal), C.byref(total)))
        return total.value

    def set_rate_tables(self, tables):
        tables = _f64(tables)
        if tables.size != int(np.prod(TABLE_SHAPE)):
            raise ValueError("tables must have shape [6][11][11][11][11]")
        self._ok(self._lib.ftte_set_rate_tables(self._ctx, _dp(tables)))

    def rate_tables(self) -> np.ndarray:
        out = np.empty(TABLE_SHAPE)
        self._ok(self._lib.ftte_get_rate_tables(self._ctx, _dp(out)))
        return out

    def get_rates_hydrogen_helium(self, tau, dust_approximation: int = 0) -> np.ndarray:
        """getRatesHydrogenHelium (equiSources.f90:4157) for tau[nsample][4] = (tau1, tau2, tau3, tauDust);
        returns [nsample][3][2] = (numberRate, heatingRate) per reaction, evaluated on the device."""
        tau = _f64(tau).reshape(-1, 4)
        out = np.empty((tau.shape[0], 3, 2))
        self._ok(self._lib.ftte_get_rates_hydrogen_helium(self._ctx, int(dust_approximation), tau.shape[0], _dp(tau), _dp(out)))
        return out

    def set_medium(self, HI, HeI, HeII, rho=None, abun2=None, dust_approximation: int = 0):
        f = [None if a is None else _f64(a) for a in (HI, HeI, HeII, rho, abun2)]
        for a in f:
            if a is not None and self.ncell and a.size != self.ncell:
                raise ValueError("cell fields must have ncell elements")
        self._ok(self._lib.ftte_set_medium(self._ctx, *[None if a is None else _dp(a) for a in f], int(dust_approximation)))

    def set_medium_device(self, HI: int, HeI: int, HeII: int, rho: int = 0, abun2: int = 0, dust_approximation: int = 0):
        self._ok(self._lib.ftte_set_medium_device(self._ctx, *[C.c_void_p(p) if p else None for p in (HI, HeI, HeII, rho, abun2)],
                                                  int(dust_approximation)))

    def set_zero_rates(self):
        """setZeroRates (equiSources.f90:4128)."""
        self._ok(self._lib.ftte_set_zero_rates(self._ctx))

    def locate_cell(self, position: Sequence[int]) -> int:
        """localizeCellFromStar (equiSources.f90:2597): call sequence (1-based) -> 0-based cell-array index."""
        pos = np.ascontiguousarray(position, dtype=np.int32)
        if pos.size % 3 or pos.size < 3:
            raise ValueError("a call sequence has 3 (level + 1) entries")
        cell = C.c_int64()
        self._ok(self._lib.ftte_locate_cell(self._ctx, pos.size // 3 - 1, pos.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(cell)))
        return cell.value

    def point_sources(self, cells, ndot) -> int:
        """Trace the stars in `cells` (0-based cell-array indices) with photon rates `ndot` through the medium, adding
        to the rates on the device.  Returns highestPixelLevel."""
        cells = np.ascontiguousarray(cells, dtype=np.int64)
        ndot = _f64(ndot)
        if cells.size != ndot.size:
            raise ValueError("cells and ndot must have equal length")
        highest = C.c_int()
        self._ok(self._lib.ftte_point_sources(self._ctx, cells.size, cells.ctypes.data_as(C.POINTER(C.c_int64)), _dp(ndot),
                                              C.byref(highest)))
        return highest.value

    def escape(self, nsrc: int) -> dict:
        """Escape bookkeeping of the last point_sources call (equiSources.f90:3198-3233, 1342-1348): dict of remaining [nsrc][7],
        boundary [nsrc][7], dust [nsrc], spectrum [nsrc][300], fraction [nsrc][7]."""
        out = dict(remaining=np.empty((nsrc, 7)), boundary=np.empty((nsrc, 7)), dust=np.empty(nsrc), spectrum=np.empty((nsrc, 300)),
                   fraction=np.empty((nsrc, 7)))
        self._ok(self._lib.ftte_point_escape(self._ctx, nsrc, _dp(out["remaining"]), _dp(out["boundary"]), _dp(out["dust"]),
                                             _dp(out["spectrum"]), _dp(out["fraction"])))
        return out

    def set_output_sigma(self, sigma):
        """outputSigma24, 25, 26, Dust [4][300] for tables handed over with set_rate_tables."""
        sigma = _f64(sigma)
        if sigma.shape != (4, 300):
            raise ValueError("sigma must have shape [4][300]")
        self._ok(self._lib.ftte_set_output_sigma(self._ctx, _dp(sigma)))

    def ray_steps(self) -> int:
        """Cell crossings of the last point_sources call, all rays."""
        return int(self._lib.ftte_point_ray_steps(self._ctx))

    def rates(self) -> np.ndarray:
        out = np.empty((6, max(self.ncell, 1)))
        self._ok(self._lib.ftte_get_point_rates(self._ctx, _dp(out)))
        return out

    def set_rates(self, rates):
        """Replace the device-resident rates (e.g. by their sum over the ranks that traced different stars)."""
        rates = _f64(rates)
        if rates.shape != (6, self.ncell):
            raise ValueError("rates must have shape [6][ncell]")
        self._ok(self._lib.ftte_set_point_rates(self._ctx, _dp(rates)))

    def rates_device_ptr(self) -> int:
        p = C.c_void_p()
        self._ok(self._lib.ftte_point_rates_device(self._ctx, C.byref(p)))
        return p.value

    # -- ionisation equilibrium: solveRateEquations (equiSources.f90:3459-3677) on the device-resident medium
    def set_rate_coefficients(self, logtem0: float, logtem9: float, dlogtem: float, k):
        """k[6][nratec] = k1a..k6a of the reference's calc_rates / coll_rates; table bounds of equiSources.f90:174-176."""
        k = _f64(k)
        if k.ndim != 2 or k.shape[0] != 6:
            raise ValueError("k must have shape [6][nratec]")
        self._ok(self._lib.ftte_set_rate_coefficients(self._ctx, k.shape[1], float(logtem0), float(logtem9), float(dlogtem),
                                                      *[_dp(k[r]) for r in range(6)]))

    def set_temperature(self, tgas):
        tgas = _f64(tgas)
        if self.ncell and tgas.size != self.ncell:
            raise ValueError("tgas must have ncell elements")
        self._ok(self._lib.ftte_set_temperature(self._ctx, _dp(tgas)))

    def _solve(self, fn, J, run_uvb, ksi, uniform, threshold, use_point_rates):
        ksi = None if ksi is None else _f64(ksi).reshape(3, 3)
        uniform = None if uniform is None else _f64(uniform).reshape(3)
        change = C.c_double()
        self._ok(fn(self._ctx, int(bool(run_uvb)), J, None if ksi is None else _dp(ksi), None if uniform is None else _dp(uniform),
                    float(threshold), int(bool(use_point_rates)), C.byref(change)))
        return change.value

    def solve_rate_equations(self, run_uvb: bool, J=None, ksi=None, uniform=None, threshold: float = 0.0,
                             use_point_rates: bool = False) -> float:
        """One equilibrium update of HI, HeI, HeII in the device-resident medium; J[3][ncell] host array (run_uvb) or the
        uniform background rates.  Returns the largest change of a species fraction."""
        J = None if J is None else _f64(J)
        if J is not None and J.shape != (3, self.ncell):
            raise ValueError("J must have shape [3][ncell]")
        return self._solve(self._lib.ftte_solve_rate_equations, None if J is None else _dp(J), run_uvb, ksi, uniform, threshold,
                           use_point_rates)

    def solve_rate_equations_device(self, j_device_ptr: int, ksi, use_point_rates: bool = False) -> float:
        """Same with J[3][ncell] in device memory (e.g. what transport_device wrote)."""
        return self._solve(self._lib.ftte_solve_rate_equations_device, C.c_void_p(j_device_ptr), True, ksi, None, 0.0, use_point_rates)

    def medium(self):
        out = [np.empty(max(self.ncell, 1)) for _ in range(3)]
        self._ok(self._lib.ftte_get_medium(self._ctx, *[_dp(a) for a in out]))
        return out

    def compute_opacities_from_medium(self, beta):
        """computeOpacities (equiSources.f90:4956) on the device-resident HI, HeI, HeII; beta[3][nnu]."""
        beta = _f64(beta)
        if beta.ndim != 2 or beta.shape[0] != 3:
            raise ValueError("beta must have shape [3][nnu]")
        self._ok(self._lib.ftte_compute_opacities(self._ctx, beta.shape[1], _dp(beta)))
        self.nnu = beta.shape[1]

    def assign_uvb_radiation(self, uvb, threshold: float) -> np.ndarray:
        """assignUvbRadiation (transportRoutinesModule.f90:1056): J[nnu][ncell] = uvb where the medium is not self-shielded."""
        uvb = _f64(uvb)
        J = np.empty((uvb.size, max(self.ncell, 1)))
        self._ok(self._lib.ftte_assign_uvb_radiation(self._ctx, uvb.size, _dp(uvb), float(threshold), _dp(J)))
        return J

    def rate_equation_steps(self) -> int:
        return int(self._lib.ftte_rate_equation_steps(self._ctx))
