"""The Fortran host (fortran/ftte_demo_driver: ISO_C_BINDING module + libftte.so) on a real GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fortran", "ftte_demo_driver")


def test_fortran_demo_driver():
    if not os.path.exists(EXE):
        pytest.skip("fortran/ftte_demo_driver not built (no Fortran compiler at build time)")
    out = subprocess.run([EXE, "40", "2"], capture_output=True, text=True, timeout=300, cwd=os.path.join(ROOT, "fortran"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ftte_demo_driver OK" in out.stdout


def test_fortran_point_source_host(golden, tmp_path):
    """The Fortran star-loop host (fortran/ftte_demo_point) on the refined, dusty case the reference itself traced."""
    import numpy as np
    exe = os.path.join(ROOT, "fortran", "ftte_demo_point")
    if not os.path.exists(exe):
        pytest.skip("fortran/ftte_demo_point not built (no Fortran compiler at build time)")
    g = golden("point10_refined_dust")
    case, out = tmp_path / "case.bin", tmp_path / "rates.bin"
    with open(case, "wb") as f:
        f.write(np.array([int(g["n"]), g["level"].size, g["src_leaf"].size, int(g["dust"])], "<i4").tobytes())
        f.write(np.array([float(g["box"])], "<f8").tobytes())
        f.write(g["level"].astype("<i4").tobytes())
        for k in ("HI", "HeI", "HeII", "rho", "abun2"):
            f.write(g[k].astype("<f8").tobytes())
        f.write(g["src_leaf"].astype("<i8").tobytes())
        f.write(g["src_weight"].astype("<f8").tobytes())
        f.write(g["tables"].astype("<f8").tobytes())
    res = subprocess.run([exe, str(case), str(out)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "ftte_demo_point OK" in res.stdout, res.stdout + res.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    nc = g["level"].size
    rates = raw[:8 * 6 * nc].view("<f8").reshape(6, nc)
    highest = int(raw[8 * 6 * nc:].view("<i4")[0])
    assert highest == int(g["highestPixelLevel"])
    ref = g["krate"]
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert np.all(np.abs(rates - ref) <= 1e-9 * np.abs(ref) + 1e-13 * scale)


DROPIN = os.path.join(ROOT, "tests", "fortran", "dropin_check")


def test_dropin_uvb_transfer_on_the_reference_tree(golden, tmp_path):
    """fortran/ftte_uvb_transfer.f90 -- the replacement of equiSources.f90:1383-1806 -- run as the reference driver would run
    it: on a fully threaded tree of the reference's own zoneType cells (module `definitions`), with the reference's own
    direction list (nAngularLevel = 3: 192 pixels, single-precision weight).  Jmean1..3 in the tree against what the
    reference's own sweep left there (tests/golden/dropin_uvb_192dir.npz)."""
    import numpy as np
    import _oracle as O
    if not os.path.exists(DROPIN):
        pytest.skip("tests/fortran/dropin_check not built (needs oracle/_ref and a Fortran compiler at build time)")
    g = golden("dropin_uvb_192dir")
    n, level = int(g["n"]), g["level"]
    case, out = tmp_path / "case.bin", tmp_path / "J.bin"
    with open(case, "wb") as f:
        f.write(np.array([n, level.size], "<i4").tobytes())
        f.write(np.array([float(g["box"])] + list(g["uvb"]), "<f8").tobytes())
        f.write(level.astype("<i4").tobytes())
        f.write(np.ascontiguousarray(g["kappa"], "<f8").tobytes())          # kappa(ncell,3) in Fortran order
    res = subprocess.run([DROPIN, "uvb", str(case), str(out)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "dropin_check OK" in res.stdout, res.stdout + res.stderr
    J = np.fromfile(out, "<f8").reshape(3, level.size)
    _, noise = O.sweep_tree(n, level, g["kappa"], float(g["box"]), g["phi"], g["theta"], g["w"], g["uvb"], with_noise=True)
    eps = np.finfo(float).eps
    assert np.all(np.abs(J - g["J"]) <= 8 * noise + 12 * 4 * n * eps * np.abs(g["J"]))
    # and bit for bit what the library gives through the Python host for the same list
    assert np.array_equal(J, O.sweep_tree(n, level, g["kappa"], float(g["box"]), g["phi"], g["theta"], g["w"], g["uvb"], arith=O.ARITH_DEVICE))


def test_dropin_rate_equations_on_the_reference_tree(golden, tmp_path):
    """fortran/ftte_rate_equations.f90 in place of the solveRateEquations calls of equiSources.f90:1824-1831, on the reference's
    tree and module tables: HI, HeI, HeII equal to the reference's own routine bit for bit."""
    import numpy as np
    if not os.path.exists(DROPIN):
        pytest.skip("tests/fortran/dropin_check not built (needs oracle/_ref and a Fortran compiler at build time)")
    g = golden("chem_uvb_refined")
    n, level = int(g["n"]), g["level"]
    case, out = tmp_path / "case.bin", tmp_path / "species.bin"
    with open(case, "wb") as f:
        f.write(np.array([n, level.size, 1], "<i4").tobytes())
        f.write(np.array([float(g["box"])], "<f8").tobytes())
        f.write(level.astype("<i4").tobytes())
        for a in (g["rho"], g["tgas"], g["HI"], g["HeI"], g["HeII"], g["krate"][0], g["krate"][1], g["krate"][2], g["J"][0], g["J"][1], g["J"][2]):
            f.write(np.asarray(a, "<f8").tobytes())
        f.write(np.asarray(g["ksi"], "<f8").reshape(3, 3).tobytes())        # [group][reaction] == Fortran ksiIn(reaction, group)
        f.write(np.array([float(g["logtem0"]), float(g["logtem9"]), float(g["dlogtem"])], "<f8").tobytes())
        f.write(np.ascontiguousarray(g["k"], "<f8").tobytes())
    res = subprocess.run([DROPIN, "chem", str(case), str(out)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "dropin_check OK" in res.stdout, res.stdout + res.stderr
    got = np.fromfile(out, "<f8").reshape(3, level.size)
    assert np.array_equal(got[0], g["HI_out"]) and np.array_equal(got[1], g["HeI_out"]) and np.array_equal(got[2], g["HeII_out"])


@pytest.mark.parametrize("name,metal_offset", [("point10_refined_dust", 0.1), ("point12_escape", 1.35)])
def test_dropin_stellar_transfer_on_the_reference_tree(golden, tmp_path, name, metal_offset):
    """fortran/ftte_stellar_transfer.f90 in place of the star loop equiSources.f90:1260-1362: stars given as the reference
    holds them (level + call sequence), population picked from the host cell's metallicity as :1281-1291 does, module
    arrays a_smc / wavelength / specificLuminosity / metallicity; krate24..26, crate24..26 in the tree against the
    reference's own tracer, and the `src:` line the reference prints per star (:1353-1357: escape fractions at the seven
    output radii) against the fractions its tracer's bookkeeping gives (tests/golden/point10_refined_dust.npz: a box so
    small that all light counts as gone; point12_escape.npz: an 80 kpc box)."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_point as M
    if not os.path.exists(DROPIN):
        pytest.skip("tests/fortran/dropin_check not built (needs oracle/_ref and a Fortran compiler at build time)")
    g = golden(name)
    n, level = int(g["n"]), g["level"]
    a_smc, wavelength, spec = M.synthetic_population()
    # call sequences of the leaves, as readCellArray.f90:154-187 numbers them
    seqs, cursor = [], 0

    def grow(lvl, path):
        nonlocal cursor
        if level[cursor] == lvl:
            seqs.append(path)
            cursor += 1
        else:
            for a in (1, 2):
                for b in (1, 2):
                    for c in (1, 2):
                        grow(lvl + 1, path + [a, b, c])
    for i in range(1, n + 1):
        for j in range(1, n + 1):
            for k in range(1, n + 1):
                grow(0, [i, j, k])
    # a metallicity grid that makes the drop-in pick the golden's (iMetal, coefMetal) -- (1, 0.1) for abun2 = 0.2, (2, 0.35) for
    # abun2 = 0.05 -- from the host cell's abundance
    tmp = np.log10(float(g["abun2"][0]))
    assert (int(g["iMetal"]) - 1) + float(g["coefMetal"]) == pytest.approx(metal_offset)
    metallicity = tmp - metal_offset + np.arange(5.0)
    case, out = tmp_path / "case.bin", tmp_path / "rates.bin"
    with open(case, "wb") as f:
        f.write(np.array([n, level.size, g["src_leaf"].size, int(g["dust"])], "<i4").tobytes())
        f.write(np.array([float(g["box"])], "<f8").tobytes())
        f.write(level.astype("<i4").tobytes())
        for k in ("HI", "HeI", "HeII", "rho", "abun2"):
            f.write(g[k].astype("<f8").tobytes())
        for leaf, weight in zip(g["src_leaf"], g["src_weight"]):
            seq = seqs[int(leaf)]
            f.write(np.array([len(seq) // 3 - 1, int(weight)] + seq + [0] * (33 - len(seq)), "<i4").tobytes())
        f.write(np.asfortranarray(a_smc).tobytes(order="F"))
        f.write(np.asarray(wavelength, "<f8").tobytes())
        f.write(np.asfortranarray(spec).tobytes(order="F"))
        f.write(metallicity.astype("<f8").tobytes())
        f.write(np.array([int(g["iSpectrum"])], "<i4").tobytes())
        f.write(np.array([float(g["coefSpectrum"])], "<f8").tobytes())
    res = subprocess.run([DROPIN, "stellar", str(case), str(out)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "dropin_check OK" in res.stdout, res.stdout + res.stderr
    rates = np.fromfile(out, "<f8").reshape(6, level.size)
    ref = g["krate"]
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert np.all(np.abs(rates - ref) <= 1e-9 * np.abs(ref) + 1e-13 * scale)
    assert np.array_equal(rates == 0, ref == 0)
    # src: iStar level neutralFraction highestPixelLevel fraction(1:7) weight   (format 1015, equiSources.f90:1357)
    lines = [ln.split() for ln in res.stdout.splitlines() if ln.startswith("src:")]
    assert len(lines) == g["src_leaf"].size
    for s, ln in enumerate(lines):
        assert int(ln[1]) == s + 1 and int(ln[-1]) == int(g["src_weight"][s])
        assert np.allclose([float(x) for x in ln[5:12]], g["fraction"][s], rtol=0, atol=6e-6)  # five decimals printed
    assert max(int(ln[4]) for ln in lines) == int(g["highestPixelLevel"])


def test_fortran_host_runs_the_whole_iteration(golden, tmp_path):
    """fortran/ftte_demo_loop: star loop -> opacities -> diffuse sweep -> equilibrium, three times, from a Fortran host;
    the final species and J against the same loop through the oracle."""
    import numpy as np
    import _oracle as O
    exe = os.path.join(ROOT, "fortran", "ftte_demo_loop")
    if not os.path.exists(exe):
        pytest.skip("fortran/ftte_demo_loop not built (no Fortran compiler at build time)")
    g = golden("chem_uvb_refined")
    tabs = golden("point16_homogeneous")["tables"]
    n, level, box = int(g["n"]), g["level"], float(g["box"])
    nc = level.size
    alpha = np.array([1.8, 1.5, 1.2])
    uvb = np.array([2e-22, 1e-22, 3e-23])
    src, ndot = np.array([5, nc // 2], np.int64), np.array([50.0, 20.0])
    niter, L = 3, 2
    case, out = tmp_path / "case.bin", tmp_path / "out.bin"
    with open(case, "wb") as f:
        f.write(np.array([n, nc, src.size, niter, L, g["k"].shape[1]], "<i4").tobytes())
        f.write(np.concatenate([[box], alpha, uvb]).astype("<f8").tobytes())
        f.write(level.astype("<i4").tobytes())
        for k in ("rho", "tgas", "HI", "HeI", "HeII"):
            f.write(g[k].astype("<f8").tobytes())
        f.write(src.astype("<i8").tobytes())
        f.write(ndot.astype("<f8").tobytes())
        f.write(tabs.astype("<f8").tobytes())
        f.write(np.array([float(g["logtem0"]), float(g["logtem9"]), float(g["dlogtem"])], "<f8").tobytes())
        f.write(np.ascontiguousarray(g["k"], "<f8").tobytes())      # k(nratec,6) in Fortran order == [6][nratec]
    res = subprocess.run([exe, str(case), str(out)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "ftte_demo_loop OK" in res.stdout, res.stdout + res.stderr
    raw = np.fromfile(out, "<f8")
    species, J = raw[:3 * nc].reshape(3, nc), raw[3 * nc:].reshape(3, nc)
    # the same loop through the oracle
    beta_g, ksi, _ = O.uvb_beta_table(alpha)              # [group][24, 25, 26]
    beta = np.array([beta_g[:, 0], beta_g[:, 2], beta_g[:, 1]])   # [species HI, HeI, HeII][group]
    phi, theta, _ = O.healpix_directions(L)
    w = np.full(phi.size, float(np.float32(1.0) / np.float32(phi.size)))
    HI, HeI, HeII = g["HI"].copy(), g["HeI"].copy(), g["HeII"].copy()
    for _ in range(niter):
        rates, _ = O.point_sources(n, level, HI, HeI, HeII, g["rho"], g["rho"], box, 0, src, ndot, tabs.reshape(6, -1))
        kappa = O.compute_opacities(HI, HeI, HeII, beta)
        Jo = O.sweep_tree(n, level, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
        Jo = Jo[0] if isinstance(Jo, tuple) else Jo
        HI, HeI, HeII, status, _ = O.solve_rate_equations(n, level, box, g["rho"], g["tgas"], HI, HeI, HeII, rates[:3], True, Jo, ksi, None, 0.0,
                                                          float(g["logtem0"]), float(g["logtem9"]), float(g["dlogtem"]), g["k"])
        assert status == 0
    for mine, ref in zip(species, (HI, HeI, HeII)):
        assert np.all(np.abs(mine - ref) <= 1e-8 * np.abs(ref) + 1e-30)
    assert np.all(np.abs(J - Jo) <= 1e-9 * np.abs(Jo))
