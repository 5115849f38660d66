"""Parity of the HIP path, through the C ABI (radiativetransfer_amd.api is a ctypes veneer over include/ftte.h),
against the oracle and against the vectors the reference's own modules produced.

Tolerances (fp64), all stated here:
  * bitwise  -- against the oracle evaluated with the device arithmetic (csrc/ftte_math.h on the host), whenever the
                order in which directions are summed is the same on both sides (single direction; or slots = 1
                and all directions in one memory layout);
  * SUM_RTOL -- same arithmetic, different summation order of the per-direction terms (slots, three layouts);
  * reference-- against the reference formula (Iin-Iout)/log(Iin/Iout) (goldens, oracle REFERENCE arithmetic):
                |J - Jref| <= 8 * noise + 12 n eps |Jref|, where `noise` is the per-cell rounding noise of the reference's
                own quotient, eps/(2 tau) per segment (oracle/ftte_oracle.h), and 12 n eps covers 1-ulp differences of
                exp along a chain of <= 3n segments.  For inputs whose segments all have tau >= 1e-3 this is < 1e-12 relative.
"""
import numpy as np
import pytest

import _oracle as O
from radiativetransfer_amd import synthetic

pytestmark = pytest.mark.gpu

EPS = np.finfo(np.float64).eps
SUM_RTOL = 64 * EPS


@pytest.fixture(autouse=True, params=["tiles", "bricks"])
def organisation(engine, request):
    """Every test of this file runs with both organisations of the uniform-grid sweep: the ray-following tile kernel (option
    "engine" = 1; its rows / slots / stack variants are options the bricks ignore) and the cell-fixed bricks a uniform grid
    takes by default (more of them in test_brick_gpu.py).  Refined cell arrays take the forest path either way."""
    engine.set_option("engine", 1 if request.param == "tiles" else 2)
    yield request.param
    engine.set_option("engine", 0)


def one_per_izone():
    phi, theta, _ = O.healpix_directions(3)
    pick = {}
    for p, t in zip(phi, theta):
        pick.setdefault(O.fold_direction(p, t)[2], (p, t))
    return [pick[z] for z in range(1, 25)]


def reference_bound(n, J_ref, noise):
    return 8 * noise + 12 * n * EPS * np.abs(J_ref)


@pytest.mark.parametrize("rows", [4, 8, 16])
@pytest.mark.parametrize("n", [5, 16, 70])
def test_every_izone_bitwise(engine, rows, n):
    """One direction per izone (all 24 rotations, all five ray classes), every row variant of the kernel, grids
    smaller than a tile (5), one tile wide (16) and several tiles wide with ragged edges (70)."""
    kappa, uvb, box = synthetic.uniform_workload(n, 2, seed=n, tau_median=0.3)
    engine.set_option("rows", rows)
    engine.set_uniform_grid(n, box)
    engine.set_opacity(kappa)
    for p, t in one_per_izone():
        phi, theta, w = np.array([p]), np.array([t]), np.array([0.37])
        J = engine.transport(phi, theta, w, uvb)
        ref = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
        assert np.array_equal(J, ref), f"izone {O.fold_direction(p, t)[2]}"
    engine.set_option("rows", 8)


def test_config1_plumbing_case(engine, golden, organisation):
    """BASELINE configs[0] on the GPU: 64^3, 1 frequency group, 6 directions (izones 1, 2, 3, 13, 14, 15), against the J the
    reference's own driver lines produced (tests/golden/config1_uniform64_6dir.npz), with the reference bound, and against
    the oracle with the device arithmetic."""
    from test_oracle_golden import config1_case
    g, n, kappa, box = config1_case(golden)
    engine.set_uniform_grid(n, box)
    engine.set_opacity(kappa)
    J = engine.transport(g["phi"], g["theta"], g["w"], g["uvb"])
    assert J.shape == (1, n ** 3)
    args = (n, kappa, box, g["phi"], g["theta"], g["w"], g["uvb"])
    assert np.allclose(J, O.sweep_uniform(*args, arith=O.ARITH_DEVICE), rtol=SUM_RTOL, atol=0)
    _, noise = O.sweep_uniform(*args, with_noise=True)
    assert np.all(np.abs(J - g["J"]) <= reference_bound(n, g["J"], noise))
    # the same six directions one by one: each alone is bit-identical to the oracle (no summation order involved)
    for d in range(6):
        one = (g["phi"][d:d + 1], g["theta"][d:d + 1], g["w"][d:d + 1])
        assert np.array_equal(engine.transport(*one, g["uvb"]), O.sweep_uniform(n, kappa, box, *one, g["uvb"], arith=O.ARITH_DEVICE))


GOLDEN_UNIFORM = ["uniform8_transparent", "uniform16_constant", "uniform16_lognormal_24zones", "uniform24_lognormal_48dir"]


@pytest.mark.parametrize("slots", [1, 5, 8])
@pytest.mark.parametrize("name", GOLDEN_UNIFORM)
def test_reference_goldens(engine, golden, name, slots):
    """The reference's own outputs (3 frequency groups, tests/golden/) within the tau-aware tolerance."""
    g = golden(name)
    n = int(g["n"])
    args = (g["kappa"], float(g["box"]), g["phi"], g["theta"], g["w"], g["uvb"])
    engine.set_option("slots", slots)
    engine.set_grid(n, g["level"], float(g["box"]))
    engine.set_opacity(g["kappa"])
    J = engine.transport(g["phi"], g["theta"], g["w"], g["uvb"])
    engine.set_option("slots", 8)   # the library default
    _, noise = O.sweep_uniform(n, *args, with_noise=True)
    assert np.all(np.abs(J - g["J"]) <= reference_bound(n, g["J"], noise))
    # and against the same arithmetic on the host: only the summation order differs
    Jd = O.sweep_uniform(n, *args, arith=O.ARITH_DEVICE)
    assert np.allclose(J, Jd, rtol=SUM_RTOL, atol=0)


def test_eight_groups_as_three_reference_runs(engine):
    """BASELINE's 8 frequency groups: groups are independent inside the sweep, so 8 groups must equal three
    oracle runs on group triples (the reference hard-wires three groups)."""
    n = 20
    kappa, uvb, box = synthetic.uniform_workload(n, 8, seed=7, tau_median=0.4)
    phi, theta, w = O.healpix_directions(2)
    engine.set_uniform_grid(n, box)
    engine.set_opacity(kappa)
    J = engine.transport(phi, theta, w, uvb)
    for lo in (0, 3, 5):
        sl = slice(lo, lo + 3)
        ref, noise = O.sweep_uniform(n, kappa[sl], box, phi, theta, w, uvb[sl], with_noise=True)
        assert np.all(np.abs(J[sl] - ref) <= reference_bound(n, ref, noise))


def test_transparent_and_empty(engine):
    n = 33
    engine.set_uniform_grid(n, 2.5)
    engine.set_opacity(np.zeros((2, n ** 3)))
    phi, theta, w = O.healpix_directions(2)
    uvb = np.array([3e-21, 1e-22])
    J = engine.transport(phi, theta, w, uvb)
    assert np.allclose(J, (uvb * w.sum())[:, None], rtol=8 * EPS, atol=0)  # kappa = 0: J = uvb * sum(w)
    J0 = engine.transport(phi[:0], theta[:0], w[:0], uvb)  # no directions: J = 0
    assert J0.shape == (2, n ** 3) and not J0.any()


def test_single_cell_grid(engine):
    kappa = np.array([[0.7], [0.0]])
    engine.set_uniform_grid(1, 1.0)
    engine.set_opacity(kappa)
    phi, theta, w = O.healpix_directions(1)
    uvb = np.array([1e-21, 2e-21])
    J = engine.transport(phi, theta, w, uvb)
    ref = O.sweep_uniform(1, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
    assert np.allclose(J, ref, rtol=SUM_RTOL, atol=0)


def test_opaque_medium_underflows_like_the_reference(engine):
    """tau ~ 100 per cell: intensities underflow to exactly 0 deep inside; J must stay finite, non-negative and
    agree with the reference arithmetic wherever J is a normal number."""
    n = 12
    kappa = np.full((1, n ** 3), 100.0 * n)
    phi, theta, w = O.healpix_directions(1)
    uvb = np.array([1e-21])
    engine.set_uniform_grid(n, 1.0)
    engine.set_opacity(kappa)
    J = engine.transport(phi, theta, w, uvb)
    ref = O.sweep_uniform(n, kappa, 1.0, phi, theta, w, uvb)
    assert np.all(np.isfinite(J)) and np.all(J >= 0)
    big = ref > 1e-290
    assert np.allclose(J[big], ref[big], rtol=1e-11, atol=0)
    assert np.all(J[~big] <= 1e-289)


def test_species_opacities(engine):
    n = 9
    rng = np.random.default_rng(5)
    HI, HeI, HeII = rng.lognormal(0, 1, (3, n ** 3)) * 3.0
    beta = rng.random((3, 4))
    engine.set_uniform_grid(n, 1.0)
    engine.compute_opacities(HI, HeI, HeII, beta)
    phi, theta, w = O.healpix_directions(1)
    uvb = np.full(4, 1e-21)
    J = engine.transport(phi, theta, w, uvb)
    kappa = O.compute_opacities(HI, HeI, HeII, beta)
    ref = O.sweep_uniform(n, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
    assert np.allclose(J, ref, rtol=SUM_RTOL, atol=0)


def test_caches_follow_their_inputs(engine):
    """The library caches the three opacity layouts and the direction plan; changing any input must invalidate them:
    new opacities (all layouts), a new direction list, a new grid size, new tuning."""
    rng = np.random.default_rng(11)
    dirs24 = np.array(one_per_izone())
    uvb = np.array([1e-21, 3e-22])
    for n in (12, 9):
        engine.set_uniform_grid(n, 1.5)
        for trial in range(2):
            kappa = rng.lognormal(0, 1, (2, n ** 3)) * n * 0.3
            engine.set_opacity(kappa)
            for sel in (slice(0, 24), slice(5, 17)):
                phi, theta = dirs24[sel, 0].copy(), dirs24[sel, 1].copy()
                w = np.full(len(phi), 1.0 / len(phi))
                J = engine.transport(phi, theta, w, uvb)
                ref = O.sweep_uniform(n, kappa, 1.5, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
                assert np.allclose(J, ref, rtol=SUM_RTOL, atol=0)
    engine.set_option("rows", 4)
    engine.set_option("waves", 6)
    J2 = engine.transport(phi, theta, w, uvb)
    engine.set_option("rows", 8)
    engine.set_option("waves", 4)
    assert np.allclose(J2, ref, rtol=SUM_RTOL, atol=0)


def test_arbitrary_directions_and_weights(engine):
    """Directions need not come from HEALPix: random angles, unequal weights, one frequency group."""
    rng = np.random.default_rng(21)
    n = 31
    pi = O.lib().fo_pi()
    phi = rng.uniform(0.01, 2 * pi - 0.01, 40)
    theta = rng.uniform(0.02, 0.5 * pi - 0.02, 40) * rng.choice([-1, 1], 40)
    keep = [i for i in range(40) if _foldable(phi[i], theta[i])]
    phi, theta = phi[keep], theta[keep]
    w = rng.random(len(phi))
    kappa, uvb, box = synthetic.uniform_workload(n, 1, seed=8, tau_median=0.5)
    engine.set_uniform_grid(n, box)
    engine.set_opacity(kappa)
    J = engine.transport(phi, theta, w, uvb)
    ref, noise = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, with_noise=True)
    assert np.all(np.abs(J - ref) <= reference_bound(n, ref, noise))


@pytest.mark.parametrize("n,nnu,tau_median", [(64, 3, 0.1), (72, 2, 1.0), (130, 1, 0.01)])
def test_gpu_within_64_eps_of_the_exact_evaluation(engine, n, nnu, tau_median):
    """The yardstick that shares nothing with the product (oracle ARITH_EXACT: every segment in extended precision, rounded once;
    tests/test_exact_arithmetic.py): the GPU's J for 48 directions lies within 64 eps of it cell by cell -- 32 for the segment
    arithmetic along the rays, the rest for the order of the sum over directions -- on whole-brick and ragged grids, thin and thick
    fields, where the reference's own double-precision evaluation is off by up to 10^7 eps (the noise of its log-mean)."""
    phi, theta, w = O.healpix_directions(2)
    kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=77, tau_median=tau_median)
    engine.set_uniform_grid(n, box)
    engine.set_opacity(kappa)
    J = engine.transport(phi, theta, w, uvb)
    exact = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_EXACT)
    assert np.all(np.abs(J - exact) <= 64 * EPS * exact)
    if tau_median <= 0.1:
        reference = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb)
        assert np.max(np.abs(reference - exact) / exact) > 100 * np.max(np.abs(J - exact) / exact)


def _foldable(p, t):
    try:
        O.fold_direction(p, t)
        return True
    except ValueError:
        return False


def test_call_order_and_refusals(engine):
    import radiativetransfer_amd as rt
    with rt.DiffuseTransfer() as e:
        with pytest.raises(rt.FtteError) as err:
            e.set_opacity(np.zeros((1, 8)))
        assert err.value.status == "FTTE_ERR_STATE"
        with pytest.raises(rt.FtteError) as err:
            e.set_grid((4, 4, 5), np.zeros(80, np.int32), 1.0)
        assert err.value.status == "FTTE_ERR_NOT_CUBIC"
        with pytest.raises(rt.FtteError) as err:
            e.set_grid(2, np.array([0, 0, 0, 1, 0, 0, 0, 0], np.int32), 1.0)  # a lone level-1 leaf list that runs out
        assert err.value.status == "FTTE_ERR_LEVELS"
        e.set_grid(2, synthetic.refine_levels(2, [(0, 0, 0)]), 1.0)  # a refined cell array is accepted
        assert e.ncell == 15
        e.set_uniform_grid(4, 1.0)
        with pytest.raises(rt.FtteError) as err:
            e.transport([0.3], [0.4], [1.0], [1e-21])
        assert err.value.status == "FTTE_ERR_STATE"
        e.set_opacity(np.ones((1, 64)))
        with pytest.raises(rt.FtteError) as err:
            e.transport([0.0], [0.4], [1.0], [1e-21])  # phi on a quadrant boundary: the reference stops
        assert err.value.status == "FTTE_ERR_PHI"
        with pytest.raises(rt.FtteError) as err:
            e.set_option("rows", 5)
        assert err.value.status == "FTTE_ERR_ARG"
        # every option refuses what it cannot mean, and unknown names
        for key, bad in (("engine", 3), ("chunk", -1), ("group", 9), ("share", 3), ("lanes", 0), ("brick_waves", 5), ("dataflow", 4),
                         ("team", 3), ("hybrid", 2), ("pipelines", 5), ("box_lanes", 3), ("forest_batch", -1), ("graph", 2),
                         ("hybrid_slots", 3), ("no such option", 1)):
            with pytest.raises(rt.FtteError) as err:
                e.set_option(key, bad)
            assert err.value.status == "FTTE_ERR_ARG", key
        for key, good in (("engine", 2), ("chunk", 8), ("group", 2), ("share", 1), ("lanes", 3), ("brick_waves", 3), ("dataflow", 0),
                          ("team", 0), ("hybrid", 1), ("pipelines", 2), ("box_lanes", 16), ("forest_batch", 7), ("graph", 0),
                          ("hybrid_slots", 1)):
            e.set_option(key, good)
        assert e.counter("hybrid_boxes") == 0 and e.counter("hybrid_passes") == 0 and e.counter("no such counter") == -1


# ---- refined cell arrays (setRaysRefined / findNeighbours / transport with the coarse-neighbour rule) -----------------

@pytest.mark.parametrize("name", ["amr8_block_level1", "amr6_scattered_level2"])
def test_refined_goldens(engine, golden, name):
    """The reference's own outputs on refined cell arrays; and the oracle's tree sweep with the device arithmetic
    bit for bit (the forest path adds directions in list order, like the reference)."""
    g = golden(name)
    n = int(g["n"])
    args = (g["kappa"], float(g["box"]), g["phi"], g["theta"], g["w"], g["uvb"])
    engine.set_grid(n, g["level"], float(g["box"]))
    engine.set_opacity(g["kappa"])
    J = engine.transport(g["phi"], g["theta"], g["w"], g["uvb"])
    assert np.array_equal(J, O.sweep_tree(n, g["level"], *args, arith=O.ARITH_DEVICE))
    _, noise = O.sweep_tree(n, g["level"], *args, with_noise=True)
    assert np.all(np.abs(J - g["J"]) <= reference_bound(4 * n, g["J"], noise))


def test_refined_ragged_tree(engine):
    n = 5
    rng = np.random.default_rng(17)

    def cell(depth, p):
        if depth < 3 and rng.random() < p:
            out = []
            for _ in range(8):
                out += cell(depth + 1, p * 0.6)
            return out
        return [depth]
    level = []
    for b in range(n ** 3):
        level += cell(0, 0.25 if b % 7 else 0.9)
    level = np.array(level, np.int32)
    kappa = rng.lognormal(0, 1, (3, len(level))) * n * 0.4 * (2.0 ** level)[None, :]
    phi, theta, w = O.healpix_directions(2)
    uvb = np.array([1e-21, 4e-22, 1e-22])
    engine.set_grid(n, level, 1.0)
    engine.set_opacity(kappa)
    J = engine.transport(phi, theta, w, uvb)
    assert np.array_equal(J, O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE))
    assert J.shape == (3, len(level))
    J0 = engine.transport(phi[:0], theta[:0], w[:0], uvb)
    assert not J0.any()


@pytest.mark.parametrize("nnu", [1, 5, 7, 100])
def test_refined_any_number_of_groups(engine, golden, nnu):
    """Group counts that are not powers of two (the general thread mapping) and beyond 96 (no cell-major copy of kappa)."""
    g = golden("amr8_block_level1")
    n, level = int(g["n"]), g["level"]
    rng = np.random.default_rng(nnu)
    kappa = rng.lognormal(0, 1, (nnu, level.size)) * n * 0.3 * (2.0 ** level)[None, :]
    uvb = 10 ** rng.uniform(-22, -21, nnu)
    phi, theta, w = O.healpix_directions(1)
    engine.set_grid(n, level, 1.0)
    engine.set_opacity(kappa)
    J = engine.transport(phi, theta, w, uvb)
    assert np.array_equal(J, O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE))


def test_forest_path_on_a_uniform_grid_equals_the_tiled_kernel(engine):
    n = 18
    kappa, uvb, box = synthetic.uniform_workload(n, 2, seed=3, tau_median=0.4)
    phi, theta, w = O.healpix_directions(2)
    engine.set_uniform_grid(n, box)
    engine.set_opacity(kappa)
    J_tiled = engine.transport(phi, theta, w, uvb)
    engine.set_option("forest", 1)
    engine.set_uniform_grid(n, box)
    engine.set_opacity(kappa)
    J_forest = engine.transport(phi, theta, w, uvb)
    engine.set_option("forest", 0)
    engine.set_uniform_grid(n, box)
    assert np.array_equal(J_forest, O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE))
    assert np.allclose(J_tiled, J_forest, rtol=SUM_RTOL, atol=0)


def test_nested_patch_like_config4(engine):
    """BASELINE configs[3] in small: a cubic base grid whose central block is refined once (here 24^3 with the central
    8^3 block), 8 frequency groups; two directions against the oracle bit for bit, the whole 48-direction set for
    sanity (0 < J <= inflow)."""
    n = 24
    blocks = [(8 + a, 8 + b, 8 + c) for a in range(8) for b in range(8) for c in range(8)]
    level = synthetic.refine_levels(n, blocks, depth=1)
    assert len(level) == n ** 3 - 512 + 4096
    rho = synthetic.lognormal_density(len(level), seed=40)
    _, s_nu, uvb = synthetic.frequency_groups(8)
    kappa = (0.2 * n) * s_nu[:, None] * rho[None, :]
    phi, theta, w = O.healpix_directions(2)
    engine.set_grid(n, level, 1.0)
    engine.set_opacity(kappa)
    J = engine.transport(phi, theta, w, uvb)
    assert np.all(J > 0) and np.all(J <= uvb[:, None] * (1 + 1e-12))
    sel = [0, 7]
    J2 = engine.transport(phi[[5, 29]], theta[[5, 29]], w[[5, 29]], uvb)
    ref = O.sweep_tree(n, level, kappa[sel], 1.0, phi[[5, 29]], theta[[5, 29]], w[[5, 29]], uvb[sel], arith=O.ARITH_DEVICE)
    assert np.array_equal(J2[sel], ref)


# ---- emission: the reference's eta term (:676) and the build's source function -------------------------------------------

@pytest.mark.parametrize("which", ["eta", "src"])
def test_emission_uniform(engine, which):
    n = 21
    kappa, uvb, box = synthetic.uniform_workload(n, 3, seed=12, tau_median=0.4)
    rng = np.random.default_rng(6)
    x = rng.random(kappa.shape) * (2e-22 if which == "eta" else 3e-21)
    kw = {which: x}
    engine.set_uniform_grid(n, box)
    engine.set_opacity(kappa)
    (engine.set_emissivity if which == "eta" else engine.set_source_function)(x)
    # one direction per izone: bit for bit against the device arithmetic on the host
    for p, t in one_per_izone()[::3]:
        phi, theta, w = np.array([p]), np.array([t]), np.array([0.4])
        J = engine.transport(phi, theta, w, uvb)
        assert np.array_equal(J, O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE, **kw))
    # a direction set against the reference's formulae (libm exp/log, (Iin-Iout)/log(Iin/Iout))
    phi, theta, w = O.healpix_directions(2)
    J = engine.transport(phi, theta, w, uvb)
    ref, noise = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, with_noise=True, **kw)
    assert np.all(np.abs(J - ref) <= 8 * noise)
    # switching emission off restores the reference as shipped
    engine.set_emissivity(None)
    J0 = engine.transport(phi, theta, w, uvb)
    assert np.allclose(J0, O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE), rtol=SUM_RTOL, atol=0)


def test_emission_refined_and_equilibrium(engine, golden):
    g = golden("amr8_block_level1")
    n = int(g["n"])
    rng = np.random.default_rng(9)
    S = rng.random(g["kappa"].shape) * 2e-21
    engine.set_grid(n, g["level"], float(g["box"]))
    engine.set_opacity(g["kappa"])
    engine.set_source_function(S)
    J = engine.transport(g["phi"], g["theta"], g["w"], g["uvb"])
    ref = O.sweep_tree(n, g["level"], g["kappa"], float(g["box"]), g["phi"], g["theta"], g["w"], g["uvb"], src=S,
                       arith=O.ARITH_DEVICE)
    assert np.array_equal(J, ref)
    # radiative equilibrium: S = inflow is a fixed point, on the refined grid too
    Seq = np.repeat(g["uvb"][:, None], len(g["level"]), 1)
    engine.set_source_function(Seq)
    Jeq = engine.transport(g["phi"], g["theta"], g["w"], g["uvb"])
    assert np.allclose(Jeq, g["uvb"][:, None] * g["w"].sum(), rtol=64 * EPS, atol=0)
    engine.set_source_function(None)


def test_source_iteration_against_host_restatement_and_thick_limit(engine):
    """Lambda iteration S = (1-eps) J + eps B: three iterations reproduce the same loop done with the oracle on the host;
    in an optically thick box the interior converges to J = B."""
    import torch
    from radiativetransfer_amd.iteration import SourceIteration
    n, nnu, eps = 16, 2, 0.5
    kappa = np.stack([np.full(n ** 3, 3.0 * n), np.full(n ** 3, 1.5 * n)])  # tau = 3 and 1.5 per cell
    phi, theta, w = O.healpix_directions(2)
    uvb = np.array([1e-21, 1e-21])
    B = np.array([5e-21, 2e-21])
    engine.set_uniform_grid(n, 1.0)
    engine.set_opacity(kappa)
    it = SourceIteration(engine, nnu, n ** 3, phi, theta, w, uvb, eps, B)
    J_host = np.zeros((nnu, n ** 3))
    for k in range(3):
        it.step()
        S = (1 - eps) * J_host + eps * B[:, None]
        J_host = O.sweep_uniform(n, kappa, 1.0, phi, theta, w, uvb, src=S, arith=O.ARITH_DEVICE)
        assert np.allclose(it.J.cpu().numpy(), J_host, rtol=1e-13, atol=0), k
    hist = it.run(40, tol=1e-10)
    assert hist[-1] < 1e-10 and all(b <= a * 1.0001 for a, b in zip(hist[5:], hist[6:]))
    J = it.J.cpu().numpy().reshape(nnu, n, n, n)
    core = J[:, 6:10, 6:10, 6:10]
    assert np.allclose(core, B[:, None, None, None], rtol=2e-3)
    engine.set_source_function(None)


# ---- BASELINE size: properties that need no oracle run ------------------------------------------------------------

@pytest.fixture(scope="module")
def big(engine):
    import torch
    n, nnu = 256, 8
    kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=12345, tau_median=0.1)
    engine.set_uniform_grid(n, box)
    dev = torch.device("cuda", 0)
    k = torch.from_numpy(kappa).to(dev)
    engine.set_opacity_device(nnu, k.data_ptr())
    torch.cuda.synchronize()
    import radiativetransfer_amd as rt
    nside = 4
    ang = np.array([rt.pix2ang_nest(nside, i) for i in range(96)])
    return dict(n=n, nnu=nnu, kappa=kappa, uvb=uvb, box=box, phi=ang[:, 0].copy(), theta=ang[:, 1].copy(),
                w=np.full(96, 1 / 96), k_dev=k)


def _sweep_dev(engine, b, uvb, ndir=96):
    import torch
    J = torch.empty((b["nnu"], b["n"] ** 3), dtype=torch.float64, device="cuda:0")
    engine.transport_device(b["phi"][:ndir], b["theta"][:ndir], b["w"][:ndir], uvb, J.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return J


def test_baseline_size_deterministic_and_linear(engine, big):
    """256^3 x 8 groups x 96 directions: two runs agree bit for bit (every cell has one owner, no atomics), J is
    bounded by the inflow, and J is exactly linear in the inflow (scaling uvb by 2 is exact in binary64)."""
    import torch
    J1 = _sweep_dev(engine, big, big["uvb"])
    J2 = _sweep_dev(engine, big, big["uvb"])
    assert torch.equal(J1, J2)
    Jd = _sweep_dev(engine, big, 2.0 * big["uvb"])
    assert torch.equal(Jd, 2.0 * J1)
    cap = torch.from_numpy(big["uvb"]).to(J1.device)[:, None] * (1 + 1e-12)
    assert bool(torch.all(J1 > 0)) and bool(torch.all(J1 <= cap))


def test_baseline_size_slab_against_oracle(engine, big):
    """One direction at 256^3 against the oracle (device arithmetic), bit for bit, on two frequency groups."""
    n = big["n"]
    for d in (0, 57):
        phi, theta, w = big["phi"][d:d + 1], big["theta"][d:d + 1], big["w"][d:d + 1]
        import torch
        J = torch.empty((big["nnu"], n ** 3), dtype=torch.float64, device="cuda:0")
        engine.transport_device(phi, theta, w, big["uvb"], J.data_ptr(), 0)
        torch.cuda.synchronize()
        Jh = J[[0, 7]].cpu().numpy()
        ref = O.sweep_uniform(n, big["kappa"][[0, 7]], big["box"], phi, theta, w, big["uvb"][[0, 7]], arith=O.ARITH_DEVICE)
        assert np.array_equal(Jh, ref)


# ---- BASELINE configs[2] (192 directions) and the shapes an 8-rank run executes, at full size -------------------------------------

def _all_192(big):
    import radiativetransfer_amd as rt
    ang = np.array([rt.pix2ang_nest(4, i) for i in range(192)])
    return dict(big, phi=ang[:, 0].copy(), theta=ang[:, 1].copy(), w=np.full(192, 1 / 192))


def test_config3_192_directions_at_full_size(engine, big, organisation):
    """BASELINE configs[2]'s workload on one GPU: 256^3 x 8 groups x all 192 directions of nside 4 (equiSources.f90:1385-1391) --
    reproducible bit for bit, exactly linear in the inflow, bounded by it; and the first 96 directions of the same list, swept
    alone with the weights of the 192-set, plus the other 96 swept alone, add up to the whole (J = sum over directions,
    transportRoutinesModule.f90:953-955) to the rounding of the sum."""
    import torch
    if organisation == "tiles":
        pytest.skip("the full 192-direction sweep is run once, with the default organisation")
    b = _all_192(big)
    J1 = _sweep_dev(engine, b, b["uvb"], ndir=192)
    assert torch.equal(J1, _sweep_dev(engine, b, b["uvb"], ndir=192))
    assert torch.equal(_sweep_dev(engine, b, 2.0 * b["uvb"], ndir=192), 2.0 * J1)
    cap = torch.from_numpy(b["uvb"]).to(J1.device)[:, None] * (1 + 1e-12)
    assert bool(torch.all(J1 > 0)) and bool(torch.all(J1 <= cap))
    # a direction-split pair of ranks (2 x 96): their sum is the unsharded J
    Ja = _sweep_dev(engine, b, b["uvb"], ndir=96)
    second = dict(b, phi=b["phi"][96:], theta=b["theta"][96:], w=b["w"][96:])
    Jb = _sweep_dev(engine, second, b["uvb"], ndir=96)
    assert torch.allclose(Ja + Jb, J1, rtol=SUM_RTOL, atol=0)


def test_every_rank_share_of_an_eight_rank_run_at_full_size(engine, big, organisation):
    """What rank r of an N = 8 run of configs[2] executes (radiativetransfer_amd/distributed.py: Shard2D with 8 groups on 8 ranks):
    ONE frequency group, all 192 directions -- nnu_local = 1, so the planner takes short bricks, the pair form and direction
    groups dealt to streams.  Each of the eight shares equals the matching group of the unsharded sweep to the rounding of the sum
    over directions -- NOT bit for bit: frequency groups never meet inside the sweep and the direction groups are the same, but the
    order in which the (up to six) groups of a shared accumulator visit a cell follows the brick length, which the planner shortens
    for a single frequency group (measured: the first share already differs in last bits)."""
    import torch
    if organisation == "tiles":
        pytest.skip("the rank shapes are those of the default organisation")
    b = _all_192(big)
    J_all = _sweep_dev(engine, b, b["uvb"], ndir=192)
    for nu in range(b["nnu"]):
        engine.set_opacity_device(1, b["k_dev"][nu:nu + 1].contiguous().data_ptr())
        J = torch.empty((1, b["n"] ** 3), dtype=torch.float64, device="cuda:0")
        engine.transport_device(b["phi"], b["theta"], b["w"], b["uvb"][nu:nu + 1], J.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert engine.counter("brick_form") == 2            # the pair form: this is the narrow-stage shape
        assert torch.allclose(J[0], J_all[nu], rtol=SUM_RTOL, atol=0), nu
    engine.set_opacity_device(b["nnu"], b["k_dev"].data_ptr())   # back to the module's eight groups
    torch.cuda.synchronize()


def test_baseline_size_several_directions_against_oracle(engine, big):
    """Four directions at 256^3 on two frequency groups against the oracle (device arithmetic): three of one izone -- one pass of a
    direction group, sharing opacity loads, J stores and an accumulator with the fourth's group where the planner pairs them --
    and one of another izone.  To the rounding of the sum over directions (the planner adds in plan order)."""
    import torch
    n = big["n"]
    zone = np.array([O.fold_direction(p, t)[2] for p, t in zip(big["phi"], big["theta"])])
    zs, counts = np.unique(zone, return_counts=True)
    z0 = zs[np.argmax(counts >= 3)]
    first = np.flatnonzero(zone == z0)[:3]
    other = np.flatnonzero(zone != z0)[:1]
    pick = np.concatenate([first, other])
    phi, theta, w = big["phi"][pick], big["theta"][pick], big["w"][pick]
    J = torch.empty((big["nnu"], n ** 3), dtype=torch.float64, device="cuda:0")
    engine.transport_device(phi, theta, w, big["uvb"], J.data_ptr(), 0)
    torch.cuda.synchronize()
    Jh = J[[0, 7]].cpu().numpy()
    ref = O.sweep_uniform(n, big["kappa"][[0, 7]], big["box"], phi, theta, w, big["uvb"][[0, 7]], arith=O.ARITH_DEVICE)
    assert np.allclose(Jh, ref, rtol=SUM_RTOL, atol=0)
    # ... and against the arithmetic that shares nothing with the product (every segment in extended precision, rounded once):
    # 256 layers of segments along every ray, thickest and thinnest group
    # (two of the four directions, one of each izone: the extended-precision sweep of 256^3 cells takes its time on a host core)
    two = np.array([0, 3])
    engine.transport_device(phi[two], theta[two], w[two], big["uvb"], J.data_ptr(), 0)
    torch.cuda.synchronize()
    exact = O.sweep_uniform(n, big["kappa"][[0, 7]], big["box"], phi[two], theta[two], w[two], big["uvb"][[0, 7]], arith=O.ARITH_EXACT)
    assert np.all(np.abs(J[[0, 7]].cpu().numpy() - exact) <= 64 * EPS * exact)
