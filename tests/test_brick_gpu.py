"""The cell-fixed brick organisation of the uniform-grid sweep (csrc/ftte_brick.hip, option "engine" = 2) against the
oracle and the reference's goldens, through the C ABI.

Tolerances as in test_parity_gpu.py: bitwise against the oracle with the device arithmetic for a single direction (the
brick engine adds the directions of one izone group first and the groups in layout order, so several directions agree to
SUM_RTOL); the reference bound against the goldens.
"""
import numpy as np
import pytest

import _oracle as O
from radiativetransfer_amd import synthetic

pytestmark = pytest.mark.gpu

EPS = np.finfo(np.float64).eps
SUM_RTOL = 64 * EPS


@pytest.fixture(params=["solo", "pair"])
def bricks(engine, request):
    """Both forms of the brick kernel: one wavefront that takes the directions of a group in turn, and a pair of wavefronts
    with half the brick's rows each."""
    engine.set_option("engine", 2)
    engine.set_option("team", {"solo": 0, "pair": 2}[request.param])
    yield engine
    for key, value in (("engine", 0), ("team", -1), ("chunk", 0), ("group", 0), ("share", 2), ("lanes", 2)):
        engine.set_option(key, value)


def one_per_izone():
    phi, theta, _ = O.healpix_directions(3)
    pick = {}
    for p, t in zip(phi, theta):
        pick.setdefault(O.fold_direction(p, t)[2], (p, t))
    return [pick[z] for z in range(1, 25)]


@pytest.mark.parametrize("chunk", [1, 7, 32, 4096])
@pytest.mark.parametrize("n", [5, 16, 70, 130])
def test_every_izone_bitwise(bricks, n, chunk):
    """One direction per izone (24 rotations, all ray classes); grids smaller than a brick (5), one brick wide with a
    ragged last row block (70 = 8 * 8 + 6, 64 + 6 columns), three bricks wide (130); chunks of one layer, of a length
    that does not divide n, 32, and longer than the grid."""
    if n == 130 and chunk in (1, 7):
        pytest.skip("covered by the smaller grids")
    kappa, uvb, box = synthetic.uniform_workload(n, 2, seed=n, tau_median=0.3)
    bricks.set_option("chunk", chunk)
    bricks.set_uniform_grid(n, box)
    bricks.set_opacity(kappa)
    for p, t in one_per_izone():
        phi, theta, w = np.array([p]), np.array([t]), np.array([0.37])
        J = bricks.transport(phi, theta, w, uvb)
        ref = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
        assert np.array_equal(J, ref), f"izone {O.fold_direction(p, t)[2]}"


@pytest.mark.parametrize("group,share", [(1, 2), (3, 1), (8, 0), (2, 2)])
@pytest.mark.parametrize("name", ["uniform8_transparent", "uniform16_constant", "uniform16_lognormal_24zones",
                                  "uniform24_lognormal_48dir"])
def test_reference_goldens(bricks, golden, name, group, share):
    g = golden(name)
    n = int(g["n"])
    bricks.set_option("group", group)
    bricks.set_option("share", share)
    bricks.set_option("chunk", 5)
    bricks.set_uniform_grid(n, float(g["box"]))
    bricks.set_opacity(g["kappa"])
    J = bricks.transport(g["phi"], g["theta"], g["w"], g["uvb"])
    args = (n, g["kappa"], float(g["box"]), g["phi"], g["theta"], g["w"], g["uvb"])
    assert np.allclose(J, O.sweep_uniform(*args, arith=O.ARITH_DEVICE), rtol=SUM_RTOL, atol=0)
    _, noise = O.sweep_uniform(*args, with_noise=True)
    assert np.all(np.abs(J - g["J"]) <= 8 * noise + 12 * n * EPS * np.abs(g["J"]))


@pytest.mark.parametrize("n,chunk", [(100, 24), (128, 16)])
def test_bricks_equal_tiles_on_the_full_direction_set(engine, n, chunk):
    """192 directions, 3 groups; a grid of 2 x 13 x 5 ragged bricks, where only the passes of one izone share an
    accumulator, and one of 2 x 16 x 8 whole bricks, where pairs of izones share one as well: both organisations of the
    sweep, same arithmetic, the sums over directions in different orders."""
    kappa, uvb, box = synthetic.uniform_workload(n, 3, seed=11, tau_median=0.2)
    phi, theta, w = O.healpix_directions(3)
    engine.set_uniform_grid(n, box)
    engine.set_opacity(kappa)
    engine.set_option("engine", 1)
    J_tiles = engine.transport(phi, theta, w, uvb)
    engine.set_option("engine", 2)
    engine.set_option("chunk", chunk)
    engine.set_option("group", 3)
    J_bricks = engine.transport(phi, theta, w, uvb)
    J_again = engine.transport(phi, theta, w, uvb)
    engine.set_option("engine", 0)
    engine.set_option("chunk", 0)
    engine.set_option("group", 0)
    assert np.array_equal(J_bricks, J_again)  # no atomics, a fixed order: reproducible bit for bit
    assert np.allclose(J_bricks, J_tiles, rtol=SUM_RTOL, atol=0)
    assert 0 < J_bricks.min() and np.all(J_bricks <= uvb[:, None] * (1 + SUM_RTOL))


def test_one_wavefront_and_a_pair_per_brick_same_bits(engine):
    """The pair form splits a brick's rows between two wavefronts and changes nothing else: the same bits as one wavefront per
    brick, on whole and ragged grids, with shared accumulators (96 directions in groups of three), without emission and with
    either form of it."""
    phi, theta, w = O.healpix_directions(2)
    engine.set_option("engine", 2)
    try:
        for n in (64, 70):
            kappa, uvb, box = synthetic.uniform_workload(n, 3, seed=5, tau_median=0.2)
            engine.set_uniform_grid(n, box)
            engine.set_opacity(kappa)
            for emission in (None, "eta", "source"):
                if emission:
                    x = np.random.default_rng(n).uniform(0.1, 2.0, kappa.shape) * uvb[:, None] * (kappa if emission == "eta" else 1.0)
                    (engine.set_emissivity if emission == "eta" else engine.set_source_function)(x)
                J = {}
                for form in (0, 2):
                    engine.set_option("team", form)
                    J[form] = engine.transport(phi, theta, w, uvb)
                assert np.array_equal(J[0], J[2]), (n, emission)
                engine.set_emissivity(None)
    finally:
        engine.set_emissivity(None)
        engine.set_option("team", -1)
        engine.set_option("engine", 0)


@pytest.mark.parametrize("n", [64, 128])
def test_brick_order_and_frame_order_same_bits(engine, n):
    """On grids made of whole bricks the library can keep the opacities and the accumulators brick by brick (option tiled):
    another place for every number, the same numbers.  All 24 izones (every combination of mirrored axes and axis orders) one by
    one against the oracle, bit for bit; the full direction set with shared accumulators against the frame order, bit for bit, for
    one wavefront per brick and for a pair, with one launch per stage and with one launch for the sweep."""
    kappa, uvb, box = synthetic.uniform_workload(n, 3, seed=n, tau_median=0.25)
    engine.set_option("engine", 2)
    try:
        engine.set_uniform_grid(n, box)
        engine.set_opacity(kappa)
        if n == 64:
            for tiled in (1, 2):
                engine.set_option("tiled", tiled)
                for p, t in one_per_izone():
                    phi, theta, w = np.array([p]), np.array([t]), np.array([0.37])
                    J = engine.transport(phi, theta, w, uvb)
                    assert np.array_equal(J, O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE)), (tiled, O.fold_direction(p, t)[2])
        phi, theta, w = O.healpix_directions(2)
        for form, dataflow in ((0, 0), (2, 0), (0, 2), (0, 3)):
            engine.set_option("team", form)
            engine.set_option("dataflow", dataflow)
            J = {}
            for tiled in (1, 2, 0):
                engine.set_option("tiled", tiled)
                J[tiled] = engine.transport(phi, theta, w, uvb)
            assert np.array_equal(J[0], J[1]) and np.array_equal(J[0], J[2]), (form, dataflow)
        # new opacities: the brick-ordered copies follow
        engine.set_option("tiled", 1)
        engine.set_opacity(2.0 * kappa)
        J2 = engine.transport(phi, theta, w, uvb)
        engine.set_option("tiled", 0)
        assert np.array_equal(J2, engine.transport(phi, theta, w, uvb))
        assert not np.array_equal(J2, J[1])
    finally:
        for key, value in (("tiled", 0), ("team", -1), ("dataflow", 0), ("engine", 0)):
            engine.set_option(key, value)


def test_form_of_the_brick_kernel_follows_the_frequency_groups(engine):
    """Left to itself (option team = -1) the library sweeps with a pair of wavefronts per brick up to four frequency groups, with
    one above, and with the pair whenever there is emission."""
    n = 32
    phi, theta, w = O.healpix_directions(1)
    engine.set_option("engine", 2)
    try:
        for nnu, emission, form in ((2, False, 2), (4, False, 2), (5, False, 0), (8, False, 0), (8, True, 2)):
            kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=3, tau_median=0.2)
            engine.set_uniform_grid(n, box)
            engine.set_opacity(kappa)
            engine.set_source_function(np.full_like(kappa, 1e-22) if emission else None)
            J = engine.transport(phi, theta, w, uvb)
            assert engine.counter("brick_form") == form, (nnu, emission)
            src = {"src": np.full_like(kappa, 1e-22)} if emission else {}
            assert np.allclose(J, O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE, **src), rtol=SUM_RTOL, atol=0)
    finally:
        engine.set_source_function(None)
        engine.set_option("engine", 0)


def test_launch_records_account_for_every_update(bricks):
    n, nnu = 70, 2
    kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=2, tau_median=0.3)
    phi, theta, w = O.healpix_directions(2)
    bricks.set_uniform_grid(n, box)
    bricks.set_opacity(kappa)
    bricks.transport(phi, theta, w, uvb)
    rec = bricks.launch_records()
    assert sum(u for _, u in rec) == n ** 3 * nnu * len(phi)
    assert all(ms >= 0 for ms, _ in rec)


@pytest.mark.parametrize("lanes", [1, 3, 5])
def test_streams_over_frequency_groups_or_direction_groups(bricks, lanes):
    """The stage sequence is issued on several streams: over the frequency groups when there are enough of them (4 groups,
    3 streams), else over the groups of directions (1 frequency group: the accumulator-sharing sets are dealt to the
    streams).  Same bits whatever the number of streams."""
    n = 64
    phi, theta, w = O.healpix_directions(2)
    for nnu in (4, 1):
        kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=21, tau_median=0.2)
        bricks.set_uniform_grid(n, box)
        bricks.set_opacity(kappa)
        bricks.set_option("lanes", 1)
        J_one = bricks.transport(phi, theta, w, uvb)
        bricks.set_option("lanes", lanes)
        J_many = bricks.transport(phi, theta, w, uvb)
        assert np.array_equal(J_one, J_many)
        assert np.allclose(J_one, O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE), rtol=SUM_RTOL, atol=0)


def test_emission_through_an_opaque_source_free_slab_stays_finite(engine):
    """With emission switched on (the reference's log-mean (Iin-Iout)/log(Iin/Iout) is then evaluated from the intensities)
    a ray that crosses an opaque region without sources falls through the bottom of the normal range: intensities of
    1e-300 ... 1e-320 and then 0.  The device's division (reciprocal + Newton steps) must give what the host's `/` gives --
    finite, bit for bit -- instead of NaN from 1/subnormal (both operands are lifted by 2^200 first, csrc/ftte_math.h)."""
    n = 48
    tau_cell = np.full((n, n, n), 0.02)
    tau_cell[10:40] = 23.0                      # 30 opaque layers along storage-i: exp(-690) ~ 1e-300 and below
    kappa = np.ascontiguousarray((tau_cell * n).reshape(1, n ** 3))
    S = np.zeros_like(kappa)
    uvb = np.array([1.0])
    for engine_id in (1, 2):
        engine.set_option("engine", engine_id)
        engine.set_uniform_grid(n, 1.0)
        engine.set_opacity(kappa)
        engine.set_source_function(S)
        for p, t in one_per_izone()[::5]:
            phi, theta, w = np.array([p]), np.array([t]), np.array([1.0])
            J = engine.transport(phi, theta, w, uvb)
            assert np.all(np.isfinite(J)) and np.all(J >= 0)
            ref = O.sweep_uniform(n, kappa, 1.0, phi, theta, w, uvb, src=S, arith=O.ARITH_DEVICE)
            assert np.array_equal(J, ref)
        assert 0 < J[J > 0].min() < 1e-290      # the subnormal range was really crossed
        engine.set_source_function(None)
    engine.set_option("engine", 0)


@pytest.mark.parametrize("nnu,ndirs,group", [(2, 3, 3), (1, 2, 2), (3, 3, 4), (8, 2, 3), (16, 1, 3)])
def test_one_launch_with_flags_equals_a_launch_per_stage(engine, nnu, ndirs, group):
    """Grids of whole bricks (n a multiple of 64) are swept in ONE launch: a workgroup takes the next brick of the list, waits for
    the flags of the bricks it depends on (upstream neighbours, the previous writer of its J tile, the readers of the ring slots
    it reuses) and raises its own.  Same bits as the launch-per-stage form, sweep after sweep (the flags carry an epoch), with
    shared accumulators and several passes per izone."""
    n = 128
    phi, theta, w = O.healpix_directions(ndirs)
    kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=31 + nnu, tau_median=0.15)
    engine.set_option("engine", 2)
    engine.set_option("group", group)
    engine.set_uniform_grid(n, box)
    engine.set_opacity(kappa)
    engine.set_option("dataflow", 0)
    J_stages = engine.transport(phi, theta, w, uvb)
    for form in (1, 2, 3):   # 2: write-through stores instead of an L2 write-back before the flag; 3: persistent workgroups, a queue per XCD
        engine.set_option("dataflow", form)
        for _ in range(3):
            J_flags = engine.transport(phi, theta, w, uvb)
            assert np.array_equal(J_flags, J_stages), form
    for key, value in (("engine", 0), ("group", 0), ("dataflow", 0)):
        engine.set_option(key, value)
    one = (phi[:1], theta[:1], w[:1])
    assert np.array_equal(engine.transport(*one, uvb), O.sweep_uniform(n, kappa, box, *one, uvb, arith=O.ARITH_DEVICE))
