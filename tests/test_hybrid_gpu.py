"""Hybrid sweep of a refined cell array (csrc/ftte_hybrid.cpp: hybrid_sweep): the brick kernel outside a box around the refined cells,
the segment forest inside it, rays handed over through the bricks' face buffers.  Against the oracle's tree sweep (the reference's
setRaysRefined / findNeighbours / transport restated, pinned by the AMR goldens) with the device arithmetic -- bit for bit for a
single direction, to summation order for several -- and against the forest path for the whole tree (option "hybrid" = 0)."""
import numpy as np
import pytest

import _oracle as O
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic

pytestmark = pytest.mark.gpu

EPS = np.finfo(np.float64).eps
SUM_RTOL = 64 * EPS


def one_per_izone():
    phi, theta, _ = O.healpix_directions(3)
    pick = {}
    for p, t in zip(phi, theta):
        pick.setdefault(O.fold_direction(p, t)[2], (p, t))
    return [pick[z] for z in range(1, 25)]


def patch_case(n, blocks, depth, nnu, seed):
    level = synthetic.refine_levels(n, blocks, depth=depth)
    rho = synthetic.lognormal_density(len(level), seed=seed)
    _, s_nu, uvb = synthetic.frequency_groups(nnu)
    kappa = (0.15 * n) * s_nu[:, None] * rho[None, :] * (2.0 ** level)[None, :]
    return level, kappa, uvb


@pytest.mark.parametrize("n,blocks,depth", [
    (64, [(30 + a, 31 + b, 33 + c) for a in range(3) for b in range(2) for c in range(4)], 1),   # one level, off-centre block
    (72, [(40, 41, 20), (41, 41, 20), (40, 42, 21)], 2),                                           # two levels, ragged grid (72 = 64 + 8)
])
def test_hybrid_every_izone_bitwise(n, blocks, depth):
    level, kappa, uvb = patch_case(n, blocks, depth, 2, seed=n)
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        for p, t in one_per_izone():
            phi, theta, w = np.array([p]), np.array([t]), np.array([0.37])
            J = e.transport(phi, theta, w, uvb)
            ref = O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
            assert np.array_equal(J, ref), f"izone {O.fold_direction(p, t)[2]}"
        assert e.counter("forest_builds") >= 24          # the hybrid plan was taken (a restricted forest per direction list)


def test_hybrid_equals_whole_tree_forest_path():
    n = 64
    blocks = [(28 + a, 30 + b, 31 + c) for a in range(4) for b in range(4) for c in range(3)]
    level, kappa, uvb = patch_case(n, blocks, 1, 3, seed=5)
    phi, theta, w = O.healpix_directions(2)
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        J_hybrid = e.transport(phi, theta, w, uvb)
        J_again = e.transport(phi, theta, w, uvb)
        e.set_option("hybrid", 0)
        J_forest = e.transport(phi, theta, w, uvb)
    assert np.array_equal(J_hybrid, J_again)
    assert np.array_equal(J_forest, O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE))
    assert np.allclose(J_hybrid, J_forest, rtol=SUM_RTOL, atol=0)
    assert np.all(J_hybrid > 0) and np.all(J_hybrid <= uvb[:, None] * (1 + 1e-12))


@pytest.mark.parametrize("pipelines", [1, 2, 4])
def test_hybrid_pipelines_agree(pipelines):
    """The hybrid sweep runs as independent bricks - forests - bricks sequences on streams of their own (option "pipelines",
    default 3), meeting only in J through a fixed chain of events: whatever their number, J is reproducible run to run and the
    same to the rounding of the sum over directions."""
    n = 64
    blocks = [(20 + a, 30 + b, 31 + c) for a in range(3) for b in range(4) for c in range(3)]
    level, kappa, uvb = patch_case(n, blocks, 1, 2, seed=11)
    phi, theta, w = O.healpix_directions(2)
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        J_default = e.transport(phi, theta, w, uvb)
        e.set_option("pipelines", pipelines)
        J_a = e.transport(phi, theta, w, uvb)
        J_b = e.transport(phi, theta, w, uvb)
    assert np.array_equal(J_a, J_b)
    assert np.allclose(J_a, J_default, rtol=SUM_RTOL, atol=0)


def test_boxes_cut_through_bricks_or_end_on_brick_boundaries():
    """Along a brick's 64 lanes the boxes end one cell beyond the refined cells (option "box_lanes" = 1, default; 16: on the next
    multiple of 16) and the bricks they cut through sweep the lanes outside (brick_kernel<..., MASKED>: rays cross the box's
    u-faces inside a brick through two face rings of their own), or they end on brick boundaries (64).  A patch in the middle of
    a 128^3 row of two bricks leaves lanes on the near side of the first brick and on the far side of the second; one near the
    edge leaves both sides in one brick.  Same J to the rounding of the sum, bit for bit for a single direction."""
    n = 128
    for blocks in ([(60 + a, 70 + b, 66 + c) for a in range(3) for b in range(2) for c in range(8)],      # straddles lanes 63 | 64 of u = k
                   [(30 + a, 90 + b, 20 + c) for a in range(2) for b in range(2) for c in range(3)]):    # inside the first brick
        level, kappa, uvb = patch_case(n, blocks, 1, 2, seed=len(blocks))
        phi, theta, w = O.healpix_directions(2)
        with rt.DiffuseTransfer() as e:
            e.set_grid(n, level, 1.0)
            e.set_opacity(kappa)
            J16 = e.transport(phi, theta, w, uvb)
            one16 = [e.transport(phi[d:d + 1], theta[d:d + 1], w[d:d + 1], uvb) for d in (0, 17, 40)]
            e.set_option("box_lanes", 16)
            assert np.allclose(e.transport(phi, theta, w, uvb), J16, rtol=SUM_RTOL, atol=0)
            e.set_option("box_lanes", 64)
            J64 = e.transport(phi, theta, w, uvb)
            one64 = [e.transport(phi[d:d + 1], theta[d:d + 1], w[d:d + 1], uvb) for d in (0, 17, 40)]
        assert np.allclose(J16, J64, rtol=SUM_RTOL, atol=0)
        for a, b in zip(one16, one64):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("corner", [0, 1])
def test_refined_cells_on_the_domain_boundary(corner):
    """A refined patch in a corner of the grid: the box ends on the domain boundary on three sides (inflow enters the forest
    directly there, or its rays leave the grid) and cuts through the corner brick.  Every izone, single directions bit for bit
    against the forest path of the whole tree; all 24 together to the rounding of the sum."""
    n = 128
    base = 0 if corner == 0 else n - 3
    blocks = [(base + a, base + b, base + c) for a in range(3) for b in range(2) for c in range(3)]
    level, kappa, uvb = patch_case(n, blocks, 1, 2, seed=7 + corner)
    dirs = one_per_izone()
    phi, theta = np.array([d[0] for d in dirs]), np.array([d[1] for d in dirs])
    w = np.full(len(dirs), 1.0 / len(dirs))
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        J_all = e.transport(phi, theta, w, uvb)
        singles = [e.transport(phi[d:d + 1], theta[d:d + 1], w[d:d + 1], uvb) for d in range(0, 24, 5)]
        assert e.counter("forest_builds") >= 2
        e.set_option("hybrid", 0)
        J_forest = e.transport(phi, theta, w, uvb)
        for k, d in enumerate(range(0, 24, 5)):
            assert np.array_equal(singles[k], e.transport(phi[d:d + 1], theta[d:d + 1], w[d:d + 1], uvb)), f"direction {d}"
    assert np.allclose(J_all, J_forest, rtol=SUM_RTOL, atol=0)


def test_several_clusters_of_refined_cells_get_boxes_of_their_own():
    """Refined cells in clusters far apart: a box per cluster, the forests of boxes that lie behind other boxes swept in later
    passes with the bricks in between (three patches along a diagonal, so that for most izones each lies behind the one before, and
    two more side by side).  Against the forest path of the whole tree: single directions bit for bit, all together to the
    rounding of the sum; reproducible."""
    n = 128
    blocks = []
    for corner in [(20, 24, 18), (60, 58, 64), (100, 104, 98), (22, 100, 60), (100, 20, 64)]:
        blocks += [(corner[0] + a, corner[1] + b, corner[2] + c) for a in range(2) for b in range(3) for c in range(2)]
    level, kappa, uvb = patch_case(n, blocks, 1, 2, seed=13)
    dirs = one_per_izone()
    phi, theta = np.array([d[0] for d in dirs]), np.array([d[1] for d in dirs])
    w = np.full(len(dirs), 1.0 / len(dirs))
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        J_all = e.transport(phi, theta, w, uvb)
        assert e.counter("hybrid_boxes") == 5 and e.counter("hybrid_passes") >= 3
        assert np.array_equal(J_all, e.transport(phi, theta, w, uvb))
        singles = [e.transport(phi[d:d + 1], theta[d:d + 1], w[d:d + 1], uvb) for d in range(0, 24, 3)]
        e.set_option("hybrid", 0)
        J_forest = e.transport(phi, theta, w, uvb)
        assert e.counter("hybrid_boxes") == 0
        for k, d in enumerate(range(0, 24, 3)):
            assert np.array_equal(singles[k], e.transport(phi[d:d + 1], theta[d:d + 1], w[d:d + 1], uvb)), f"direction {d}"
    assert np.allclose(J_all, J_forest, rtol=SUM_RTOL, atol=0)


@pytest.mark.parametrize("which", ["source", "eta"])
def test_hybrid_with_emission(which):
    """A source function (the build's own form) or the reference's disabled emissivity term on a refined cell array:
    the hybrid sweep carries it through bricks (whole and cut), forests and the hand-overs between them.  Every izone bit for bit
    against the oracle's tree sweep with the same term; many directions against the forest path; radiative equilibrium (S = inflow)
    is a fixed point."""
    n = 64
    blocks = [(30 + a, 31 + b, 33 + c) for a in range(3) for b in range(2) for c in range(4)]
    level, kappa, uvb = patch_case(n, blocks, 1, 2, seed=17)
    rng = np.random.default_rng(4)
    X = rng.random(kappa.shape) * (2e-21 if which == "source" else 2e-21 * kappa.mean())
    kw = dict(src=X) if which == "source" else dict(eta=X)
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        (e.set_source_function if which == "source" else e.set_emissivity)(X)
        for p, t in one_per_izone()[::3]:
            phi, theta, w = np.array([p]), np.array([t]), np.array([0.37])
            J = e.transport(phi, theta, w, uvb)
            ref = O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE, **kw)
            assert np.array_equal(J, ref), f"izone {O.fold_direction(p, t)[2]}"
        assert e.counter("hybrid_boxes") == 1
        phi, theta, w = O.healpix_directions(2)
        J_hybrid = e.transport(phi, theta, w, uvb)
        e.set_option("hybrid", 0)
        J_forest = e.transport(phi, theta, w, uvb)
        assert np.allclose(J_hybrid, J_forest, rtol=SUM_RTOL, atol=0)
        if which == "source":
            e.set_option("hybrid", 1)
            e.set_source_function(np.repeat(uvb[:, None], len(level), 1))
            Jeq = e.transport(phi, theta, w, uvb)
            assert e.counter("hybrid_boxes") == 1
            assert np.allclose(Jeq, uvb[:, None] * w.sum(), rtol=64 * EPS, atol=0)


def test_launches_replayed_from_a_captured_graph():
    """Option "graph": the hybrid sweep's launches (three streams, forks and joins) captured once into a hipGraph and replayed while
    nothing they name changes; new opacities and a new J array go through.  Same bits as the launches issued one by one."""
    n = 128
    blocks = [(40 + a, 50 + b, 60 + c) for a in range(3) for b in range(3) for c in range(3)] + [(100 + a, 20, 90) for a in range(2)]
    level, kappa, uvb = patch_case(n, blocks, 1, 2, seed=21)
    phi, theta, w = O.healpix_directions(2)
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        J_plain = e.transport(phi, theta, w, uvb)
        e.set_opacity(0.5 * kappa)
        J_half = e.transport(phi, theta, w, uvb)
        e.set_option("graph", 1)
        e.set_opacity(kappa)
        J_captured = e.transport(phi, theta, w, uvb)       # captures
        J_replayed = e.transport(phi, theta, w, uvb)       # replays
        e.set_opacity(0.5 * kappa)
        J_half_replayed = e.transport(phi, theta, w, uvb)  # replays on new opacities
    assert np.array_equal(J_captured, J_plain) and np.array_equal(J_replayed, J_plain)
    assert np.array_equal(J_half_replayed, J_half)


def test_hybrid_more_directions_than_a_forest_batch():
    """192 directions, first all at once, then with forest batches capped at 40 directions (what a tree too large for the
    device memory gets): the pipelines can then not run side by side (their scratch would overlap) and take turns on one stream
    instead, their forests in batches.  Same J either way, and as the forest path of the whole tree."""
    n = 64
    blocks = [(30 + a, 29 + b, 33 + c) for a in range(2) for b in range(3) for c in range(2)]
    level, kappa, uvb = patch_case(n, blocks, 1, 2, seed=3)
    phi, theta, w = O.healpix_directions(3)
    assert len(phi) == 192
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        J_hybrid = e.transport(phi, theta, w, uvb)
        assert np.array_equal(J_hybrid, e.transport(phi, theta, w, uvb))
        e.set_option("forest_batch", 40)
        J_batched = e.transport(phi, theta, w, uvb)
        assert np.array_equal(J_batched, e.transport(phi, theta, w, uvb))
        e.set_option("hybrid", 0)
        J_forest = e.transport(phi, theta, w, uvb)
    assert np.allclose(J_hybrid, J_forest, rtol=SUM_RTOL, atol=0)
    assert np.allclose(J_batched, J_forest, rtol=SUM_RTOL, atol=0)


def test_slot_lists_with_batched_forests_and_several_pipelines():
    """An off-centre patch (box depths differ between izones, so the pipelines' places for the forest pass differ in slot form),
    option "hybrid_slots" = 2 (slots even with one pass), three pipelines and forest batches smaller than the direction list: the
    pipelines' forests then share one run on one stream, which has one place in the launch sequence -- the sweep must fall back to
    the phase form (or the forest path) rather than launch them at pipeline 0's place.  Against the forest path of the whole tree,
    single directions bit for bit."""
    n = 128
    blocks = [(18 + a, 90 + b, 40 + c) for a in range(2) for b in range(3) for c in range(2)]
    level, kappa, uvb = patch_case(n, blocks, 1, 2, seed=29)
    dirs = one_per_izone()
    phi, theta = np.array([d[0] for d in dirs]), np.array([d[1] for d in dirs])
    w = np.full(len(dirs), 1.0 / len(dirs))
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        e.set_option("hybrid_slots", 2)
        e.set_option("pipelines", 3)
        J_slots = e.transport(phi, theta, w, uvb)               # slot lists, every direction resident
        e.set_option("forest_batch", 5)
        J_batched = e.transport(phi, theta, w, uvb)
        assert np.array_equal(J_batched, e.transport(phi, theta, w, uvb))
        e.set_option("hybrid", 0)
        J_forest = e.transport(phi, theta, w, uvb)
    assert np.allclose(J_slots, J_forest, rtol=SUM_RTOL, atol=0)
    assert np.allclose(J_batched, J_forest, rtol=SUM_RTOL, atol=0)


def test_small_trees_stay_on_the_forest_path(golden):
    """Where the box around the refined cells takes up most of the grid (the AMR goldens: 8^3 and 6^3) nothing is left for the bricks:
    the whole tree goes through the forest, bit for bit as before."""
    g = golden("amr8_block_level1")
    with rt.DiffuseTransfer() as e:
        e.set_grid(int(g["n"]), g["level"], float(g["box"]))
        e.set_opacity(g["kappa"])
        J = e.transport(g["phi"], g["theta"], g["w"], g["uvb"])
    assert np.array_equal(J, O.sweep_tree(int(g["n"]), g["level"], g["kappa"], float(g["box"]), g["phi"], g["theta"], g["w"], g["uvb"],
                                          arith=O.ARITH_DEVICE))


def _cube_case(n, q, lo, nnu, seed):
    blocks = [(lo[0] + a, lo[1] + b, lo[2] + c) for a in range(q) for b in range(q) for c in range(q)]
    return patch_case(n, blocks, 1, nnu, seed)


@pytest.mark.parametrize("lo", [(16, 16, 16), (8, 24, 12)])
def test_fine_block_swept_by_bricks_of_its_own_every_izone_bitwise(lo):
    """A fully refined block -- a cube of 32^3 base cells refined exactly once -- is a uniform grid of 64^3 fine cells with a ray
    pattern per sub-layer (setRaysRefined, transportRoutinesModule.f90:150-187): option "fine_bricks" sweeps it with bricks of its own
    on the fine level and leaves to the segment forest only what lies around it; rays cross between the two through the fine bricks'
    face rings, coarse to fine by the reference's rules (the piece that ends on the shared face, or the mean of two, :612-634), fine
    to coarse from the one fine cell that holds the entry point.  One direction per izone against the oracle's tree sweep, bit for
    bit, the block in the middle of the grid and off centre."""
    n = 64
    level, kappa, uvb = _cube_case(n, 32, lo, 2, seed=41)
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        for p, t in one_per_izone():
            phi, theta, w = np.array([p]), np.array([t]), np.array([0.37])
            J = e.transport(phi, theta, w, uvb)
            assert e.counter("fine_block") == 64
            ref = O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
            assert np.array_equal(J, ref), f"izone {O.fold_direction(p, t)[2]}"


def test_fine_block_many_directions_pipelines_and_the_forest_for_it():
    """The same with 48 directions in three pipelines: reproducible, equal to the sweep that keeps the block in the forest
    (fine_bricks = 0) and to the forest path for the whole tree to the rounding of the sum; a block the bricks cannot take (30^3:
    twice its side is no multiple of 64) stays in the forest."""
    n = 64
    level, kappa, uvb = _cube_case(n, 32, (16, 16, 16), 3, seed=43)
    phi, theta, w = O.healpix_directions(2)
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        J = e.transport(phi, theta, w, uvb)
        assert e.counter("fine_block") == 64 and e.counter("hybrid_boxes") == 1
        assert np.array_equal(J, e.transport(phi, theta, w, uvb))
        for pipelines in (1, 2):
            e.set_option("pipelines", pipelines)
            assert np.allclose(e.transport(phi, theta, w, uvb), J, rtol=SUM_RTOL, atol=0)
        e.set_option("pipelines", 3)
        e.set_option("fine_bricks", 0)
        J_forest_block = e.transport(phi, theta, w, uvb)
        assert e.counter("fine_block") == 0 and e.counter("hybrid_boxes") == 1
        e.set_option("hybrid", 0)
        J_forest = e.transport(phi, theta, w, uvb)
    assert np.allclose(J, J_forest_block, rtol=SUM_RTOL, atol=0)
    assert np.allclose(J, J_forest, rtol=SUM_RTOL, atol=0)
    assert np.all(J > 0) and np.all(J <= uvb[:, None] * (1 + 1e-12))
    level, kappa, uvb = _cube_case(n, 30, (17, 17, 17), 2, seed=44)
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        J = e.transport(phi, theta, w, uvb)
        assert e.counter("fine_block") == 0
        e.set_option("hybrid", 0)
        assert np.allclose(J, e.transport(phi, theta, w, uvb), rtol=SUM_RTOL, atol=0)


@pytest.mark.parametrize("which", ["source", "eta"])
def test_fine_block_with_emission(which):
    """A source function or the reference's emissivity term through the fine block's own bricks (their rows gathered and transposed
    like the opacities): every third izone bit for bit against the oracle's tree sweep with the same term; in radiative equilibrium
    (S = inflow) J = inflow to the rounding of the sum, fine cells and coarse cells alike."""
    n = 64
    level, kappa, uvb = _cube_case(n, 32, (16, 16, 16), 2, seed=47)
    rng = np.random.default_rng(5)
    X = rng.random(kappa.shape) * (2e-21 if which == "source" else 2e-21 * kappa.mean())
    kw = dict(src=X) if which == "source" else dict(eta=X)
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        (e.set_source_function if which == "source" else e.set_emissivity)(X)
        for p, t in one_per_izone()[::3]:
            phi, theta, w = np.array([p]), np.array([t]), np.array([0.37])
            J = e.transport(phi, theta, w, uvb)
            assert e.counter("fine_block") == 64
            assert np.array_equal(J, O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE, **kw)), f"izone {O.fold_direction(p, t)[2]}"
        if which == "source":
            phi, theta, w = O.healpix_directions(2)
            e.set_source_function(np.repeat(uvb[:, None], len(level), 1))
            Jeq = e.transport(phi, theta, w, uvb)
            assert np.allclose(Jeq, uvb[:, None] * w.sum(), rtol=64 * EPS, atol=0)


@pytest.mark.parametrize("nnu", [1, 3, 8])
def test_thin_forest_levels_in_one_launch(nnu):
    """Option "forest_fuse": runs of thin levels of the segment forests go in one launch, a workgroup per direction and a barrier
    per level (amr_levels_kernel), thick levels keep a launch of their own.  Same arithmetic, same order: the bits of a launch per
    level -- the forest path of the whole tree (every level thin on a small tree, thin and thick ones mixed at the default bound
    on a larger one), the hybrid sweep, with and without a fine block -- and the whole-tree path still equals the oracle."""
    n = 64
    blocks = [(28 + a, 30 + b, 31 + c) for a in range(4) for b in range(4) for c in range(3)]
    level, kappa, uvb = patch_case(n, blocks, 2, nnu, seed=17)
    phi, theta, w = O.healpix_directions(2)
    phi, theta, w = phi[::4], theta[::4], w[::4]
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        J = {}
        for hybrid in (1, 0):
            e.set_option("hybrid", hybrid)
            for fuse in (0, 4096, 1 << 24, 600):
                e.set_option("forest_fuse", fuse)
                J[hybrid, fuse] = e.transport(phi, theta, w, uvb)
            for fuse in (4096, 1 << 24, 600):
                assert np.array_equal(J[hybrid, fuse], J[hybrid, 0]), (hybrid, fuse)
        assert np.array_equal(J[0, 0], O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE))
    level, kappa, uvb = _cube_case(n, 32, (16, 16, 16), nnu, seed=43)
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        e.set_option("forest_fuse", 0)
        J0 = e.transport(phi, theta, w, uvb)
        assert e.counter("fine_block") == 64
        for fuse in (4096, 30000):
            e.set_option("forest_fuse", fuse)
            assert np.array_equal(e.transport(phi, theta, w, uvb), J0), fuse


def test_refined_cell_arrays_within_64_eps_of_the_exact_evaluation():
    """The hybrid sweep with a fine block, the hybrid sweep of scattered two-level patches and the forest path of the whole tree
    against the oracle's arithmetic that shares nothing with the product (every segment in extended precision, rounded once;
    tests/test_exact_arithmetic.py): J within 64 eps cell by cell, coarse and fine cells alike."""
    phi, theta, w = O.healpix_directions(2)
    phi, theta, w = phi[::4], theta[::4], w[::4]
    n = 64
    cases = [_cube_case(n, 32, (16, 8, 20), 2, seed=91),
             patch_case(n, [(30 + a, 31 + b, 33 + c) for a in range(3) for b in range(2) for c in range(4)] + [(5, 50, 9)], 2, 2, seed=92)]
    for k, (level, kappa, uvb) in enumerate(cases):
        exact = O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_EXACT)
        with rt.DiffuseTransfer() as e:
            e.set_grid(n, level, 1.0)
            e.set_opacity(kappa)
            J = e.transport(phi, theta, w, uvb)
            assert (e.counter("fine_block") == 64) == (k == 0)
            e.set_option("hybrid", 0)
            J_forest = e.transport(phi, theta, w, uvb)
        assert np.all(np.abs(J - exact) <= 64 * EPS * exact), k
        assert np.all(np.abs(J_forest - exact) <= 64 * EPS * exact), k
