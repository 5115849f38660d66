"""BASELINE.json configs[3] and configs[4] at their full sizes on one GPU, through size-independent properties (the oracle
finishes only slabs of them in seconds):

  configs[3]  128^3 base grid, central 32^3 block refined once (2 326 528 leaves), 8 frequency groups, 96 directions, one star
              at the centre of the patch: J reproducible bit for bit, 0 < J <= inflow, a two-direction slab bit for bit
              against the oracle's tree sweep, photon conservation of the point source.
  configs[4]  256^3, 8 groups, 96 directions, source iterations S = (1 - eps) J + eps B on a plane-parallel stratification
              (one GPU's worth: the 8-GPU run shards these same sweeps): residuals fall monotonically like an unaccelerated
              Lambda iteration, J <= max(inflow, B), J -> B deep in the thick layers.
"""
import numpy as np
import pytest

import _oracle as O
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic

pytestmark = pytest.mark.gpu

EPS = np.finfo(np.float64).eps


def bench_directions(count=96):
    ang = np.array([rt.pix2ang_nest(4, i) for i in range(count)])
    return ang[:, 0].copy(), ang[:, 1].copy(), np.full(count, 1.0 / count)


@pytest.fixture(scope="module")
def config3():
    n = 128
    q = n // 4
    lo = n // 2 - q // 2
    blocks = [(lo + a, lo + b, lo + c) for a in range(q) for b in range(q) for c in range(q)]
    level = synthetic.refine_levels(n, blocks, depth=1)
    assert len(level) == 2326528
    box = 3.0e22
    nnu = 8
    rho = synthetic.lognormal_density(len(level), seed=4)
    _, s_nu, uvb = synthetic.frequency_groups(nnu)
    kappa = (0.1 * n / box) * s_nu[:, None] * rho[None, :] * (2.0 ** level)[None, :]
    eng = rt.StellarTransfer(device=0)
    eng.set_grid(n, level, box)
    eng.set_opacity(kappa)
    yield dict(n=n, level=level, box=box, nnu=nnu, kappa=kappa, uvb=uvb, eng=eng)
    eng.close()


def test_config3_diffuse_full_size(config3):
    c, eng = config3, config3["eng"]
    phi, theta, w = bench_directions()
    J1 = eng.transport(phi, theta, w, c["uvb"])
    J2 = eng.transport(phi, theta, w, c["uvb"])
    assert eng.counter("forest_builds") == 1          # the second sweep reuses the 96 segment forests
    assert eng.counter("fine_block") == 64            # the refined 32^3 block is swept by bricks of its own on the fine level
    assert np.array_equal(J1, J2)                     # directions are added in list order, no atomics
    eng.set_option("fine_bricks", 0)                  # ... and with the block left in the segment forest: the same J to the rounding of the sum
    J3 = eng.transport(phi, theta, w, c["uvb"])
    eng.set_option("fine_bricks", 1)
    assert eng.counter("fine_block") == 0
    assert np.allclose(J1, J3, rtol=64 * np.finfo(float).eps, atol=0)
    assert J1.shape == (c["nnu"], len(c["level"]))
    assert np.all(J1 > 0) and np.all(J1 <= c["uvb"][:, None] * (1 + 1e-12))
    # the refined patch is denser per unit length (kappa doubles with the level): on average darker than the base grid around it
    fine = c["level"] == 1
    assert J1[0, fine].mean() < J1[0, ~fine].mean()


def test_config3_two_direction_slab_against_the_oracle(config3):
    """Two directions (a refined-patch entry from two different izones), three of the eight groups: the oracle's tree sweep
    (the reference's findNeighbours / setRaysRefined / transport restated, pinned by the AMR goldens) with the device
    arithmetic, bit for bit."""
    c, eng = config3, config3["eng"]
    phi, theta, w = bench_directions()
    pick = [3, 70]
    sel = [0, 4, 7]
    J = eng.transport(phi[pick], theta[pick], w[pick], c["uvb"])
    assert eng.counter("fine_block") == 64            # (the refined block went through the fine bricks)
    ref = O.sweep_tree(c["n"], c["level"], c["kappa"][sel], c["box"], phi[pick], theta[pick], w[pick], c["uvb"][sel], arith=O.ARITH_DEVICE)
    assert np.array_equal(J[sel], ref)


def test_config3_point_source_conserves_photons(config3):
    """The Stromgren-sphere set-up: one star in the centre of the refined patch, homogeneous hydrogen of optical depth 6 across
    the box.  What the cells absorb is what the star emits minus what reaches the boundary: more in a denser medium, all of it
    in an opaque one, never more than was emitted; and every lit cell has a non-negative rate."""
    c, eng = config3, config3["eng"]
    n, box, ncell = c["n"], c["box"], len(c["level"])
    pop = synthetic.stellar_population()
    eng.stellar_beta_table(*pop, 3, 0.4, 2, 0.3)
    tau_box = 6.0
    HI = np.full(ncell, tau_box / (6.3e-18 * box))
    HeI, HeII = 1e-10 * HI, 1e-10 * HI   # hydrogen only: every absorbed photon is counted by krate24
    rho, abun2 = HI * 1.67e-24 / 0.76, np.full(ncell, 0.02)
    src = eng.locate_cell([n // 2, n // 2, n // 2, 2, 2, 2])
    weight = 1000.0
    emitted = eng.rate_tables()[0, 0, 0, 0, 0] * weight
    absorbed = []
    for scale in (1.0, 2.0, 100.0):   # tau_box = 6, 12, 600 at the hydrogen threshold (harder photons see less)
        eng.set_medium(scale * HI, scale * HeI, scale * HeII, scale * rho, abun2, 0)
        eng.set_zero_rates()
        highest = eng.point_sources([src], [weight])
        k = eng.rates()
        assert 1 <= highest <= 6
        # the helium deposits are differences of nearly equal table values here (no helium): rounding noise around zero
        assert np.all(k[0] >= 0) and np.all(k >= -1e-9 * np.abs(k).max())
        absorbed.append(k[0].sum() / emitted)
    assert 0.5 < absorbed[0] < absorbed[1] < absorbed[2] <= 1 + 1e-9   # never more than was emitted
    assert absorbed[2] > 0.999                                          # an opaque box keeps everything
    eng.set_zero_rates()
    assert not eng.rates().any()


@pytest.mark.parametrize("engine_id", [2, 1])
def test_config4_source_iterations_full_size(engine_id):
    import torch
    from radiativetransfer_amd.iteration import SourceIteration
    n, nnu, eps = 256, 8, 1e-2
    _, s_nu, _ = synthetic.frequency_groups(nnu)
    z = (np.arange(n) + 0.5) / n
    tau_cell = 10.0 ** (-2.0 + 3.0 * z)           # plane-parallel: tau per cell 0.01 ... 10 along storage-i
    kappa_host = np.ascontiguousarray(((tau_cell * n)[None, :, None, None] * s_nu[:, None, None, None]
                                       * np.ones((1, 1, n, n))).reshape(nnu, n ** 3))
    phi, theta, w = bench_directions()
    inflow = np.full(nnu, 1e-30)
    B = 1e-21 * s_nu ** 0.5
    kappa = torch.from_numpy(kappa_host).to("cuda:0")
    with rt.DiffuseTransfer(device=0) as eng:
        eng.set_option("engine", engine_id)
        eng.set_uniform_grid(n, 1.0)
        eng.set_opacity_device(nnu, kappa.data_ptr())
        it = SourceIteration(eng, nnu, n ** 3, phi, theta, w, inflow, eps, B)
        hist = it.run(3 if engine_id == 1 else 6)
        J = it.J
        assert bool(torch.isfinite(J).all())
        # an unaccelerated Lambda iteration from J = 0: the relative change goes like 1/k
        assert hist[0] == pytest.approx(1.0, abs=1e-12)
        assert all(b < a for a, b in zip(hist, hist[1:]))
        assert hist[1] == pytest.approx(0.5, rel=0.05) and hist[2] == pytest.approx(1 / 3, rel=0.08)
        cap = torch.from_numpy(np.maximum(inflow, B)).to(J.device)[:, None] * (1 + 1e-12)
        assert bool(torch.all(J >= 0)) and bool(torch.all(J <= cap))
        # thick layers: J = (1 - (1-eps)^k) B after k iterations from J = 0 (every sweep returns S there)
        k = len(hist)
        deep = J.reshape(nnu, n, n, n)[0, -8:, n // 2, n // 2].cpu().numpy()
        assert np.allclose(deep, (1 - (1 - eps) ** k) * B[0], rtol=2e-3)
