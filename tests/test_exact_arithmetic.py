"""An arithmetic yardstick that shares nothing with the product: the oracle's ARITH_EXACT evaluates every segment of the reference's
algorithm in extended precision (long double: expl, expm1l) and rounds once -- for a sweep without emission the reference's log-mean
(Iin - Iout)/log(Iin/Iout) IS Iin (1 - exp(-tau))/tau (transportRoutinesModule.f90:651-678) --, so it says how far from the value of
the reference's formulae each double-precision evaluation ends up:

  * the device arithmetic (radiativetransfer_amd/csrc/ftte_math.h, here through the oracle's ARITH_DEVICE, bit for bit what the GPU
    computes: tests/test_parity_gpu.py) stays within 32 eps of it, in thin cells and behind 120 e-folds of attenuation alike;
  * the reference's own evaluation (libm exp and log in double) is off by up to 10^7 eps in thin cells -- the noise of its
    logarithm of a ratio near one, which is what the oracle's `noise` estimate bounds and what the goldens' tolerance allows for.

So the tolerance the reference-anchored goldens need is the reference's, not the device's."""
import numpy as np
import pytest

import _oracle as O
from radiativetransfer_amd import synthetic

EPS = np.finfo(np.float64).eps


@pytest.mark.parametrize("n,nnu,tau_median", [(32, 3, 0.01), (32, 3, 0.1), (24, 2, 3.0)])
def test_device_arithmetic_within_32_eps_of_the_exact_evaluation_uniform(n, nnu, tau_median):
    kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=3, tau_median=tau_median)
    phi, theta, w = O.healpix_directions(2)
    phi, theta, w = phi[::3], theta[::3], w[::3]
    exact = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_EXACT)
    device = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
    reference, noise = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_REFERENCE, with_noise=True)
    assert np.all(exact > 0)
    assert np.all(np.abs(device - exact) <= 32 * EPS * exact)
    assert np.all(np.abs(reference - exact) <= noise + 64 * EPS * exact)      # the reference is where its noise estimate says
    if tau_median <= 0.1:                                                     # ... and that is far: thin cells, log of a ratio near 1
        assert np.max(np.abs(reference - exact) / exact) > 1e4 * EPS > 100 * np.max(np.abs(device - exact) / exact)


def test_device_arithmetic_within_32_eps_of_the_exact_evaluation_with_a_source_function_and_on_a_tree():
    n, nnu = 24, 2
    kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=5, tau_median=0.3)
    phi, theta, w = O.healpix_directions(2)
    phi, theta, w = phi[1::4], theta[1::4], w[1::4]
    S = np.random.default_rng(1).random(kappa.shape) * uvb[:, None]
    exact = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, src=S, arith=O.ARITH_EXACT)
    device = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, src=S, arith=O.ARITH_DEVICE)
    assert np.all(np.abs(device - exact) <= 32 * EPS * exact)
    # the reference's emissivity term (:656-676): its formulae in extended precision, the log-mean through log1p
    X = np.random.default_rng(2).random(kappa.shape) * uvb[:, None] * kappa.mean()
    exact = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, eta=X, arith=O.ARITH_EXACT)
    device = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, eta=X, arith=O.ARITH_DEVICE)
    reference, noise = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, eta=X, with_noise=True)
    assert np.all(np.abs(device - exact) <= 32 * EPS * exact)
    assert np.all(np.abs(reference - exact) <= noise + 64 * EPS * exact)
    # a refined cell array: two levels, the patterns of the sub-layers and the mean-of-two hand-over in play
    n = 16
    level = synthetic.refine_levels(n, [(7, 7, 7), (8, 8, 7), (3, 12, 5)], depth=2)
    rho = synthetic.lognormal_density(len(level), seed=2)
    _, s_nu, uvb = synthetic.frequency_groups(nnu)
    kappa = (0.4 * n) * s_nu[:, None] * rho[None, :] * (2.0 ** level)[None, :]
    exact = O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_EXACT)
    device = O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
    reference, noise = O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_REFERENCE, with_noise=True)
    assert np.all(np.abs(device - exact) <= 32 * EPS * exact)
    assert np.all(np.abs(reference - exact) <= noise + 64 * EPS * exact)


@pytest.mark.parametrize("name", ["uniform16_constant", "uniform16_lognormal_24zones", "uniform24_lognormal_48dir", "amr8_block_level1",
                                  "amr6_scattered_level2"])
def test_the_reference_s_own_output_and_the_device_arithmetic_around_the_exact_value(golden, name):
    """The golden J of these files is what the reference's compiled code returned (tests/golden/make_golden.py).  Measured against the
    exact evaluation of the same inputs: the reference's output lies within its noise bound of it, the device arithmetic within 32 eps
    -- the device is the closer of the two to what the reference's formulae mean."""
    g = golden(name)
    n = int(g["n"])
    args = (g["kappa"], float(g["box"]), g["phi"], g["theta"], g["w"], g["uvb"])
    exact = O.sweep_tree(n, g["level"], *args, arith=O.ARITH_EXACT)
    device = O.sweep_tree(n, g["level"], *args, arith=O.ARITH_DEVICE)
    _, noise = O.sweep_tree(n, g["level"], *args, with_noise=True)
    assert np.all(np.abs(device - exact) <= 32 * EPS * exact)
    assert np.all(np.abs(g["J"] - exact) <= noise + 64 * EPS * exact)
    assert np.max(np.abs(device - exact) / exact) <= np.max(np.abs(g["J"] - exact) / exact)
