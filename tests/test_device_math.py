"""radiativetransfer_amd/csrc/ftte_math.h evaluated on the host (same source the kernel compiles): accuracy of
the attenuation pair and exactness of the division-free cell mean."""
import mpmath as mp
import numpy as np

import _oracle as O

EPS = np.finfo(np.float64).eps


def test_attenuation_accuracy():
    rng = np.random.default_rng(1)
    tau = np.concatenate([10 ** rng.uniform(-12, 2.8, 4000), rng.uniform(0, 2, 2000), [0.0, 1e-300, 0.34657, 0.34658, 745.0, 800.0, 1e9]])
    e, g = O.device_attenuation(tau)
    mp.mp.dps = 40
    worst_e = worst_g = 0.0
    for t, ee, gg in zip(tau, e, g):
        te = mp.e ** (-mp.mpf(float(t)))
        tg = -mp.expm1(-mp.mpf(float(t))) / mp.mpf(float(t)) if t > 0 else mp.mpf(1)
        if te > mp.mpf("1e-290"):
            worst_e = max(worst_e, float(abs(mp.mpf(float(ee)) / te - 1)))
        else:
            assert ee < 1e-289
        worst_g = max(worst_g, float(abs(mp.mpf(float(gg)) / tg - 1)))
    assert worst_e < 2 * EPS, worst_e
    assert worst_g < 3 * EPS, worst_g
    assert e[-1] == 0.0 and g[-1] == 1e-9  # huge tau: exp underflows to 0, g = 1/tau


def test_attenuation_monotone_and_bounded():
    tau = np.linspace(0, 50, 20001)
    e, g = O.device_attenuation(tau)
    assert e[0] == 1.0 and g[0] == 1.0
    assert np.all(np.diff(e) <= 0) and np.all((e >= 0) & (e <= 1))
    assert np.all((g > 0) & (g <= 1))


def test_cell_mean_equals_ieee_division():
    rng = np.random.default_rng(2)
    acc = rng.lognormal(0, 3, 200000) * 1e-21
    for nseg in (1, 2, 3):
        assert np.array_equal(O.device_cell_mean(acc, nseg, 0.013), acc / nseg * 0.013)


def test_log1p_accuracy():
    """ftte_log1p (used as log(Iin/Iout) = log1p((Iin-Iout)/Iout) in the emission path), through log(x) = log1p(x-1)."""
    rng = np.random.default_rng(3)
    x = np.concatenate([1 + 10 ** rng.uniform(-15.5, 0, 3000), 10 ** rng.uniform(0, 300, 3000), [2.0, np.sqrt(2), 1e308]])
    y = O.device_log(x)
    mp.mp.dps = 40
    worst = 0.0
    for a, b in zip(x, y):
        t = mp.log1p(mp.mpf(float(a)) - 1)
        worst = max(worst, float(abs(mp.mpf(float(b)) / t - 1)))
    assert worst < 2.5 * EPS, worst
    assert O.device_log(np.array([1.0]))[0] == 0.0
