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


def test_log_mean_with_emission_accuracy():
    """ftte_segment_emit's path mean (Iin-Iout)/log(Iin/Iout) from the intensities it produced itself: the logarithm-free
    form below Iin/Iout = sqrt(2), the general one above, continuous across the switch, exact limits at the ends."""
    rng = np.random.default_rng(8)
    n = 6000
    Iin = 10 ** rng.uniform(-24, -18, n)
    S = Iin * 10 ** rng.uniform(-6, 1, n)                 # source function relative to the incoming intensity
    tau = 10 ** rng.uniform(-9, 1.5, n)
    # near equilibrium (the case the reference's own quotient form loses) and across the switch at Iin/Iout = sqrt(2)
    S[:1500] = Iin[:1500] * (1 + rng.uniform(-1, 1, 1500) * 10 ** rng.uniform(-14, -2, 1500))
    tau[1500:2500] = -np.log(1 / np.sqrt(2)) * (1 + rng.uniform(-1e-3, 1e-3, 1000))
    S[1500:2500] = 0.0
    Iout, mean = O.device_segment_emit(Iin, tau, S * tau, 0.0)   # the reference's emissivity term with eta = S tau: Iout = Iin e + S tau g
    mp.mp.dps = 50
    worst = 0.0
    for a, b, m in zip(Iin, Iout, mean):
        a, b = mp.mpf(float(a)), mp.mpf(float(b))
        true = (a - b) / mp.log(a / b) if b < a else (a + b) / 2
        if a == b:
            true = a
        worst = max(worst, float(abs(mp.mpf(float(m)) / true - 1)))
    assert worst < 3 * EPS, worst
    # in equilibrium nothing changes, and the mean is the intensity itself
    Iout, mean = O.device_segment_emit(np.array([3e-21]), np.array([0.7]), np.array([0.7 * 3e-21]), 0.0)
    assert abs(Iout[0] / 3e-21 - 1) < 4 * EPS and abs(mean[0] / 3e-21 - 1) < 4 * EPS
    # complete extinction without a source: zero out, zero mean (the reference's (Iin-0)/log(Iin/0))
    Iout, mean = O.device_segment_emit(np.array([3e-21]), np.array([2000.0]), 0.0, np.array([0.0]))
    assert Iout[0] == 0.0 and mean[0] == 0.0


def test_source_function_segment_is_the_exact_path_mean():
    """ftte_segment_source: with a source function S constant along the piece, I(t) = S + (Iin - S) exp(-t); Iout is its end and the
    cell's share its exact mean S + (Iin - S)(1 - exp(-tau))/tau -- for S = 0 bit for bit what ftte_segment gives (the reference's
    log-mean IS that mean then), in equilibrium exactly S."""
    rng = np.random.default_rng(11)
    n = 6000
    Iin = 10 ** rng.uniform(-24, -18, n)
    S = Iin * 10 ** rng.uniform(-6, 2, n)
    tau = 10 ** rng.uniform(-9, 2.5, n)
    S[:1000] = Iin[:1000] * (1 + rng.uniform(-1, 1, 1000) * 10 ** rng.uniform(-14, -2, 1000))   # near equilibrium
    Iout, mean = O.device_segment_source(Iin, tau, S)
    mp.mp.dps = 50
    worst_out = worst_mean = 0.0
    for a, s, t, b, m in zip(Iin, S, tau, Iout, mean):
        a, s, t = mp.mpf(float(a)), mp.mpf(float(s)), mp.mpf(float(t))
        e = mp.e ** (-t)
        true_out, true_mean = s + (a - s) * e, s - (a - s) * mp.expm1(-t) / t
        # (the error is relative to the larger of the two terms: S + (Iin - S) x cancels when Iin << S and x -> 1)
        scale = max(abs(s), abs(a))
        worst_out = max(worst_out, float(abs(mp.mpf(float(b)) - true_out) / scale))
        worst_mean = max(worst_mean, float(abs(mp.mpf(float(m)) - true_mean) / scale))
    assert worst_out < 4 * EPS and worst_mean < 4 * EPS, (worst_out, worst_mean)
    # S = 0: the same bits as the segment without emission
    I0 = Iin.copy()
    out0, mean0 = O.device_segment_source(I0, tau, 0.0)
    out1, mean1 = O.device_segment_emit(I0, tau, 0.0, 0.0)  # (reference form with nothing to emit: Iin e, and the log-mean evaluated from the intensities)
    e, g = O.device_attenuation(tau)
    assert np.array_equal(out0, Iin * e) and np.array_equal(mean0[out0 > 0], (Iin * g)[out0 > 0]) and np.all(mean0[out0 == 0] == 0)
    # equilibrium: nothing changes, exactly
    out, mean = O.device_segment_source(np.array([3e-21]), np.array([0.7]), np.array([3e-21]))
    assert out[0] == 3e-21 and mean[0] == 3e-21


def test_thin_bound_is_where_the_range_reduction_starts():
    """ftte_consts.thin_max = the last tau whose rounded product with 1/ln2 rounds to n = 0: the wavefront's test `|tau| <= thin_max`
    must decide exactly as `rint(-tau * log2e) == 0` did, or a lane next to the bound would get another range reduction.  Checked
    ulp by ulp on both sides, and the attenuation pair stays continuous across it."""
    log2e, thin_max = float.fromhex("0x1.71547652b82fep+0"), float.fromhex("0x1.62e42fefa39efp-2")
    t = np.float64(thin_max)
    for _ in range(2000):
        t = np.nextafter(t, 0.0)
    for k in range(4000):
        n_is_zero = np.rint(-t * np.float64(log2e)) == 0.0
        assert n_is_zero == (t <= thin_max), (k, float(t).hex())
        t = np.nextafter(t, 1.0)
    around = np.array([np.nextafter(thin_max, 0.0), thin_max, np.nextafter(thin_max, 1.0)])
    e, g = O.device_attenuation(around)
    mp.mp.dps = 40
    for tau, ee, gg in zip(around, e, g):
        assert abs(mp.mpf(float(ee)) / mp.e ** (-mp.mpf(float(tau))) - 1) < 2 * EPS
        assert abs(mp.mpf(float(gg)) / (-mp.expm1(-mp.mpf(float(tau))) / mp.mpf(float(tau))) - 1) < 3 * EPS
    # negative tau (unphysical, still evaluated): the same bound on |tau|
    e_neg, _ = O.device_attenuation(-around)
    assert np.all(e_neg > 1.0)
