"""The ionisation-equilibrium oracle (oracle/ftte_oracle_chem.c) against vectors produced by the reference's own compiled
solveRateEquations (oracle/_ref/chem_harness, tests/golden/make_golden_chem.py)."""
import numpy as np

import _oracle as O


def _run(g, k, uvb):
    tab = k
    return O.solve_rate_equations(int(g["n"]), g["level"], float(g["box"]), g["rho"], g["tgas"], g["HI"], g["HeI"], g["HeII"],
                                  g["krate"] if "krate" in g.files else None, uvb, g["J"] if "J" in g.files else None,
                                  g["ksi"] if "ksi" in g.files else None, g["uniform"] if "uniform" in g.files else None,
                                  float(g["threshold"]) if "threshold" in g.files else 0.0, float(tab["logtem0"]),
                                  float(tab["logtem9"]), float(tab["dlogtem"]), tab["k"])


def test_solve_rate_equations_bitwise(golden):
    tab = golden("chem_uvb_refined")
    for name, uvb in (("chem_uvb_refined", True), ("chem_uniform_background", False)):
        g = golden(name)
        HI, HeI, HeII, status, its = _run(g, tab, uvb)
        assert status == 0 and its > 20 * g["level"].size
        for mine, key in ((HI, "HI_out"), (HeI, "HeI_out"), (HeII, "HeII_out")):
            ref = g[key]
            assert np.array_equal(mine, ref), (name, key, np.abs(mine / ref - 1).max())


def test_equilibrium_properties(golden):
    """What the update guarantees whatever the input state: species within their element's budget, a shielded cell keeps
    only its collisional balance, and a second application changes nothing that the first one fixed."""
    tab, g = golden("chem_uvb_refined"), golden("chem_uniform_background")
    HI, HeI, HeII, status, _ = _run(g, tab, False)
    psi, mp, mn = float(np.float32(0.76)), float(np.float32(1.6726231e-24)), float(np.float32(1.67492728e-24))
    nh, nhe = psi * g["rho"] / mp, (1 - psi) * g["rho"] / (2 * (mp + mn))
    assert status == 0
    assert np.all((HI >= 0) & (HI <= nh)) and np.all((HeI >= 0) & (HeI <= nhe)) and np.all(HeII >= -1e-12 * nhe)
    # the result does not depend on the neutral fractions it starts from, except through the shielding test
    shuffled = O.solve_rate_equations(int(g["n"]), g["level"], float(g["box"]), g["rho"], g["tgas"], HI, HeI, HeII, None, False, None,
                                      None, g["uniform"], float(g["threshold"]), float(tab["logtem0"]), float(tab["logtem9"]),
                                      float(tab["dlogtem"]), tab["k"])
    mfp0 = 1.0 / (np.minimum(g["HI"], nh) * float(np.float32(6.3e-18)) + g["HeI"] * float(np.float32(7.42e-18)) +
                  g["HeII"] * float(np.float32(1.58e-18)))
    mfp1 = 1.0 / (HI * float(np.float32(6.3e-18)) + HeI * float(np.float32(7.42e-18)) + HeII * float(np.float32(1.58e-18)))
    same = (mfp0 >= float(g["threshold"])) == (mfp1 >= float(g["threshold"]))
    assert same.sum() > 100 and np.array_equal(shuffled[0][same], HI[same]) and np.array_equal(shuffled[1][same], HeI[same])


def test_uvb_beta_table_bitwise(golden):
    """Group cross-sections, photo-rate and heating coefficients (uvbBetaTable.f90) against the reference's own output."""
    g = golden("uvb_beta_table")
    for a, beta, ksi, gamma in zip(g["alpha"], g["beta"], g["ksi"], g["gamma"]):
        mine = O.uvb_beta_table(a)
        assert np.array_equal(mine[0], beta) and np.array_equal(mine[1], ksi) and np.array_equal(mine[2], gamma), a
    for a, ksi, gamma in zip(g["alpha"], g["uniform_ksi"], g["uniform_gamma"]):   # uniformTable(alpha1, alpha2)
        mine = O.uniform_table(a[0], a[1])
        assert np.array_equal(mine[0], ksi) and np.array_equal(mine[1], gamma), a


def test_assign_uvb_radiation_bitwise(golden):
    g = golden("thin_limit_uvb")
    J = O.assign_uvb_radiation(g["HI"], g["HeI"], g["HeII"], g["rho"], g["uvb"], float(g["threshold"]))
    assert np.array_equal(J, g["J"]) and 0 < (J[0] > 0).mean() < 1


def test_coll_rates_and_tables_bitwise(golden):
    """coll_rates.f (case A and case B recombination) at 64 temperatures, and the 5000-entry tables the reference's driver
    builds from it (calc_rates.f:324-337 with equiSources.f90:174-176), against the reference's own compiled routine."""
    g = golden("uvb_beta_table")
    for rtype in (1, 2):
        for T, ref in zip(g["coll_temperature"], g["coll_rates"][rtype - 1]):
            assert np.array_equal(O.coll_rates(T, rtype), ref), (rtype, T)
    c = golden("chem_uvb_refined")
    k, l0, l9, dl = O.rate_coefficient_tables(c["k"].shape[1], 1.0, float(np.float32(1.0e8)), 2)
    assert (l0, l9, dl) == (float(c["logtem0"]), float(c["logtem9"]), float(c["dlogtem"]))
    assert np.array_equal(k, c["k"])
