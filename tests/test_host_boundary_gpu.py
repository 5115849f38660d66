"""The host-array boundary as the reference's driver would use it: the same cell array handed over on every outer
iteration (the tree is static over a run, equiSources.f90:1230-1843 never rebuilds it), host arrays for kappa and J.

  * ftte_set_grid with the level list the context already holds keeps the tree, the sweep plan and the segment forests
    (counters of ftte_counter), and the results stay bit-identical;
  * pageable arrays go through the library's pinned staging blocks, registered arrays by DMA directly: same bits.
"""
import numpy as np
import pytest

import _oracle as O
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic

pytestmark = pytest.mark.gpu


def test_unchanged_level_list_keeps_tree_plan_and_forests(golden):
    g = golden("amr8_block_level1")
    n, level, box = int(g["n"]), g["level"], float(g["box"])
    with rt.DiffuseTransfer() as e:
        e.set_grid(n, level, box)
        e.set_opacity(g["kappa"])
        J1 = e.transport(g["phi"], g["theta"], g["w"], g["uvb"])
        assert (e.counter("grid_builds"), e.counter("forest_builds")) == (1, 1)
        for _ in range(3):  # the drop-in's call sequence, every outer iteration
            e.set_grid(n, level, box)
            e.set_opacity(g["kappa"])
            J2 = e.transport(g["phi"], g["theta"], g["w"], g["uvb"])
        assert (e.counter("grid_builds"), e.counter("forest_builds")) == (1, 1)
        assert np.array_equal(J1, J2)
        # a different box with the same tree: the tree stays, the forests (segment lengths) are rebuilt
        e.set_grid(n, level, 2 * box)
        J3 = e.transport(g["phi"], g["theta"], g["w"], g["uvb"])
        assert (e.counter("grid_builds"), e.counter("forest_builds")) == (1, 2)
        assert np.array_equal(J3, O.sweep_tree(n, level, g["kappa"], 2 * box, g["phi"], g["theta"], g["w"], g["uvb"],
                                               arith=O.ARITH_DEVICE))
        # a different list: everything is rebuilt
        e.set_uniform_grid(n, box)
        assert e.counter("grid_builds") == 2
        assert e.counter("unknown") == -1


def test_unchanged_uniform_grid_keeps_the_plan():
    n = 20
    kappa, uvb, box = synthetic.uniform_workload(n, 2, seed=5, tau_median=0.3)
    phi, theta, w = O.healpix_directions(2)
    with rt.DiffuseTransfer() as e:
        e.set_uniform_grid(n, box)
        e.set_opacity(kappa)
        J1 = e.transport(phi, theta, w, uvb)
        e.set_uniform_grid(n, box)
        e.set_opacity(kappa)
        J2 = e.transport(phi, theta, w, uvb)
        assert (e.counter("grid_builds"), e.counter("plan_builds")) == (1, 1)
        assert np.array_equal(J1, J2)
        J3 = e.transport(phi[:7], theta[:7], w[:7], uvb)  # another direction list: a new plan, the same tree
        assert (e.counter("grid_builds"), e.counter("plan_builds")) == (1, 2)
        assert J3.shape == J1.shape


def test_registered_and_pageable_host_arrays_give_the_same_bits():
    """64^3 x 3 groups = 6.3 MB per array: beyond the 1 MiB below which the staging is skipped; 130^3 x 8 = 140 MB crosses
    several 64 MiB staging blocks in both directions."""
    for n, nnu in ((64, 3), (130, 8)):
        kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=n, tau_median=0.2)
        phi, theta, w = O.healpix_directions(1)
        with rt.DiffuseTransfer() as e:
            e.set_uniform_grid(n, box)
            e.set_opacity(kappa)
            J_pageable = e.transport(phi, theta, w, uvb)
            kap_pinned = kappa.copy()
            J_pinned = np.empty_like(J_pageable)
            e.host_register(kap_pinned)
            e.host_register(J_pinned)
            e.set_opacity(kap_pinned)
            e.transport_into(phi, theta, w, uvb, J_pinned)
            assert np.array_equal(J_pageable, J_pinned)
            e.host_unregister(J_pinned)
            e.host_unregister(kap_pinned)
            with pytest.raises(rt.FtteError):
                e.host_unregister(J_pinned)
        if n == 64:
            # twelve directions: the device adds them group by group, the oracle in list order
            assert np.allclose(J_pageable, O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE),
                               rtol=64 * np.finfo(float).eps, atol=0)


@pytest.mark.parametrize("n,nnu", [(70, 3), (64, 8), (33, 2)])
def test_iteration_in_one_call_equals_the_two_calls(n, nnu):
    """ftte_diffuse_iteration = ftte_set_opacity + ftte_diffuse_sweep with the frequency groups crossing PCIe and being swept in
    overlapping lanes: the same bits, with pageable arrays (through the staging blocks) and with registered ones (DMA in place),
    and a later device-side sweep finds the opacities the call left behind."""
    kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=n, tau_median=0.3)
    phi, theta, w = O.healpix_directions(2)
    with rt.DiffuseTransfer() as e:
        e.set_uniform_grid(n, box)
        e.set_opacity(kappa)
        J_two = e.transport(phi, theta, w, uvb)
        e.set_opacity(0.5 * kappa)                      # something else in between
        J_one = np.full_like(J_two, np.nan)
        e.iterate_into(kappa, phi, theta, w, uvb, J_one)
        assert np.array_equal(J_one, J_two)
        assert np.array_equal(e.transport(phi, theta, w, uvb), J_two)   # kappa and its layouts are in place
        k2, J_reg = np.ascontiguousarray(2.0 * kappa), np.empty_like(J_two)
        e.host_register(k2); e.host_register(J_reg)
        e.iterate_into(k2, phi, theta, w, uvb, J_reg)
        e.set_opacity(k2)
        assert np.array_equal(J_reg, e.transport(phi, theta, w, uvb))
        e.host_unregister(k2); e.host_unregister(J_reg)
        e.set_option("lanes", 1)                        # no lanes: the two calls internally
        J_seq = np.empty_like(J_two)
        e.iterate_into(kappa, phi, theta, w, uvb, J_seq)
        assert np.array_equal(J_seq, J_two)


def test_iteration_in_one_call_on_a_refined_cell_array(golden):
    g = golden("amr8_block_level1")
    with rt.DiffuseTransfer() as e:
        e.set_grid(int(g["n"]), g["level"], float(g["box"]))
        J = np.empty_like(g["kappa"])
        e.iterate_into(g["kappa"], g["phi"], g["theta"], g["w"], g["uvb"], J)
        e.set_opacity(g["kappa"])
        assert np.array_equal(J, e.transport(g["phi"], g["theta"], g["w"], g["uvb"]))


@pytest.mark.parametrize("n", [64, 70])
def test_new_device_opacities_reach_all_three_layouts(n):
    """ftte_set_opacity_device on a context whose last sweep used all three memory layouts rewrites the three copies in one pass over
    the caller's array (set_layouts_kernel) instead of a copy and two transposes: the next sweep must see the NEW opacities in every
    layout -- bit for bit what a fresh context gets for them, also on a grid that is no multiple of the 32 x 32 tiles, also when the
    number of groups changes (the one-pass form is not taken then)."""
    import torch
    phi, theta, w = O.healpix_directions(2)          # 48 directions: izones of all three march axes
    k1, uvb, box = synthetic.uniform_workload(n, 3, seed=n, tau_median=0.2)
    k2 = np.ascontiguousarray(k1[::-1] * 1.7)
    dev = torch.device("cuda", 0)

    def sweep(e, k, u):
        kd = torch.from_numpy(k).to(dev)
        e.set_opacity_device(k.shape[0], kd.data_ptr())
        J = torch.empty(k.shape, dtype=torch.float64, device=dev)
        e.transport_device(phi, theta, w, u, J.data_ptr(), 0)
        torch.cuda.synchronize()
        return J.cpu().numpy()

    with rt.DiffuseTransfer() as e:
        e.set_uniform_grid(n, box)
        sweep(e, k1, uvb)
        J2 = sweep(e, k2, uvb)                         # the one-pass form
        J2_two_groups = sweep(e, k2[:2], uvb[:2])      # another number of groups: copy, then the sweep's own transposes
        J1_again = sweep(e, k1, uvb)
    with rt.DiffuseTransfer() as fresh:
        fresh.set_uniform_grid(n, box)
        assert np.array_equal(J2, sweep(fresh, k2, uvb))
    with rt.DiffuseTransfer() as fresh:
        fresh.set_uniform_grid(n, box)
        assert np.array_equal(J1_again, sweep(fresh, k1, uvb))
    assert np.array_equal(J2_two_groups, J2[:2])
    assert np.allclose(J2, O.sweep_uniform(n, k2, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE), rtol=64 * np.finfo(float).eps, atol=0)
