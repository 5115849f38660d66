"""ftte_create with ndev > 1 (csrc/ftte_multi.cpp): ONE context that splits the sweep over several devices -- frequency groups
first, then directions, as radiativetransfer_amd/distributed.py does for one process per GPU -- and sums J where the directions are
split (the only coupling of the directions: transportRoutinesModule.f90:953-955).  On the one-GPU test box the devices are the same
physical device given several times; RCCL cannot put two ranks on one device, so the sums take the kernel that reads the partners'
buffers in place (ftte_multi_info says so); the RCCL path needs distinct devices (the driver's 8-GPU node).
Tolerances: a pure frequency split is bit for bit (frequency groups never meet); a direction split changes the order in which the
directions are added: SUM_RTOL = 64 eps."""
import numpy as np
import pytest

import _oracle as O
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic

pytestmark = pytest.mark.gpu

EPS = np.finfo(np.float64).eps
SUM_RTOL = 64 * EPS


def single(n, box, kappa, phi, theta, w, uvb, level=None):
    with rt.DiffuseTransfer(device=0) as e:
        if level is None:
            e.set_uniform_grid(n, box)
        else:
            e.set_grid(n, level, box)
        e.set_opacity(kappa)
        return e.transport(phi, theta, w, uvb)


@pytest.mark.parametrize("ndev,nnu,slices", [(2, 2, (2, 1)), (2, 8, (2, 1)), (2, 3, (1, 2)), (2, 1, (1, 2)), (4, 2, (2, 2)), (3, 3, (3, 1)),
                                             (3, 2, (1, 3)), (4, 6, (2, 2))])
def test_devices_of_one_context_split_groups_then_directions(ndev, nnu, slices):
    n = 64
    kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=40 + nnu, tau_median=0.2)
    phi, theta, w = O.healpix_directions(2)
    J_one = single(n, box, kappa, phi, theta, w, uvb)
    with rt.DiffuseTransfer(devices=[0] * ndev) as e:
        assert e.counter("devices") == ndev
        e.set_uniform_grid(n, box)
        e.set_opacity(kappa)
        J = e.transport(phi, theta, w, uvb)
        assert (e.counter("frequency_slices"), e.counter("direction_slices")) == slices
        again = e.transport(phi, theta, w, uvb)
        info = e.multi_info()
    assert np.array_equal(J, again)
    if slices[1] == 1:
        assert np.array_equal(J, J_one)                      # frequency groups never meet: the same bits
        assert "nothing to sum" in info
    else:
        assert np.allclose(J, J_one, rtol=SUM_RTOL, atol=0)
        assert "peer kernel" in info and "share a physical device" in info and e.counter("multi_rccl") in (0, -1)
    assert np.allclose(J, O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE), rtol=SUM_RTOL, atol=0)


def test_multi_device_context_on_a_refined_cell_array_with_a_source_function_and_options():
    """The forest / hybrid path, emission, the one-call iteration and options all pass through the split."""
    n = 32
    blocks = [(14 + a, 15 + b, 13 + c) for a in range(3) for b in range(2) for c in range(2)]
    level = synthetic.refine_levels(n, blocks, depth=1)
    rho = synthetic.lognormal_density(len(level), seed=9)
    _, s_nu, uvb = synthetic.frequency_groups(3)
    kappa = (0.2 * n) * s_nu[:, None] * rho[None, :] * (2.0 ** level)[None, :]
    S = np.random.default_rng(2).uniform(0.1, 2.0, kappa.shape) * uvb[:, None]
    phi, theta, w = O.healpix_directions(2)
    with rt.DiffuseTransfer(device=0) as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        e.set_source_function(S)
        J_one = e.transport(phi, theta, w, uvb)
    with rt.DiffuseTransfer(devices=[0, 0]) as e:
        e.set_grid(n, level, 1.0)
        e.set_opacity(kappa)
        e.set_source_function(S)
        e.set_option("chunk", 4)
        J = e.transport(phi, theta, w, uvb)
        assert np.allclose(J, J_one, rtol=SUM_RTOL, atol=0)
        e.set_source_function(None)
        J_plain = e.iterate_into(0.5 * kappa, phi, theta, w, uvb, np.empty_like(kappa))   # new opacities and the sweep in one call
    assert np.allclose(J_plain, O.sweep_tree(n, level, 0.5 * kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE), rtol=SUM_RTOL, atol=0)
    return
    assert np.allclose(J_plain, O.sweep_tree(n, level, kappa, 1.0, phi, theta, w, uvb, arith=O.ARITH_DEVICE), rtol=SUM_RTOL, atol=0)


def test_what_a_multi_device_context_refuses_and_how_it_fails():
    kappa, uvb, box = synthetic.uniform_workload(16, 2, seed=1, tau_median=0.2)
    phi, theta, w = O.healpix_directions(1)
    with rt.DiffuseTransfer(devices=[0, 0]) as e:
        with pytest.raises(rt.FtteError) as err:
            e.transport(phi, theta, w, uvb)                         # no grid yet
        assert err.value.status == "FTTE_ERR_STATE"
        e.set_uniform_grid(16, box)
        with pytest.raises(rt.FtteError) as err:
            e.transport(phi, theta, w, uvb)                         # no opacities yet
        assert err.value.status == "FTTE_ERR_STATE"
        e.set_opacity(kappa)
        with pytest.raises(rt.FtteError) as err:
            e.set_opacity_device(2, 0x1000)                         # device pointers belong to one device
        assert err.value.status == "FTTE_ERR_UNSUPPORTED"
        with pytest.raises(rt.FtteError) as err:
            e.set_option("chunk", -1)                               # options are checked by the devices' contexts
        assert err.value.status == "FTTE_ERR_ARG"
        with pytest.raises(rt.FtteError) as err:
            e.transport([0.0], [0.4], [1.0], uvb)                   # the reference stops on a quadrant boundary
        assert err.value.status == "FTTE_ERR_PHI"
        J = e.transport(phi, theta, w, uvb)                         # and the context is still good
    assert np.allclose(J, O.sweep_uniform(16, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE), rtol=SUM_RTOL, atol=0)
    with pytest.raises(rt.FtteError):
        rt.DiffuseTransfer(devices=[0, 99])                         # no such device


def test_rccl_is_there_for_distinct_devices():
    """What the one-GPU box can check of the RCCL path: librccl.so loads and exports the entry points csrc/ftte_multi.cpp calls
    (ncclCommInitAll, ncclReduceScatter, ncclGroupStart/End, ncclCommDestroy, ncclGetErrorString)."""
    with rt.DiffuseTransfer(devices=[0, 0]) as e:
        assert e.counter("rccl_loadable") == 1


def test_rccl_calling_sequence_on_a_clique_of_one():
    """ncclCommInitAll, a grouped ncclReduceScatter of doubles on the sweep's stream and ncclCommDestroy, on one rank: the library
    takes the arguments csrc/ftte_multi.cpp passes and the piece lands in the receive buffer unchanged (sum over one rank)."""
    with rt.DiffuseTransfer(devices=[0, 0]) as e:
        assert e.counter("rccl_selftest") == 1
        n = 16                                                      # and the context still sweeps afterwards
        kappa, uvb, box = synthetic.uniform_workload(n, 2, seed=3, tau_median=0.2)
        phi, theta, w = O.healpix_directions(1)
        e.set_uniform_grid(n, box)
        e.set_opacity(kappa)
        J = e.transport(phi, theta, w, uvb)
    assert np.allclose(J, O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE), rtol=SUM_RTOL, atol=0)
