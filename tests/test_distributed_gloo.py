"""The N > 1 path on CPU: two gloo ranks shard the direction list with radiativetransfer_amd.distributed, each sweeps
its share (with the oracle standing in for the GPU, which this container does not have) and the all-reduced J
equals the single-process J."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import _oracle as O
    from radiativetransfer_amd import synthetic
    from radiativetransfer_amd.distributed import allreduce_J, shard_directions
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kappa, uvb, box = synthetic.uniform_workload(n, 2, seed=4, tau_median=0.3)
    phi, theta, w = O.healpix_directions(2)
    p, t, ww = shard_directions(phi, theta, w, rank, world)
    J = torch.from_numpy(O.sweep_uniform(n, kappa, box, p, t, ww, uvb, arith=O.ARITH_DEVICE))
    allreduce_J(J)
    if rank == 0:
        np.save(os.path.join(out_dir, "J.npy"), J.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_directions_allreduce(tmp_path, world):
    import torch.multiprocessing as mp
    import _oracle as O
    from radiativetransfer_amd import synthetic
    O.build()
    n = 10
    port = 29500 + os.getpid() % 2000 + world
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    J = np.load(tmp_path / "J.npy")
    kappa, uvb, box = synthetic.uniform_workload(n, 2, seed=4, tau_median=0.3)
    phi, theta, w = O.healpix_directions(2)
    ref = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
    assert np.allclose(J, ref, rtol=1e-14, atol=0)


def _worker_groups_and_stars(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import _oracle as O
    from radiativetransfer_amd import synthetic
    from radiativetransfer_amd.distributed import allreduce_rates, gather_J, shard_groups, shard_sources
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # frequency groups split: no reduction, an all-gather assembles J
    nnu = 5
    kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=4, tau_median=0.3)
    phi, theta, w = O.healpix_directions(1)
    lo, hi = shard_groups(nnu, rank, world)
    J_local = torch.from_numpy(O.sweep_uniform(n, kappa[lo:hi], box, phi, theta, w, uvb[lo:hi], arith=O.ARITH_DEVICE)) if hi > lo \
        else torch.empty((0, n ** 3), dtype=torch.float64)
    J = gather_J(J_local, nnu)
    # stars split: rates summed
    g = np.load(os.path.join(HERE, "golden", "point16_homogeneous.npz"))
    rng = np.random.default_rng(1)
    cells, ndot = rng.choice(16 ** 3, 5, replace=False), rng.uniform(1, 2, 5)
    mine_c, mine_n = shard_sources(cells, ndot, rank, world)
    rates = np.zeros((6, 16 ** 3))
    if len(mine_c):
        rates, _ = O.point_sources(16, g["level"], g["HI"], g["HeI"], g["HeII"], g["rho"], g["abun2"], float(g["box"]), 0, mine_c, mine_n,
                                   g["tables"].reshape(6, -1))
    rates = allreduce_rates(torch.from_numpy(rates))
    if rank == 0:
        np.save(os.path.join(out_dir, "J.npy"), J.numpy())
        np.save(os.path.join(out_dir, "rates.npy"), rates.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_groups_and_stars(tmp_path, world):
    import torch.multiprocessing as mp
    import _oracle as O
    from radiativetransfer_amd import synthetic
    O.build()
    n = 8
    port = 31500 + os.getpid() % 2000 + world
    mp.spawn(_worker_groups_and_stars, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    kappa, uvb, box = synthetic.uniform_workload(n, 5, seed=4, tau_median=0.3)
    phi, theta, w = O.healpix_directions(1)
    assert np.array_equal(np.load(tmp_path / "J.npy"), O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE))
    g = np.load(os.path.join(HERE, "golden", "point16_homogeneous.npz"))
    rng = np.random.default_rng(1)
    cells, ndot = rng.choice(16 ** 3, 5, replace=False), rng.uniform(1, 2, 5)
    ref, _ = O.point_sources(16, g["level"], g["HI"], g["HeI"], g["HeII"], g["rho"], g["abun2"], float(g["box"]), 0, cells, ndot,
                             g["tables"].reshape(6, -1))
    rates = np.load(tmp_path / "rates.npy")
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert np.all(np.abs(rates - ref) <= 1e-13 * np.abs(ref) + 1e-15 * scale)
