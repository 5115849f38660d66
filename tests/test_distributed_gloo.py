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
