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


def _worker_2d(rank, world, port, n, nnu, out_dir, use_gpu):
    """What bench.py --gpus N does per rank: its frequency slice x its direction slice, then Shard2D.combine.  The sweep is
    the oracle on CPU, or (use_gpu, run by the -m gpu test on one card) the HIP library through the C ABI."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import _oracle as O
    from radiativetransfer_amd import synthetic
    from radiativetransfer_amd.distributed import Shard2D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=9, tau_median=0.3)
    phi, theta, w = O.healpix_directions(2)
    sh = Shard2D(rank, world, nnu)
    lo, hi = sh.groups
    p, t, ww = sh.directions(phi, theta, w)
    if use_gpu:
        import radiativetransfer_amd as rt
        with rt.DiffuseTransfer(device=0) as eng:
            eng.set_uniform_grid(n, box)
            eng.set_opacity(kappa[lo:hi])
            J_local = eng.transport(p, t, ww, uvb[lo:hi])
    else:
        J_local = O.sweep_uniform(n, kappa[lo:hi], box, p, t, ww, uvb[lo:hi], arith=O.ARITH_DEVICE)
    J_slab = sh.exchange(torch.from_numpy(np.ascontiguousarray(J_local)).clone())
    np.save(os.path.join(out_dir, f"slab{rank}.npy"), J_slab.numpy())
    J = sh.combine(torch.from_numpy(np.ascontiguousarray(J_local)))
    np.save(os.path.join(out_dir, f"J{rank}.npy"), J.numpy())
    if rank == 0:
        with open(os.path.join(out_dir, "layout.txt"), "w") as f:
            f.write(sh.describe())
    dist.barrier()
    dist.destroy_process_group()


def _check_2d(tmp_path, world, nnu, n, use_gpu):
    import torch.multiprocessing as mp
    import _oracle as O
    from radiativetransfer_amd import synthetic
    from radiativetransfer_amd.distributed import decompose
    O.build()
    port = 33500 + os.getpid() % 2000 + 7 * world + nnu
    mp.spawn(_worker_2d, args=(world, port, n, nnu, str(tmp_path), use_gpu), nprocs=world, join=True)
    kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=9, tau_median=0.3)
    phi, theta, w = O.healpix_directions(2)
    ref = O.sweep_uniform(n, kappa, box, phi, theta, w, uvb, arith=O.ARITH_DEVICE)
    for rank in range(world):  # every rank ends up with the whole J
        J = np.load(tmp_path / f"J{rank}.npy")
        assert J.shape == ref.shape
        assert np.allclose(J, ref, rtol=64 * np.finfo(float).eps, atol=0)
    # ... or, after `exchange`, with all groups for its slab of the cells: the slabs tile the grid exactly once
    from radiativetransfer_amd.distributed import Shard2D
    covered = 0
    for rank in range(world):
        lo, hi = Shard2D(rank, world, nnu).slab(ref.shape[1])
        slab = np.load(tmp_path / f"slab{rank}.npy")
        assert slab.shape == (nnu, hi - lo) and lo == covered
        assert np.allclose(slab, ref[:, lo:hi], rtol=64 * np.finfo(float).eps, atol=0)
        covered = hi
    assert covered == ref.shape[1]
    r_nu, r_dir = decompose(world, nnu)
    assert r_nu * r_dir == world and nnu % r_nu == 0
    return open(tmp_path / "layout.txt").read()


@pytest.mark.parametrize("world,nnu,expect", [(2, 2, "2 frequency slice(s) x 1 direction"), (3, 2, "1 frequency slice(s) x 3 direction"),
                                              (4, 2, "2 frequency slice(s) x 2 direction"), (2, 8, "2 frequency slice(s) x 1 direction"),
                                              (8, 8, "8 frequency slice(s) x 1 direction")])   # (the layout of the driver's 8-GPU run)
def test_frequency_by_direction_sharding(tmp_path, world, nnu, expect):
    """bench.py's multi-GPU layout on CPU (gloo): frequency groups first (no reduction, an all-gather), directions split
    only beyond that (all-reduce within a frequency slice, then the all-gather)."""
    assert _check_2d(tmp_path, world, nnu, 8, use_gpu=False).startswith(expect)


@pytest.mark.gpu
@pytest.mark.parametrize("world,nnu", [(2, 4), (3, 2)])
def test_frequency_by_direction_sharding_through_the_library(tmp_path, world, nnu):
    """The same layout with every rank sweeping its share on the GPU through libftte.so (the ranks share the one card of the
    test box; the exchange stays on gloo: RCCL needs one device per rank)."""
    _check_2d(tmp_path, world, nnu, 24, use_gpu=True)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,exchange", [(2, "slabs"), (3, "slabs"), (2, "gather")])
def test_bench_multi_rank_path_rehearsed_on_one_gpu(tmp_path, ranks, exchange):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), rehearsed on the one-GPU box:
    the ranks share GPU 0 and the collectives run on host copies over gloo (RCCL refuses two ranks on one device).  Checks that
    the script's multi-rank branch runs through and reports BASELINE's fixed workload, sharded."""
    import json
    import subprocess
    port = 36500 + os.getpid() % 2000 + ranks
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
           "--grid", "64", "--rehearse-on-one-gpu", "--exchange", exchange]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == ranks and line["scaling"] == "strong" and line["steps"] == 2
    assert line["config"]["ndir_total"] == 96 and line["config"]["nnu"] == 8
    assert line["config"]["nnu_this_rank"] * line["config"]["ndir_this_rank"] * ranks >= 8 * 96   # (3 ranks: 32 directions each)
    assert line["value"] > 0 and "cpu_baseline" not in line


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_source_iteration_over_ranks_rehearsed_on_one_gpu(tmp_path, ranks):
    """BASELINE configs[4] over several ranks (tools/bench_config5.py under torch.distributed.run; the ranks share GPU 0, the
    collectives run on host copies over gloo): a rank keeps J and S of its frequency groups.  Two ranks split the eight groups
    -- nothing is summed, every group's J equals the single-process run bit for bit --, three ranks split the directions -- one
    all-reduce per iteration, equal to the rounding of the sum over directions carried through the iterations."""
    import subprocess
    script = os.path.join(ROOT, "tools", "bench_config5.py")
    one, many = tmp_path / "one", tmp_path / "many"
    one.mkdir(); many.mkdir()
    run = subprocess.run([sys.executable, script, "64", "5", f"--dump={one}"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-3000:]
    port = 37100 + os.getpid() % 2000 + ranks
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script, "64", "5", "--rehearse-on-one-gpu", f"--dump={many}"]
    multi = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert multi.returncode == 0, multi.stderr[-3000:]
    history = lambda out: [ln.split("|dJ|/|J| = ")[1] for ln in out.splitlines() if "|dJ|/|J|" in ln]
    assert len(history(multi.stdout)) == len(history(run.stdout)) == 4           # iterations 1, 2, 3 and the last
    J = np.load(one / "J0.npy")
    from radiativetransfer_amd.distributed import Shard2D
    for rank in range(ranks):
        sh = Shard2D(rank, ranks, 8)
        lo, hi = sh.groups
        mine = np.load(many / f"J{rank}.npy")
        assert mine.shape == (hi - lo, 64 ** 3)
        if sh.r_dir == 1:
            assert np.array_equal(mine, J[lo:hi])
        else:
            assert np.allclose(mine, J[lo:hi], rtol=1e-12, atol=0)
    if ranks == 2:
        assert history(multi.stdout) == history(run.stdout)
    else:
        assert np.allclose([float(x) for x in history(multi.stdout)], [float(x) for x in history(run.stdout)], rtol=1e-9)


def _worker_sum(rank, world, port, nnu, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from radiativetransfer_amd.distributed import Shard2D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = Shard2D(rank, world, nnu)
    lo, hi = sh.groups
    J = torch.from_numpy(np.random.default_rng(100 + rank).random((hi - lo, 50)))
    np.save(os.path.join(out_dir, f"part{rank}.npy"), J.numpy().copy())
    sh.sum_directions(J)
    np.save(os.path.join(out_dir, f"sum{rank}.npy"), J.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nnu", [(2, 2), (3, 2), (4, 2)])
def test_sum_over_direction_slices_only(tmp_path, world, nnu):
    """Shard2D.sum_directions, what a source iteration over ranks does between two sweeps: a rank's groups summed over the ranks
    that sweep the SAME groups for other directions, nothing from the other frequency slices (2 ranks x 2 groups: nothing at all)."""
    import torch.multiprocessing as mp
    from radiativetransfer_amd.distributed import Shard2D
    port = 38100 + os.getpid() % 2000 + 5 * world
    mp.spawn(_worker_sum, args=(world, port, nnu, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        sh = Shard2D(rank, world, nnu)
        peers = [r for r in range(world) if Shard2D(r, world, nnu).i_nu == sh.i_nu]
        want = sum(np.load(tmp_path / f"part{r}.npy") for r in peers)
        assert len(peers) == sh.r_dir
        assert np.allclose(np.load(tmp_path / f"sum{rank}.npy"), want, rtol=4 * np.finfo(float).eps, atol=0)
