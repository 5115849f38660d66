#!/usr/bin/env python3
"""bench.py -- the headline measurement: cell.direction.frequency updates/s of one diffuse-transfer
iteration on a 256^3 uniform grid, and the sweep kernel's achieved fraction of the HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 256] [--nnu 8] [--ndir 96]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one full source iteration of the hot path on inputs already resident in HBM:
new opacities are handed to the library (device-to-device, including the two layout transposes the
sweep needs), every direction of this rank is swept for every frequency group, the per-slot
accumulators are merged into J, and -- for N > 1 -- J is summed over ranks with an RCCL all-reduce.

Workload (BASELINE.json configs[1], SURVEY.md section 8(d) "config 2"): 256^3 cells, 8 frequency groups,
96 directions per GPU = NESTED pixels of nside 4 after the reference's rotateAngles, equal weights;
log-normal opacity field (sigma_ln = 1, seed 12345), zero emissivity.  Multi-GPU is weak scaling in the
angular quadrature: rank r sweeps pixels [96 r, 96 (r+1)) of the first 96 N pixels with weight 1/(96 N)
(N = 2 is the reference's own 192-direction set, N = 8 all 768 pixels of nside 4... capped at 768).

The JSON line also carries
  roofline    : sweep kernel, algorithmic bytes (24 B per update) / HIP-event time of its launches
  cpu_baseline: the reference's own compiled transport (oracle/_ref) or the C oracle, on the host cores
"""
from __future__ import annotations

import argparse
import json
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_UPDATE = 24  # kappa 8 B read + J 8 B read + 8 B write
CPU_SAMPLE_DIRS = 4    # directions in the bounded CPU sample (about 10-20 s of host work at 256^3)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--nnu", type=int, default=8)
    ap.add_argument("--ndir", type=int, default=96, help="directions per GPU")
    ap.add_argument("--rows", type=int, default=0, help="rays per lane (4/8/16); 0 = library default")
    ap.add_argument("--slots", type=int, default=0, help="directions in flight per launch; 0 = library default")
    ap.add_argument("--waves", type=int, default=0, help="waves per SIMD the kernel is compiled for; 0 = library default")
    ap.add_argument("--engine", type=int, default=0, help="0 library default, 1 ray-following tiles, 2 cell-fixed bricks")
    ap.add_argument("--chunk", type=int, default=0, help="bricks: layers per brick; 0 = library default")
    ap.add_argument("--group", type=int, default=0, help="bricks: directions per group; 0 = library default")
    ap.add_argument("--brick-waves", type=int, default=0, help="bricks: waves per SIMD the kernel is compiled for")
    ap.add_argument("--team", type=int, default=-1, help="bricks: 1 one wavefront per direction (default), 0 one wavefront per group")
    ap.add_argument("--share", type=int, default=-1, help="bricks: accumulator sharing 0/1/2")
    ap.add_argument("--lanes", type=int, default=0, help="bricks: streams the frequency groups are spread over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=0, help="grid size of the CPU sample (default: --n)")
    return ap.parse_args()


def directions(total: int, per_rank: int, rank: int):
    import radiativetransfer_amd as rt
    nside = 4
    while 12 * nside * nside < total:
        nside *= 2
    lo = rank * per_rank
    ang = np.array([rt.pix2ang_nest(nside, i) for i in range(lo, lo + per_rank)])
    return ang[:, 0].copy(), ang[:, 1].copy(), np.full(per_rank, 1.0 / total)


def cpu_baseline(n, kappa_host3, uvb3, box, phi, theta, w):
    """Times the CPU path on a bounded sample of the same workload: the first three frequency groups
    (the reference hard-wires three, definitionsModule.f90:169-171) and the first direction."""
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    ncell = n ** 3
    nd = min(CPU_SAMPLE_DIRS, len(phi))
    updates = ncell * 3 * nd
    sample = f"{n}^3 grid, first 3 frequency groups, first {nd} directions of the workload ({updates:.3g} updates)"
    if os.path.exists(harness):
        with tempfile.TemporaryDirectory() as tmp:
            case, out = os.path.join(tmp, "case.bin"), os.path.join(tmp, "out.bin")
            with open(case, "wb") as f:
                f.write(struct.pack("<4i", n, ncell, nd, 0))
                f.write(struct.pack("<d", box))
                f.write(np.asarray(uvb3, "<f8").tobytes())
                f.write(np.zeros(ncell, "<i4").tobytes())
                f.write(np.ascontiguousarray(kappa_host3, "<f8").tobytes())
                for a in (phi[:nd], theta[:nd], w[:nd]):
                    f.write(np.asarray(a, "<f8").tobytes())
            try:
                res = subprocess.run([harness, case, out], capture_output=True, text=True, timeout=900)
                secs = None
                for line in res.stdout.splitlines():
                    if line.strip().startswith("SWEEP_SECONDS"):
                        secs = float(line.split()[1])
                if secs and secs > 0:
                    return {"value": updates / secs, "unit": "updates/s", "cores": 1, "kind": "reference",
                            "sample": sample + "; reference modules compiled with amdflang -O2 (oracle/_ref), "
                                               "pattern + neighbour set-up + transport timed, tree build excluded"}
            except Exception as e:  # fall through to the port
                print(f"[bench] reference harness failed: {e}", file=sys.stderr)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    t0 = time.perf_counter()
    O.sweep_uniform(n, kappa_host3, box, phi[:nd], theta[:nd], w[:nd], uvb3)
    secs = time.perf_counter() - t0
    return {"value": updates / secs, "unit": "updates/s", "cores": 1, "kind": "port",
            "sample": sample + "; C restatement oracle/ftte_oracle.c, gcc -O2"}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    import radiativetransfer_amd as rt
    from radiativetransfer_amd import synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the sweep has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    n, nnu, ndir = a.n, a.nnu, a.ndir
    ncell = n ** 3
    total_dirs = ndir * world
    phi, theta, w = directions(total_dirs, ndir, rank)
    kappa_host, uvb, box = synthetic.uniform_workload(n, nnu, seed=12345, tau_median=0.1)
    kappa = torch.from_numpy(kappa_host).to(dev)
    J = torch.empty((nnu, ncell), dtype=torch.float64, device=dev)

    eng = rt.DiffuseTransfer(device=local)
    eng.set_uniform_grid(n, box)
    if a.rows:
        eng.set_option("rows", a.rows)
    if a.slots:
        eng.set_option("slots", a.slots)
    if a.waves:
        eng.set_option("waves", a.waves)
    if a.engine:
        eng.set_option("engine", a.engine)
    if a.chunk:
        eng.set_option("chunk", a.chunk)
    if a.group:
        eng.set_option("group", a.group)
    if a.brick_waves:
        eng.set_option("brick_waves", a.brick_waves)
    if a.team >= 0:
        eng.set_option("team", a.team)
    if a.share >= 0:
        eng.set_option("share", a.share)
    if a.lanes:
        eng.set_option("lanes", a.lanes)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.set_opacity_device(nnu, kappa.data_ptr())
        eng.transport_device(phi, theta, w, uvb, J.data_ptr(), stream)
        if world > 1:
            dist.all_reduce(J)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    launch_ms, launch_updates = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        # the launch records of a sweep are read after the sweep has drained; the next sweep starts with a stream
        # synchronise anyway (it rewrites device tables), so this adds no bubble of its own
        torch.cuda.synchronize()
        for ms, upd in eng.launch_records():
            launch_ms += ms
            launch_updates += upd
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    updates_per_step = ncell * total_dirs * nnu
    value = updates_per_step * a.steps / elapsed
    nlaunch = len(eng.launch_records()) * a.steps
    achieved = launch_updates * BYTES_PER_UPDATE / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0

    out = {
        "metric": "cell·dir·ν updates/sec per iteration, 256³ grid; achieved HBM GB/s vs peak",
        "value": value, "unit": "updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n}^3 uniform grid, {nnu} frequency groups, {ndir} directions per GPU "
                               f"({total_dirs} in all, NESTED pixels of the rotated HEALPix set), diffuse sweep, "
                               "log-normal opacity, zero emissivity (BASELINE.json configs[1])",
                   "grid": n, "nnu": nnu, "ndir_per_gpu": ndir, "ndir_total": total_dirs,
                   "parallelism": f"directions sharded over {world} rank(s), RCCL all-reduce of J" if world > 1
                   else "single GPU"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(n, nnu),
                     "kernel": "ftte::sweep_kernel", "launches": nlaunch,
                     "avg_launch_ms": launch_ms / nlaunch if nlaunch else None,
                     "bytes_per_update": BYTES_PER_UPDATE},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cn = a.cpu_n or n
        if cn == n:
            k3 = kappa_host[:3]
        else:
            k3, _, _ = synthetic.uniform_workload(cn, 3, seed=12345, tau_median=0.1)
        out["cpu_baseline"] = cpu_baseline(cn, k3, uvb[:3], box, phi, theta, w)
    if rank == 0:
        print(json.dumps(out, ensure_ascii=False))
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(n, nnu):
    """HBM bytes per sweep-kernel launch from the committed rocprofv3 --pmc passes (profiles/), if one matches this
    workload; bench.py cannot collect counters itself."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))
        if rec.get("grid") == n and rec.get("nnu") == nnu:
            return rec.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


if __name__ == "__main__":
    main()
