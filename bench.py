#!/usr/bin/env python3
"""bench.py -- the headline measurement: cell.direction.frequency updates/s of one diffuse-transfer
iteration on a 256^3 uniform grid, and the sweep kernel's achieved fraction of the HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 256] [--nnu 8] [--ndir 96]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one full source iteration of the hot path on inputs already resident in HBM:
new opacities are handed to the library (device-to-device, including the two layout transposes the
sweep needs), every direction of this rank is swept for every frequency group of this rank, the accumulators are merged
into J, and -- for N > 1 -- the ranks' pieces of J are combined over RCCL.

Workload (BASELINE.json configs[1], SURVEY.md section 8(d) "config 2"): 256^3 cells, 8 frequency groups,
96 directions = NESTED pixels 0..95 of nside 4 after the reference's rotateAngles, equal weights;
log-normal opacity field (sigma_ln = 1, seed 12345), zero emissivity.  --ndir 192 is configs[2] (all pixels of nside 4).

Multi-GPU is STRONG scaling: the same 96 (or 192) directions x 8 groups whatever N (north_star: "96 directions, 8
frequency bins at 1/2/4/8 MI355X").  Ranks form a (frequency slice) x (direction slice) grid, frequency groups first
(radiativetransfer_amd/distributed.py: Shard2D): at N = 2, 4, 8 with 8 groups every rank owns 8/N groups for all
directions and its J_nu is complete where it is computed.  What follows the sweep in the reference's loop is the per-cell
equilibrium update (equiSources.f90:3459-3677), which needs every J_nu of a cell and nothing of other cells, so the
exchange that closes a step hands every rank ALL groups for 1/N of the cells (Shard2D.exchange: an all-to-all between the
frequency slices, 1/N of J leaves each rank; where N does not divide the groups the directions are split as well and a
reduce-scatter over the direction slices comes first).  --exchange gather assembles the whole J on every rank instead
(all-gather, N-1 times the traffic).  The line reports the time of the sweep and of the collective separately
(config.compute_ms_per_step, config.collective_ms_per_step).
--weak restores round 1's mode (96 directions per GPU, 96 N in all, all-reduce).

The JSON line also carries
  roofline    : sweep kernel: `achieved` = algorithmic bytes (24 B per update) / HIP-event time of its launches; `frac` = HBM bytes the
                counters saw (a committed --measure-traffic record of the same configuration) / the same time / peak
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
    ap.add_argument("--n", "--grid", dest="n", type=int, default=256)  # (--grid: torch.distributed.run takes "--n" for an abbreviation of its own options)
    ap.add_argument("--nnu", type=int, default=8)
    ap.add_argument("--ndir", type=int, default=96, help="directions in all (per GPU with --weak)")
    ap.add_argument("--weak", action="store_true", help="96 directions per GPU instead of 96 in all")
    ap.add_argument("--tau-median", type=float, default=0.1,
                    help="median optical depth of a cell in the lowest frequency group (0.1: the headline field; 1.0: most segments of the "
                         "first groups leave the thin range and pay the division)")
    ap.add_argument("--exchange", choices=["slabs", "gather"], default="slabs",
                    help="N > 1: every rank ends with all groups for 1/N of the cells (all-to-all), or with the whole J (all-gather)")
    ap.add_argument("--rows", type=int, default=0, help="rays per lane (4/8/16); 0 = library default")
    ap.add_argument("--slots", type=int, default=0, help="directions in flight per launch; 0 = library default")
    ap.add_argument("--waves", type=int, default=0, help="waves per SIMD the kernel is compiled for; 0 = library default")
    ap.add_argument("--engine", type=int, default=0, help="0 library default, 1 ray-following tiles, 2 cell-fixed bricks")
    ap.add_argument("--chunk", type=int, default=0, help="bricks: layers per brick; 0 = library default")
    ap.add_argument("--group", type=int, default=0, help="bricks: directions per group; 0 = library default")
    ap.add_argument("--brick-waves", type=int, default=0, help="bricks: waves per SIMD the kernel is compiled for")
    ap.add_argument("--tiled", type=int, default=-1, help="bricks: 1 opacities and accumulators stored brick by brick, 0 as frames (default)")
    ap.add_argument("--pair-waves", type=int, default=0, help="pair kernel (--team 2): workgroups per SIMD it is built for (2..4)")
    ap.add_argument("--team", type=int, default=-1, help="bricks: 0 one wavefront per brick, 2 two wavefronts per brick; default: by the number of frequency groups")
    ap.add_argument("--share", type=int, default=-1, help="bricks: accumulator sharing 0/1/2")
    ap.add_argument("--dataflow", type=int, default=-1, help="bricks: 0 a launch per stage (default); 1, 2 one launch for the sweep, bricks wait on flags; 3 one persistent launch with a task queue per XCD")
    ap.add_argument("--lanes", type=int, default=0, help="bricks: streams the frequency groups are spread over")
    ap.add_argument("--opt", action="append", default=[], help="any other library option, key=value (ftte_set_option)")
    ap.add_argument("--ldspad", type=int, default=0, help="diagnostic: extra dynamic LDS per workgroup (bytes), to cap residency")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks sharing GPU 0 with the collectives on host copies over gloo: exercises this script's multi-rank "
                         "path on a one-GPU box (RCCL refuses two ranks on one device); the timings mean nothing")
    ap.add_argument("--in-process", type=int, default=0, metavar="N",
                    help="ONE process, one library context over N devices (ftte_create with ndev = N: what a serial Fortran host gets), "
                         "host arrays in and out: the PCIe-inclusive rate of ftte_diffuse_iteration, not the contract's `value`")
    ap.add_argument("--same-device", action="store_true", help="--in-process on a one-GPU box: all N contexts on device 0 (rehearsal)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="collect this configuration's HBM traffic and SQ counters with rocprofv3 --pmc passes of this script (child "
                         "processes, a few minutes) and write profiles/pmc_traffic.json, which later runs of the same configuration quote")
    ap.add_argument("--cpu-n", type=int, default=0, help="grid size of the CPU sample (default: --n)")
    return ap.parse_args()


def directions(total: int):
    """The first `total` NESTED pixels of the smallest HEALPix level that has that many (nside 4 for 96 and 192:
    equiSources.f90:1385-1391 with nAngularLevel = 3), rotated as the reference rotates them, equal weights."""
    import radiativetransfer_amd as rt
    nside = 1
    while 12 * nside * nside < total:
        nside *= 2
    ang = np.array([rt.pix2ang_nest(nside, i) for i in range(total)])
    return ang[:, 0].copy(), ang[:, 1].copy(), np.full(total, 1.0 / total)


def host_identity():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def _write_case(path, n, box, uvb3, kappa3, phi, theta, w):
    ncell = n ** 3
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", n, ncell, len(phi), 0))
        f.write(struct.pack("<d", box))
        f.write(np.asarray(uvb3, "<f8").tobytes())
        f.write(np.zeros(ncell, "<i4").tobytes())
        f.write(np.ascontiguousarray(kappa3, "<f8").tobytes())
        for a in (phi, theta, w):
            f.write(np.asarray(a, "<f8").tobytes())


def _sweep_seconds(res):
    for line in res.stdout.splitlines():
        if line.strip().startswith("SWEEP_SECONDS"):
            return float(line.split()[1])
    return None


def cpu_baseline(n, kappa_host3, uvb3, box, phi, theta, w):
    """Times the CPU path on a bounded sample of the same workload: the first three frequency groups
    (the reference hard-wires three, definitionsModule.f90:169-171) and the first directions; one core (the reference
    is serial), and -- what a host could do at best with the reference's code -- one process per direction on all cores."""
    from radiativetransfer_amd import synthetic
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    model, cores_total, cores_usable = host_identity()
    ncell = n ** 3
    nd = min(CPU_SAMPLE_DIRS, len(phi))
    updates = ncell * 3 * nd
    sample = f"{n}^3 grid, first 3 frequency groups, first {nd} directions of the workload ({updates:.3g} updates)"
    ident = {"cpu_model": model, "host_cores": cores_total, "host_cores_usable": cores_usable}
    if os.path.exists(harness):
        with tempfile.TemporaryDirectory() as tmp:
            case, out = os.path.join(tmp, "case.bin"), os.path.join(tmp, "out.bin")
            _write_case(case, n, box, uvb3, kappa_host3, phi[:nd], theta[:nd], w[:nd])
            try:
                res = subprocess.run([harness, case, out], capture_output=True, text=True, timeout=900)
                secs = _sweep_seconds(res)
                if secs and secs > 0:
                    rec = {"value": updates / secs, "unit": "updates/s", "cores": 1, "kind": "reference",
                           "sample": sample + "; reference modules compiled with amdflang -O2 (oracle/_ref), "
                                              "pattern + neighbour set-up + transport timed, tree build excluded", **ident}
                    rec.update(all_cores_leg(harness, tmp, box, uvb3, phi, theta, w, cores_usable))
                    return rec
            except Exception as e:  # fall through to the port
                print(f"[bench] reference harness failed: {e}", file=sys.stderr)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    t0 = time.perf_counter()
    O.sweep_uniform(n, kappa_host3, box, phi[:nd], theta[:nd], w[:nd], uvb3)
    secs = time.perf_counter() - t0
    return {"value": updates / secs, "unit": "updates/s", "cores": 1, "kind": "port",
            "sample": sample + "; C restatement oracle/ftte_oracle.c, gcc -O2", **ident}


def all_cores_leg(harness, tmp, box, uvb3, phi, theta, w, cores_usable):
    """The reference is serial; directions are independent, so the most a host can get out of its code is one process per
    direction.  Each process holds the reference's 680-byte-per-cell tree (11.4 GB at 256^3), so this leg runs at 128^3
    (1.4 GB per process): P = min(usable cores, 32) processes, 2 directions each, wall time of the slowest."""
    from radiativetransfer_amd import synthetic
    n2, per = 128, 2
    procs = max(1, min(cores_usable, 32))
    try:
        k3, _, _ = synthetic.uniform_workload(n2, 3, seed=12345, tau_median=0.1)
        cases = []
        for p in range(procs):
            lo = (p * per) % max(1, len(phi) - per)
            case = os.path.join(tmp, f"case{p}.bin")
            _write_case(case, n2, box, uvb3, k3, phi[lo:lo + per], theta[lo:lo + per], w[lo:lo + per])
            cases.append(case)
        running = [subprocess.Popen([harness, c, c + ".out"], stdout=subprocess.PIPE, text=True) for c in cases]
        secs = []
        for pr in running:
            out, _ = pr.communicate(timeout=600)
            t = None
            for line in out.splitlines():
                if line.strip().startswith("SWEEP_SECONDS"):
                    t = float(line.split()[1])
            secs.append(t)
        if all(t and t > 0 for t in secs):
            total = n2 ** 3 * 3 * per * procs
            return {"all_cores": {"value": total / max(secs), "unit": "updates/s", "cores": procs,
                                  "sample": f"{procs} concurrent processes of the same harness, {n2}^3 grid, 3 groups, {per} "
                                            f"directions each ({total:.3g} updates), slowest process's sweep time"}}
    except Exception as e:
        print(f"[bench] all-cores leg failed: {e}", file=sys.stderr)
    return {}


def in_process(a):
    """One host thread, one context, N devices: the drop-in's call sequence (ftte_set_grid once, then ftte_diffuse_iteration per
    source iteration) on pinned host arrays.  Prints one JSON line of its own kind."""
    import radiativetransfer_amd as rt
    from radiativetransfer_amd import synthetic
    n, nnu, nd = a.n, a.nnu, a.in_process
    phi, theta, w = directions(a.ndir)
    kappa, uvb, box = synthetic.uniform_workload(n, nnu, seed=12345, tau_median=a.tau_median)
    J = np.empty_like(kappa)
    eng = rt.DiffuseTransfer(devices=[0] * nd if a.same_device else list(range(nd)))
    eng.set_uniform_grid(n, box)
    for kv in a.opt:
        key, value = kv.split("=")
        eng.set_option(key, int(value))
    eng.host_register(kappa)
    eng.host_register(J)
    for _ in range(a.warmup):
        eng.iterate_into(kappa, phi, theta, w, uvb, J)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.iterate_into(kappa, phi, theta, w, uvb, J)
    elapsed = time.perf_counter() - t0
    out = {"metric": "cell·dir·ν updates/sec per iteration through the host-array boundary (PCIe-inclusive), one process over N devices",
           "value": n ** 3 * nnu * a.ndir * a.steps / elapsed, "unit": "updates/s", "n_gpus": nd, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"{n}^3 uniform grid, {nnu} frequency groups, {a.ndir} directions, ftte_diffuse_iteration on pinned host arrays",
                      "devices": "device 0 given %d times (rehearsal)" % nd if a.same_device else list(range(nd)),
                      "frequency_slices": eng.counter("frequency_slices"), "direction_slices": eng.counter("direction_slices"),
                      "combine": eng.multi_info(), "rccl": eng.counter("multi_rccl")}}
    eng.host_unregister(kappa)
    eng.host_unregister(J)
    eng.close()
    print(json.dumps(out, ensure_ascii=False))


def main():
    a = parse()
    if a.measure_traffic:
        return measure_traffic(a)
    if a.in_process > 1:
        return in_process(a)
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
    if a.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from radiativetransfer_amd.distributed import Shard2D
    n, nnu = a.n, a.nnu
    ncell = n ** 3
    total_dirs = a.ndir * world if a.weak else a.ndir
    phi_all, theta_all, w_all = directions(total_dirs)
    kappa_host, uvb_all, box = synthetic.uniform_workload(n, nnu, seed=12345, tau_median=a.tau_median)
    if a.weak:  # round 1's mode: every rank all groups, its own 96 directions, all-reduce
        shard = None
        nu_lo, nu_hi = 0, nnu
        lo = rank * a.ndir
        phi, theta, w = phi_all[lo:lo + a.ndir], theta_all[lo:lo + a.ndir], w_all[lo:lo + a.ndir]
    else:
        shard = Shard2D(rank, world, nnu)
        nu_lo, nu_hi = shard.groups
        phi, theta, w = shard.directions(phi_all, theta_all, w_all)
    nnu_local = nu_hi - nu_lo
    uvb = uvb_all[nu_lo:nu_hi].copy()
    kappa = torch.from_numpy(np.ascontiguousarray(kappa_host[nu_lo:nu_hi])).to(dev)
    J = torch.empty((nnu_local, ncell), dtype=torch.float64, device=dev)
    J_full = torch.empty((nnu, ncell), dtype=torch.float64, device=dev) if (shard and shard.r_nu > 1 and a.exchange == "gather") else None

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
    if a.pair_waves:
        eng.set_option("pair_waves", a.pair_waves)
    if a.tiled >= 0:
        eng.set_option("tiled", a.tiled)
    if a.share >= 0:
        eng.set_option("share", a.share)
    if a.lanes:
        eng.set_option("lanes", a.lanes)
    if a.dataflow >= 0:
        eng.set_option("dataflow", a.dataflow)
    if a.ldspad:
        eng.set_option("ldspad", a.ldspad)
    for kv in a.opt:
        key, value = kv.split("=")
        eng.set_option(key, int(value))
    stream = torch.cuda.current_stream().cuda_stream

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def step(timed=False):
        if timed:
            ev[0].record()
        eng.set_opacity_device(nnu_local, kappa.data_ptr())
        eng.transport_device(phi, theta, w, uvb, J.data_ptr(), stream)
        if timed:
            ev[1].record()
        if world > 1:
            Jx = J.cpu() if a.rehearse_on_one_gpu else J   # rehearsal: gloo moves host tensors
            if shard is None:
                dist.all_reduce(Jx)
            elif a.exchange == "gather":
                shard.combine(Jx, out=None if a.rehearse_on_one_gpu else J_full)
            else:
                shard.exchange(Jx)
        if timed:
            ev[2].record()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    launch_ms, launch_updates, compute_ms, collective_ms = 0.0, 0, 0.0, 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(timed=True)
        # the launch records of a sweep are read after the sweep has drained; the next sweep starts with a stream
        # synchronise anyway (it rewrites device tables), so this adds no bubble of its own
        torch.cuda.synchronize()
        compute_ms += ev[0].elapsed_time(ev[1])
        collective_ms += ev[1].elapsed_time(ev[2])
        for ms, upd in eng.launch_records():
            launch_ms += ms
            launch_updates += upd
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed, compute_ms, collective_ms], dtype=torch.float64, device="cpu" if a.rehearse_on_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, compute_ms, collective_ms = (float(x) for x in t.tolist())

    updates_per_step = ncell * total_dirs * nnu
    value = updates_per_step * a.steps / elapsed
    nlaunch = len(eng.launch_records()) * a.steps
    achieved = launch_updates * BYTES_PER_UPDATE / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0

    if world == 1:
        parallelism = "single GPU"
    elif shard is None:
        parallelism = f"weak: {a.ndir} directions per rank, RCCL all-reduce of J"
    else:
        parallelism = f"{world} ranks = " + shard.describe("combine" if a.exchange == "gather" else "exchange") + " (RCCL)"
    bricks = len(eng.launch_records()) == 1
    form = eng.counter("brick_form") if bricks else -1
    kernel = ("ftte::brick_pair_kernel" if form == 2 else "ftte::brick_kernel") + " (all stage launches of a sweep)" if bricks else "ftte::sweep_kernel"
    pmc = pmc_record(config_key(a, n, nnu, total_dirs, world)) if bricks else None
    traffic = pmc["hbm_bytes_per_launch"] if pmc else None
    avg_launch_ms = launch_ms / nlaunch if nlaunch else None
    # what the counters say the kernel really moved per second: the roofline fraction proper (the 24-byte figure of the contract is
    # not a lower bound for a kernel whose directions share the opacity load and the J store)
    moved = traffic / (avg_launch_ms * 1e-3) / 1e9 if (traffic and avg_launch_ms) else None
    out = {
        "metric": "cell·dir·ν updates/sec per iteration, 256³ grid; achieved HBM GB/s vs peak",
        "value": value, "unit": "updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak" if a.weak else "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n}^3 uniform grid, {nnu} frequency groups, {total_dirs} directions in all "
                               "(NESTED pixels of the rotated HEALPix set), diffuse sweep, log-normal opacity (median cell optical depth "
                               f"{a.tau_median:g} in the lowest group, falling as nu^-3), zero "
                               "emissivity (BASELINE.json configs[" + ("2" if total_dirs == 192 else "1") + "])",
                   "grid": n, "nnu": nnu, "ndir_total": total_dirs, "ndir_this_rank": len(phi), "nnu_this_rank": nnu_local,
                   "parallelism": parallelism,
                   "compute_ms_per_step": compute_ms / a.steps, "collective_ms_per_step": collective_ms / a.steps},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": moved / HBM_PEAK_GBS if moved else None,
                     "frac_contract_24B": achieved / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "traffic_source": (pmc["source"] + " (committed rocprofv3 --pmc passes of this configuration, not this run)") if pmc else None,
                     "kernel": kernel, "launches": nlaunch,
                     "avg_launch_ms": avg_launch_ms,
                     "bytes_per_update": BYTES_PER_UPDATE,
                     "moved_GBs": moved,
                     "valu_busy": pmc.get("valu_busy_fraction") if pmc else None,
                     "valu_instructions_per_update": pmc.get("valu_instructions_per_update") if pmc else None,
                     "note": "achieved = 24 B x updates of rank 0's sweep launches / their HIP-event time: the contract's algorithmic "
                             "figure (SURVEY.md 8(d)); 24 B is what a kernel pays that reads kappa and read-modify-writes J per "
                             "direction, NOT a lower bound for the brick kernel, whose directions share both, so frac_contract_24B "
                             "can pass 1.  frac = moved_GBs / peak, moved_GBs = HBM bytes of the same launches from the PMC counters "
                             "(traffic; FETCH_SIZE doubled per the guide, + WRITE_SIZE) / the same HIP-event time: what the memory "
                             "system delivered.  null where no committed counter pass matches this configuration (options, sharding)."},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cn = a.cpu_n or n
        if cn == n:
            k3 = kappa_host[:3]
        else:
            k3, _, _ = synthetic.uniform_workload(cn, 3, seed=12345, tau_median=0.1)
        out["cpu_baseline"] = cpu_baseline(cn, k3, uvb_all[:3], box, phi_all, theta_all, w_all)
    if rank == 0:
        print(json.dumps(out, ensure_ascii=False))
    if world > 1:
        dist.destroy_process_group()


SWEEP_OPTIONS = ("rows", "slots", "waves", "engine", "chunk", "group", "brick_waves", "tiled", "pair_waves", "team", "share", "dataflow", "ldspad")


def config_key(a, n, nnu, ndir, world):
    """What decides the sweep kernel's traffic: the workload and every option that shapes the sweep (streams do not: a counter pass
    runs with --lanes 1 so that dispatches do not overlap)."""
    opts = {k: getattr(a, k) for k in SWEEP_OPTIONS}
    opts["opt"] = sorted(a.opt)
    return {"grid": n, "nnu": nnu, "ndir": ndir, "world": world, "weak": bool(a.weak), "tau_median": a.tau_median, "options": opts}


def pmc_record(key):
    """The committed counter record (profiles/pmc_traffic.json, written by --measure-traffic) if it was taken for exactly this
    configuration; None otherwise: a bench line never quotes traffic measured on another kernel form or option set."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if rec.get("key") == key and rec.get("hbm_bytes_per_launch"):
            return rec
    except Exception:
        pass
    return None


def measure_traffic(a):
    """rocprofv3 --pmc passes over one sweep of this configuration (separate passes, counters only: FETCH_SIZE, WRITE_SIZE, two SQ
    sets), summed over the sweep kernel's dispatches; FETCH_SIZE doubled (gfx950 tallies 128-byte read requests at 64 B,
    MI355X_MICROARCH.md).  Children of this process; needs a GPU box with rocprofv3."""
    import collections
    import csv
    import glob
    import shutil
    n, nnu, ndir = a.n, a.nnu, a.ndir
    passes = {"FETCH_SIZE": ["FETCH_SIZE"], "WRITE_SIZE": ["WRITE_SIZE"],
              "SQ": ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"],
              "SQ2": ["SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_INSTS_SMEM"]}
    fwd = ["--grid", str(n), "--nnu", str(nnu), "--ndir", str(ndir), "--tau-median", str(a.tau_median), "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--lanes", "1"]
    for k in SWEEP_OPTIONS:
        v = getattr(a, k)
        if v not in (0, -1):
            fwd += ["--" + k.replace("_", "-"), str(v)]
    for kv in a.opt:
        fwd += ["--opt", kv]
    out = os.path.join(ROOT, "gpurun_out", "pmc_traffic")
    shutil.rmtree(out, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    totals, ndisp = {}, 0
    for name, counters in passes.items():
        d = os.path.join(out, name)
        cmd = ["rocprofv3", "--pmc", *counters, "--kernel-include-regex", "brick_|sweep_kernel", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               "python3", os.path.abspath(__file__), *fwd]
        res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
        if res.returncode:
            sys.exit(f"[bench] counter pass {name} failed:\n{res.stderr[-2000:]}")
        tot, disp = collections.defaultdict(float), set()
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                disp.add(r["Dispatch_Id"])
        totals.update(tot)
        ndisp = max(ndisp, len(disp))
    updates = n ** 3 * nnu * ndir
    rec = {"key": config_key(a, n, nnu, ndir, 1), "source": "profiles/pmc_traffic.json", "command": "bench.py --measure-traffic " + " ".join(fwd),
           "unit": "one sweep = all sweep-kernel dispatches of an iteration (bench.py's launch record)", "dispatches_per_sweep": ndisp,
           "updates_per_sweep": updates, "counters": dict(totals)}
    rec["fetch_bytes_x2"] = 2 * 1024 * totals["FETCH_SIZE"]      # KB; doubled per the guide
    rec["write_bytes"] = 1024 * totals["WRITE_SIZE"]
    rec["hbm_bytes_per_launch"] = rec["fetch_bytes_x2"] + rec["write_bytes"]
    rec["hbm_bytes_per_update"] = rec["hbm_bytes_per_launch"] / updates
    cycles = totals["GRBM_GUI_ACTIVE"] / 8
    rec["valu_instructions_per_update"] = totals["SQ_INSTS_VALU"] * 64 / updates
    rec["salu_instructions_per_update"] = totals["SQ_INSTS_SALU"] * 64 / updates
    rec["valu_busy_fraction"] = totals["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cycles
    rec["mean_waves_per_simd"] = totals["SQ_WAVE_CYCLES"] * 4 / 1024 / cycles
    rec["wave_time_in_waitcnt"] = totals["SQ_WAIT_ANY"] / totals["SQ_WAVE_CYCLES"]
    rec["wave_time_waiting_to_issue"] = totals["SQ_WAIT_INST_ANY"] / totals["SQ_WAVE_CYCLES"]
    rec["wave_time_issuing"] = totals["SQ_ACTIVE_INST_ANY"] / totals["SQ_WAVE_CYCLES"]
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    json.dump(rec, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "counters"}, indent=1))


if __name__ == "__main__":
    main()
