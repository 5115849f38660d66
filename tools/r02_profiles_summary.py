#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/r02_profiles.sh) into the tracked summaries under profiles/: kernel-stats CSVs, the bench lines,
and one JSON with the brick kernel's counters per sweep (all its stage launches of one iteration summed)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02a"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def stats(sub, name):
    hits = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))


for sub, name in (("bench", "bench"), ("bench_l1", "bench_one_stream"), ("config4", "config4"), ("config5", "config5"), ("loop", "loop"), ("point", "point")):
    stats(sub, name)
for f in ("bench_plain.json", "bench.json", "bench_l1.json"):
    p = os.path.join(src, f)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{f}"))
for f in ("config4.log", "config5.log", "loop.log", "point.log"):
    p = os.path.join(src, f)
    if os.path.exists(p):
        lines = [ln for ln in open(p, errors="replace") if "rocprofv3" not in ln and "amdgpu.ids" not in ln and not ln.startswith(("W20", "E20", "I20"))]
        open(os.path.join(dst, f"{tag}_{f}"), "w").writelines(lines)


def counters(sub, pattern="brick_kernel"):
    tot, disp = collections.defaultdict(float), set()
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pattern in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                disp.add(r["Dispatch_Id"])
    return dict(tot), len(disp)


n, nnu, ndir = 256, 8, 96
updates = n ** 3 * nnu * ndir
rec = {"round": tag, "grid": n, "nnu": nnu, "ndir": ndir, "kernel": "ftte::brick_kernel<4, 0>, all stage launches of one sweep (bench.py --steps 1 --lanes 1)",
       "updates_per_sweep": updates, "algorithmic_bytes_per_sweep": 24 * updates}
fetch, nd = counters("FETCH_SIZE")
write, _ = counters("WRITE_SIZE")
if fetch and write:
    # FETCH_SIZE / WRITE_SIZE are reported in KB; gfx950 tallies 128-B read requests at 64 B: doubled (MI355X_MICROARCH.md, HBM)
    rec.update({"launches_per_sweep": nd, "FETCH_SIZE_KB": fetch["FETCH_SIZE"], "WRITE_SIZE_KB": write["WRITE_SIZE"],
                "fetch_bytes_x2": 2 * 1024 * fetch["FETCH_SIZE"], "write_bytes": 1024 * write["WRITE_SIZE"]})
    rec["hbm_bytes_per_launch"] = rec["fetch_bytes_x2"] + rec["write_bytes"]   # "launch" = bench.py's unit: one sweep
    rec["hbm_bytes_per_update"] = rec["hbm_bytes_per_launch"] / updates
    merge_f, _ = counters("FETCH_SIZE", "merge_kernel")
    merge_w, _ = counters("WRITE_SIZE", "merge_kernel")
    if merge_f:
        rec["merge_kernel_bytes"] = 2 * 1024 * merge_f["FETCH_SIZE"] + 1024 * merge_w.get("WRITE_SIZE", 0.0)
sq, _ = counters("SQ")
sq2, _ = counters("SQ2")
if sq:
    rec["sq_per_sweep"] = sq
    rec["valu_instructions_per_update"] = sq["SQ_INSTS_VALU"] * 64 / updates
    cycles = sq["GRBM_GUI_ACTIVE"] / 8
    rec["valu_busy_fraction"] = sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cycles
    rec["mean_waves_per_simd"] = sq["SQ_WAVE_CYCLES"] * 4 / 1024 / cycles
    rec["gpu_cycles_per_sweep"] = cycles
    if sq.get("SQ_WAVE_CYCLES"):  # of the time a wavefront is resident: waiting on a counter, waiting to issue, issuing
        rec["wave_time_in_waitcnt"] = sq.get("SQ_WAIT_ANY", 0.0) / sq["SQ_WAVE_CYCLES"]
        rec["wave_time_waiting_to_issue"] = sq.get("SQ_WAIT_INST_ANY", 0.0) / sq["SQ_WAVE_CYCLES"]
        rec["wave_time_issuing"] = sq.get("SQ_ACTIVE_INST_ANY", 0.0) / sq["SQ_WAVE_CYCLES"]
if sq2:
    rec["sq2_per_sweep"] = sq2
    rec["salu_instructions_per_update"] = sq2.get("SQ_INSTS_SALU", 0) * 64 / updates
json.dump(rec, open(os.path.join(dst, f"{tag}_pmc_brick_kernel.json"), "w"), indent=1)
if "hbm_bytes_per_launch" in rec:
    json.dump({"grid": n, "nnu": nnu, "hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "source": f"profiles/{tag}_pmc_brick_kernel.json",
               "unit": "one sweep = all brick_kernel stage launches of an iteration (bench.py's launch record)"},
              open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in rec.items() if not isinstance(v, dict)}, indent=1))
