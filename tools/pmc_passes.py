#!/usr/bin/env python3
"""rocprofv3 --pmc passes over any of the repository's bench scripts: FETCH_SIZE, WRITE_SIZE and two SQ sets, each in a run of its
own (counters only, never with a trace domain), summed per kernel.  Children of this process; run it on the GPU box.

    python tools/pmc_passes.py --out profiles/r03_config4_pmc.json [--kernels REGEX] -- tools/bench_config4.py 128

What it writes, per kernel whose name matches: dispatches, HBM bytes (FETCH_SIZE doubled: gfx950 tallies 128-byte read requests at
64 B, MI355X_MICROARCH.md; + WRITE_SIZE), vector and scalar instructions, and -- from the cycles the kernel's dispatches were
active -- how busy the vector units were, how many wavefronts a SIMD held and where a resident wavefront's time went."""
import argparse, collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {"FETCH_SIZE": ["FETCH_SIZE"], "WRITE_SIZE": ["WRITE_SIZE"],
          "SQ": ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"],
          "SQ2": ["SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_INSTS_SMEM"]}

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True)
ap.add_argument("--kernels", default="ftte::")
ap.add_argument("--note", default="")
ap.add_argument("cmd", nargs=argparse.REMAINDER)
a = ap.parse_args()
cmd = [c for c in a.cmd if c != "--"]
scratch = os.path.join(ROOT, "gpurun_out", "pmc_" + os.path.splitext(os.path.basename(a.out))[0])
shutil.rmtree(scratch, ignore_errors=True)
per = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for name, counters in PASSES.items():
    d = os.path.join(scratch, name)
    full = ["rocprofv3", "--pmc", *counters, "--kernel-include-regex", a.kernels, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
            "python3", os.path.join(ROOT, cmd[0]), *cmd[1:]]
    res = subprocess.run(full, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=1100)
    if res.returncode:
        sys.exit(f"pass {name} failed:\n{res.stderr[-3000:]}")
    open(os.path.join(scratch, name + ".log"), "w").write(res.stdout[-4000:])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if name == "SQ2":
                disp[k].add(r["Dispatch_Id"])
out = {"command": " ".join(cmd), "note": a.note, "kernels": {}}
for k, c in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    rec = {"dispatches": len(disp[k]), "counters": dict(c)}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rec["hbm_bytes"] = 2 * 1024 * c["FETCH_SIZE"] + 1024 * c["WRITE_SIZE"]
        rec["hbm_read_bytes_x2"], rec["hbm_write_bytes"] = 2 * 1024 * c["FETCH_SIZE"], 1024 * c["WRITE_SIZE"]
    if c.get("GRBM_GUI_ACTIVE"):
        cycles = c["GRBM_GUI_ACTIVE"] / 8          # summed over the 8 XCDs
        rec["active_cycles"] = cycles
        rec["valu_busy_fraction"] = c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cycles
        rec["mean_waves_per_simd"] = c.get("SQ_WAVE_CYCLES", 0) * 4 / 1024 / cycles
        if rec.get("hbm_bytes"):
            rec["hbm_bytes_per_cycle"] = rec["hbm_bytes"] / cycles
    if c.get("SQ_WAVE_CYCLES"):
        rec["wave_time_in_waitcnt"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
        rec["wave_time_waiting_to_issue"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
        rec["wave_time_issuing"] = c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    out["kernels"][k] = rec
json.dump(out, open(os.path.join(ROOT, a.out), "w"), indent=1)
for k, rec in out["kernels"].items():
    print(f"{k[:70]:70s} disp {rec['dispatches']:5d} hbm {rec.get('hbm_bytes', 0) / 1e9:8.2f} GB valu busy {rec.get('valu_busy_fraction', 0):.2f} "
          f"waves/SIMD {rec.get('mean_waves_per_simd', 0):.2f} wait {rec.get('wave_time_in_waitcnt', 0):.2f}")
