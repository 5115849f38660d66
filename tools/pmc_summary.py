#!/usr/bin/env python3
"""Mean per-launch value of every counter rocprofv3 --pmc collected for kernels matching a substring."""
import collections
import csv
import glob
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else "sweep_kernel"
agg = collections.defaultdict(list)
dur = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items()):
    print(f"{k:32s} n={len(v):3d} mean={sum(v) / len(v):.6g}")
if dur:
    print(f"{'duration_ns':32s} n={len(dur):3d} mean={sum(dur) / len(dur):.6g}")
