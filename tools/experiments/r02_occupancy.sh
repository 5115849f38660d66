#!/bin/bash
# how the brick sweep responds to residency: extra dynamic LDS per workgroup (one wavefront) on top of the 12 KB of ray state
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02w
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; timeout -k 10 150 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "sweep phase ms/step %.2f"%(r["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-300:])
PY
}
run pad0
run pad1k --ldspad 1024      # 13 KB: 12 waves per CU
run pad4k --ldspad 4096      # 16 KB: 10
run pad8k --ldspad 8192      # 20 KB: 8
run pad14k --ldspad 14336    # 26 KB: 6
run g2_pad0 --group 2        # 8 KB: 16 (VGPR limit)
run g2_pad2k --group 2 --ldspad 2048   # 10 KB: 16
run g2_pad4k --group 2 --ldspad 4096   # 12 KB: 13
run g2_pad8k --group 2 --ldspad 8192   # 16 KB: 10
