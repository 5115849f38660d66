#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02y
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
run w4
run w5 --brick-waves 5
run w5_l3 --brick-waves 5 --lanes 3
run w4_l3 --lanes 3
