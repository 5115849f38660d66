#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "frac %.3f"%r["roofline"]["frac"], "sweep phase ms/step %.2f"%(r["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-300:])
PY
}
run default
run g3_c8 --group 3 --chunk 8
run g4_c16 --group 4 --chunk 16
run nnu1 --nnu 1
run nnu2 --nnu 2
