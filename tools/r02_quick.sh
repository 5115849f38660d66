#!/bin/bash
# brick tests, then the bench line of the default configuration (and whatever flags follow)
OUT=$GRAFT_REPO_ROOT/gpurun_out/quick
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_brick_gpu.py tests/test_hybrid_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
run() { tag=$1; shift; timeout -k 10 150 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "sweep phase ms/step %.2f"%(r["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-300:])
PY
}
run default
run default2
run g4 --group 4
run nnu1 --nnu 1
