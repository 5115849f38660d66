#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02r
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_brick_gpu.py -x -q -m gpu -k "one_launch" 2>&1 | tail -5 || exit 1
run() { tag=$1; shift; timeout -k 10 120 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "frac %.3f"%r["roofline"]["frac"], "sweep phase ms/step %.2f"%(r["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-300:])
PY
}
run stages --dataflow 0
run through --dataflow 2
run through_g4 --dataflow 2 --group 4
run through_g6 --dataflow 2 --group 6
run through_nnu1 --dataflow 2 --nnu 1
run through_c8 --dataflow 2 --chunk 8
run flags
run flags_g4 --group 4
