#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02z
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
run c32 --chunk 32
run g4 --group 4
run g4_c32 --group 4 --chunk 32
run g5_c32 --group 5 --chunk 32
run g6_c32 --group 6 --chunk 32
run l3 --lanes 3
run nnu1 --nnu 1
run nnu1_c8 --nnu 1 --chunk 8
run nnu2 --nnu 2
