#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_brick_gpu.py tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { tag=$1; shift; python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "value %.3g"%r["value"], "frac %.3f"%r["roofline"]["frac"], "sweep phase ms/step %.2f"%(r["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-500:])
PY
}
run default
run nnu1_auto --nnu 1
run nnu1_l1 --nnu 1 --lanes 1
run nnu1_l3 --nnu 1 --lanes 3
run nnu1_l4 --nnu 1 --lanes 4
run nnu1_l4_c8 --nnu 1 --lanes 4 --chunk 8
run nnu1_l2_g3 --nnu 1 --group 3
run nnu2_auto --nnu 2
run nnu2_l4 --nnu 2 --lanes 4
run nnu4_auto --nnu 4
