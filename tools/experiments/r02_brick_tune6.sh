#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_brick_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { tag=$1; shift; python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "frac %.3f"%r["roofline"]["frac"], "sweep phase ms/step %.2f"%(r["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-300:])
PY
}
for G in 2 3 4 6 8; do for C in 16 32; do
run g${G}_c${C} --group $G --chunk $C
done; done
run g4_c16_l1 --group 4 --chunk 16 --lanes 1
run g6_c16_l1 --group 6 --chunk 16 --lanes 1
run g4_c16_w2 --group 4 --chunk 16 --brick-waves 2
run nnu1 --nnu 1
run nnu1_g4 --nnu 1 --group 4
