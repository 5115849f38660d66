#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_brick_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { tag=$1; shift; python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "frac %.3f"%r["roofline"]["frac"], "launches", r["roofline"]["launches"], "sum kernel ms/step %.2f"%(r["roofline"]["avg_launch_ms"]*r["roofline"]["launches"]/4), flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-500:])
PY
}
for G in 2 3 4; do for C in 16 32; do
run b_c${C}_g${G} --engine 2 --chunk $C --group $G
done; done
