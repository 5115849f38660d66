#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02p
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
for G in 3 4 6; do for C in 8 16; do
run g${G}_c${C}_w2 --group $G --chunk $C --brick-waves 2
done; done
run g3_c16_w3 --group 3 --chunk 16 --brick-waves 3
run g4_c8_w2_l3 --group 4 --chunk 8 --brick-waves 2 --lanes 3
run g4_c8_w2_l4 --group 4 --chunk 8 --brick-waves 2 --lanes 4
run nnu1_w2 --nnu 1 --brick-waves 2
