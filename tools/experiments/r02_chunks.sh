#!/bin/bash
# sweep phase against the brick length, with the accumulator pairing off everywhere (lengths that do not divide 256 cannot pair)
OUT=$GRAFT_REPO_ROOT/gpurun_out/chunks
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
run c16_s2 --chunk 16
for c in 12 16 18 20 22 24 26 28 32; do run c${c}_s1 --chunk $c --share 1; done
