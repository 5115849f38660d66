#!/bin/bash
# brick engine: first timings over chunk / group / waves
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "frac %.3f"%r["roofline"]["frac"], "launches", r["roofline"]["launches"], "avg ms %.3f"%r["roofline"]["avg_launch_ms"], flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-500:])
PY
}
run tiles --engine 1
run b_c32_g4_w2 --engine 2 --chunk 32 --group 4 --brick-waves 2
run b_c32_g4_w3 --engine 2 --chunk 32 --group 4 --brick-waves 3
run b_c32_g4_w4 --engine 2 --chunk 32 --group 4 --brick-waves 4
run b_c16_g4_w2 --engine 2 --chunk 16 --group 4 --brick-waves 2
run b_c64_g4_w2 --engine 2 --chunk 64 --group 4 --brick-waves 2
run b_c32_g2_w2 --engine 2 --chunk 32 --group 2 --brick-waves 2
run b_c32_g3_w2 --engine 2 --chunk 32 --group 3 --brick-waves 2
run b_c32_g6_w2 --engine 2 --chunk 32 --group 6 --brick-waves 2
run b_c32_g8_w2 --engine 2 --chunk 32 --group 8 --brick-waves 2
run b_c32_g1_w2 --engine 2 --chunk 32 --group 1 --brick-waves 2
