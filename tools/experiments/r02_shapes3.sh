#!/bin/bash
# what a rank of a frequency-sharded run sweeps: one launch with flags (dataflow 2) against a launch per stage, longer bricks
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02u
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
run nnu1_auto --nnu 1
run nnu1_df2_c4 --nnu 1 --dataflow 2
run nnu1_df2_c8 --nnu 1 --dataflow 2 --chunk 8
run nnu1_df2_c16 --nnu 1 --dataflow 2 --chunk 16
run nnu1_df2_c8_g3 --nnu 1 --dataflow 2 --chunk 8 --group 3
run nnu1_df2_c16_g3 --nnu 1 --dataflow 2 --chunk 16 --group 3
run nnu2_auto --nnu 2
run nnu2_df2_c8 --nnu 2 --dataflow 2 --chunk 8
run nnu2_df2_c16 --nnu 2 --dataflow 2 --chunk 16
run nnu4_auto --nnu 4
run nnu4_df2 --nnu 4 --dataflow 2
run nnu1_d192 --nnu 1 --ndir 192
run nnu1_d192_df2_c8 --nnu 1 --ndir 192 --dataflow 2 --chunk 8
