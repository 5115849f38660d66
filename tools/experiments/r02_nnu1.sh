#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/nnu1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; timeout -k 10 150 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --nnu 1 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "sweep phase ms/step %.2f"%(r["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-300:])
PY
}
run auto
run g1_c4 --group 1 --chunk 4
run g1_c8 --group 1 --chunk 8
run g1_c16 --group 1 --chunk 16
run g2_c4_s1 --group 2 --chunk 4 --share 1
run g2_c4_s0 --group 2 --chunk 4 --share 0
run g2_c6 --group 2 --chunk 6
run g2_c4_l1 --group 2 --chunk 4 --lanes 1
