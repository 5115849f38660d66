#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "frac %.3f"%r["roofline"]["frac"], "sweep phase ms/step %.2f"%(r["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-500:])
PY
}
for C in 8 16 24 32; do for G in 3 4; do for LN in 2 3; do
run s_c${C}_g${G}_l${LN} --engine 2 --chunk $C --group $G --lanes $LN
done; done; done
run s_c16_g3_l2_s1 --engine 2 --chunk 16 --group 3 --lanes 2 --share 1
run s_c16_g3_l2_s0 --engine 2 --chunk 16 --group 3 --lanes 2 --share 0
run s_c16_g5_l2 --engine 2 --chunk 16 --group 5 --lanes 2
