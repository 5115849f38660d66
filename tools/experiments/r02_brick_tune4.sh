#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_brick_gpu.py -x -q -m gpu -k "goldens or full_direction or records" 2>&1 | tail -3
run() { tag=$1; shift; python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "frac %.3f"%r["roofline"]["frac"], "sweep phase ms/step %.2f"%(r["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-500:])
PY
}
for LN in 1 2 4 8; do for G in 2 3 4; do
run s_c32_g${G}_l${LN} --engine 2 --chunk 32 --group $G --lanes $LN
done; done
run s_c16_g3_l4 --engine 2 --chunk 16 --group 3 --lanes 4
run s_c16_g4_l4 --engine 2 --chunk 16 --group 4 --lanes 4
run s_c64_g4_l4 --engine 2 --chunk 64 --group 4 --lanes 4
run s_c32_g6_l4 --engine 2 --chunk 32 --group 6 --lanes 4
run t_c32_g4_l4 --engine 2 --chunk 32 --group 4 --lanes 4 --team 1
run t_c32_g8_l4 --engine 2 --chunk 32 --group 8 --lanes 4 --team 1
