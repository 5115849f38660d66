#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_distributed_gloo.py -x -q -m gpu 2>&1 | tail -3
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json
run() { tag=$1; shift; python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "value %.3g"%r["value"], "frac %.3f"%r["roofline"]["frac"], "sweep phase ms/step %.2f"%(r["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-500:])
PY
}
# what one rank of 8 / 4 / 2 runs under frequency-major sharding, and the direction-split alternative
for C in 4 8 16; do for G in 2 3 4; do
run nnu1_c${C}_g${G} --nnu 1 --chunk $C --group $G
done; done
run nnu2_c8 --nnu 2 --chunk 8
run nnu2_c16 --nnu 2 --chunk 16
run nnu4_c16 --nnu 4 --chunk 16
run nnu4_c8 --nnu 4 --chunk 8
run nnu8_d12_c16 --nnu 8 --ndir 12 --chunk 16
run nnu8_d12_c8 --nnu 8 --ndir 12 --chunk 8
run nnu8_d192 --nnu 8 --ndir 192
run nnu8_d192_g4 --nnu 8 --ndir 192 --group 4
