#!/bin/bash
# Round 3: bricks of FOUR rows (kBrickRows = 4 compiled in) on the small shapes: twice the bricks per diagonal, half a brick's latency
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_rows4
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_brick_gpu.py -x -q -m gpu -k "izone or one_launch" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
line() {
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "$* failed"; tail -5 $OUT/b.err; return; }
    python - "$*" <<P
import json, sys
d=json.load(open("$OUT/b.json"))
print("%-44s step %6.2f ms, sweep phase %6.2f ms" % (sys.argv[1], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
P
}
line --nnu 1
line --nnu 1 --team 0
line --nnu 1 --chunk 8
line --nnu 1 --chunk 8 --team 0
line --nnu 1 --chunk 8 --team 0 --group 3
line --nnu 2
line --nnu 2 --team 0
line --nnu 2 --chunk 16 --team 0
line --nnu 4
line --nnu 8
line --grid 128
line --grid 128 --team 0
