#!/bin/bash
# Round 3: the one-group rank shape on one stream instead of two (a launch per stage holding every group's bricks)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_nnu1_lanes
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
line() {
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "$* failed"; tail -5 $OUT/b.err; return; }
    python - "$*" <<P
import json, sys
d=json.load(open("$OUT/b.json"))
print("%-52s step %6.2f ms, sweep phase %6.2f ms" % (sys.argv[1], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
P
}
line --nnu 1 --lanes 1
line --nnu 1 --lanes 2
line --nnu 1 --lanes 1 --team 0
line --nnu 1 --lanes 1 --team 0 --opt ablate=63
line --nnu 1 --lanes 2 --team 0 --opt ablate=63
line --nnu 1 --lanes 1 --share 0
line --nnu 1 --lanes 1 --chunk 2
line --nnu 2 --lanes 1
line --nnu 2 --lanes 2
