#!/bin/bash
# Round 3: the shapes a rank of a 1-, 2-, 4-, 8-rank run of the benchmark sweeps (8, 4, 2, 1 frequency groups x 96 directions), the
# direction-split alternative at 8 ranks (8 groups x 12 directions), and the persistent form on the small shapes
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_rank_shapes
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
line() {
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "$* failed"; tail -5 $OUT/b.err; return; }
    python - "$*" <<P
import json, sys
d=json.load(open("$OUT/b.json"))
print("%-44s step %6.2f ms, sweep phase %6.2f ms, %.3e updates/s" % (sys.argv[1], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["value"]))
P
}
line --nnu 8
line --nnu 4
line --nnu 2
line --nnu 1
line --nnu 8 --ndir 12
line --nnu 2 --dataflow 3 --share 0
line --nnu 1 --dataflow 3 --share 0
line --nnu 1 --dataflow 3 --share 0 --chunk 2
