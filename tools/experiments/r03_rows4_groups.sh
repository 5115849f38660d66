#!/bin/bash
# Round 3: bricks of four rows park 2 KB per direction instead of 4: larger groups of directions at the same residency
# (fewer visits of every cell: less opacity and J traffic), against twice the v-face bytes.  kBrickRows = 4 compiled in.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_rows4_groups
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
line() {
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "$* failed"; tail -3 $OUT/b.err; return; }
    python - "$*" <<P
import json, sys
d=json.load(open("$OUT/b.json"))
print("%-36s step %6.2f ms, sweep phase %6.2f ms" % (sys.argv[1], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
P
}
for g in 3 4 5 6 8; do for c in 16 32; do line --group $g --chunk $c; done; done
line --group 6 --chunk 16 --share 1
line --group 6 --chunk 16 --brick-waves 3
