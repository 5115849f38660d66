#!/bin/bash
# Round 3: the one- and two-group rank shapes: streams x chunk x accumulator sharing x atomic accumulation.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_shapes
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for nnu in 1 2; do
for lanes in 2 3 4; do
for chunk in 4 8; do
for extra in "" "--opt atomic_acc=1" "--share 1 --opt atomic_acc=1"; do
    timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --nnu $nnu --lanes $lanes --chunk $chunk $extra > $OUT/b.json 2> $OUT/b.err || { echo "failed"; tail -3 $OUT/b.err; continue; }
    python - <<P
import json
d=json.load(open("$OUT/b.json"))
print("nnu $nnu lanes $lanes chunk $chunk $extra: step %.2f ms, sweep phase %.2f ms" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
P
done; done; done; done
