#!/bin/bash
# Round 3: the one- and two-group rank shapes, stage form: directions per group x layers per brick x accumulator sharing
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_nnu1_groups
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for nnu in 1 2; do
for group in 2 3 4; do
for chunk in 4 8; do
for share in 1 2; do
    timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --nnu $nnu --group $group --chunk $chunk --share $share > $OUT/b.json 2> $OUT/b.err || { echo "failed"; tail -3 $OUT/b.err; continue; }
    python - <<P
import json
d=json.load(open("$OUT/b.json"))
print("nnu $nnu group $group chunk $chunk share $share: step %.2f ms, sweep phase %.2f ms" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
P
done; done; done; done
