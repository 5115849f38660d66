#!/bin/bash
# Round 3: what the one-group rank shape waits for: the solo kernel with the accesses of the layer loop switched off one after the other
# (option ablate: wrong J, timing only), stage form
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_nnu1_ablate
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for ab in 0 1 3 7 15 31 63; do
    timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --nnu 1 --team 0 --opt ablate=$ab > $OUT/b.json 2> $OUT/b.err || { echo "failed"; tail -3 $OUT/b.err; continue; }
    python - <<P
import json
d=json.load(open("$OUT/b.json"))
print("nnu 1 solo ablate $ab: step %.2f ms, sweep phase %.2f ms" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
P
done
