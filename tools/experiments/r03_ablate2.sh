#!/bin/bash
# Round 3: the brick kernel with more and more of its memory traffic left out (option "ablate"; J is wrong, timing only), a launch
# per stage and the persistent form: what does the instruction stream alone take?
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_ablate2
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for df in 0 3; do
for ab in 0 15 31 47 63; do
    timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --dataflow $df --opt queue_mix=2 --opt ablate=$ab > $OUT/b_$ab.json 2> $OUT/b_$ab.err || { echo failed; tail -5 $OUT/b_$ab.err; exit 1; }
    python -c "
import json; d=json.load(open('$OUT/b_$ab.json')); print('dataflow $df ablate $ab: step %.2f ms, sweep phase %.2f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
done
