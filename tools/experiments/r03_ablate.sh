#!/bin/bash
# Round 3: which of the brick kernel's memory accesses does the sweep wait for?  Diagnostic option "ablate" leaves parts out (J is
# wrong then; timing only): 1 rays from the left / below, 2 rays to the right / above, 4 the earlier J of shared accumulators,
# 8 the chunk faces.  Headline workload, a launch per stage.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_ablate
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for ab in 0 1 2 3 4 8 7 15; do
    timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --opt ablate=$ab > $OUT/b_$ab.json 2> $OUT/b_$ab.err || { echo failed; tail -5 $OUT/b_$ab.err; exit 1; }
    python -c "
import json; d=json.load(open('$OUT/b_$ab.json')); print('ablate $ab: step %.2f ms, sweep phase %.2f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
