#!/bin/bash
# Round 3: how long the persistent form's workgroups wait for the bricks they depend on (polls per task; a poll is s_sleep 20,
# about half a microsecond), one frequency group (the shape of a rank of an 8-GPU run) against eight
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_polls
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export FTTE_QUEUE_STATS=1
for args in "--nnu 1 --dataflow 3 --share 0" "--nnu 1 --dataflow 3 --share 0 --chunk 2" "--nnu 1 --dataflow 3 --share 0 --chunk 8" "--nnu 1 --dataflow 3 --share 2" "--nnu 2 --dataflow 3 --share 0" "--nnu 8 --dataflow 3"; do
    timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline $args > $OUT/b.json 2> $OUT/b.err || { echo "$args failed"; tail -5 $OUT/b.err; exit 1; }
    python - <<P
import json
d=json.load(open("$OUT/b.json"))
print("$args: step %.2f ms, sweep phase %.2f ms" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
P
    grep "queue" $OUT/b.err | tail -8
done
