#!/bin/bash
# Round 3: the one-group rank shape (what an 8-rank run executes) in the persistent form: accumulator sharing decides how evenly the
# queues can be filled (units are (frequency group, accumulator) pairs), chunk how long a brick is.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_nnu1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export FTTE_QUEUE_STATS=1
for nnu in 1 2; do
for args in "--dataflow 0" "--dataflow 3 --share 1" "--dataflow 3 --share 0" "--dataflow 3 --share 1 --chunk 8" "--dataflow 3 --share 0 --chunk 8" "--dataflow 3 --share 0 --chunk 16" "--dataflow 3 --share 0 --chunk 8 --group 2"; do
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --nnu $nnu $args > $OUT/b.json 2> $OUT/b.err || { echo "$args failed"; tail -5 $OUT/b.err; exit 1; }
    python - <<P
import json
d=json.load(open("$OUT/b.json"))
print("nnu $nnu $args: step %.2f ms, sweep phase %.2f ms, value %.3e" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["value"]))
P
    grep "queue" $OUT/b.err | tail -8 | awk '{printf "%s ", $8}'; echo
done
done
