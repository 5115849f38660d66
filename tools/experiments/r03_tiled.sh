#!/bin/bash
# Round 3: brick-ordered storage of opacities and accumulators (option tiled, built in round 2 when the kernel had 52.8 instructions per
# update and found to change nothing) on the round-3 kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_tiled
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for args in "" "--tiled 1" "--tiled 2" "--tiled 1 --dataflow 3 --opt queue_mix=2" "--dataflow 3 --opt queue_mix=2" "--tiled 2 --chunk 8" ""; do
    timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline $args > $OUT/b.json 2> $OUT/b.err || { echo "$args failed"; tail -3 $OUT/b.err; continue; }
    python - "$args" <<P
import json, sys
d=json.load(open("$OUT/b.json"))
print("%-44s step %6.2f ms, sweep phase %6.2f ms" % (sys.argv[1], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
P
done
