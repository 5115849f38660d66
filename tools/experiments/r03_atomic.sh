#!/bin/bash
# Round 3: later visitors of an accumulator add with fp64 atomics (option atomic_acc) instead of read-add-store.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_atomic
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for args in "--nnu 8" "--nnu 8 --opt atomic_acc=1" "--nnu 8 --dataflow 3 --opt queue_mix=2 --opt atomic_acc=1" "--nnu 8 --team 2 --opt atomic_acc=1" "--nnu 4" "--nnu 4 --opt atomic_acc=1" "--nnu 1" "--nnu 1 --opt atomic_acc=1"; do
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $args > $OUT/b.json 2> $OUT/b.err || { echo "$args failed"; tail -5 $OUT/b.err; exit 1; }
    python - <<P
import json
d=json.load(open("$OUT/b.json"))
print("$args: step %.2f ms, sweep phase %.2f ms, value %.3e" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["value"]))
P
done
