#!/bin/bash
# Round 3, first look at the persistent form of the brick sweep (option dataflow = 3: a task queue per XCD): parity tests of the
# forms, then the headline workload and the per-rank shapes with a launch per stage (0), one launch with flags (1) and the persistent form.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_persistent
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_brick_gpu.py -x -q -m gpu -k "one_launch or brick_order" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
for nnu in 8 4 2 1; do
  for df in 0 1 3; do
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --nnu $nnu --dataflow $df > $OUT/bench_nnu${nnu}_df$df.json 2> $OUT/bench_nnu${nnu}_df$df.err || { echo "nnu $nnu df $df failed"; tail -5 $OUT/bench_nnu${nnu}_df$df.err; exit 1; }
    python - <<P
import json
d=json.load(open("$OUT/bench_nnu${nnu}_df$df.json"))
print("nnu $nnu dataflow $df: step %.2f ms, sweep phase %.2f ms, value %.3e" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["value"]))
P
  done
done
