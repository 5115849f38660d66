#!/bin/bash
# Round 3: the thin test on tau itself (no range reduction in a wavefront of thin segments): parity, then the headline and the per-rank shapes.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_thin
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_brick_gpu.py tests/test_parity_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
for nnu in 8 4 2 1; do
  for df in 0 3; do
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --nnu $nnu --dataflow $df --opt queue_mix=2 > $OUT/bench_nnu${nnu}_df$df.json 2> $OUT/bench_nnu${nnu}_df$df.err || { echo "nnu $nnu df $df failed"; tail -5 $OUT/bench_nnu${nnu}_df$df.err; exit 1; }
    python - <<P
import json
d=json.load(open("$OUT/bench_nnu${nnu}_df$df.json"))
print("nnu $nnu dataflow $df: step %.2f ms, sweep phase %.2f ms, value %.3e" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["value"]))
P
  done
done
