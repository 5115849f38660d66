#!/bin/bash
# Round 3: quick A/B of a kernel change: a handful of parity tests, then the headline at three and four waves per SIMD, pair form, small shapes.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_quick_${1:-x}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_brick_gpu.py -x -q -m gpu -k "every_izone or goldens or pair or one_launch" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for args in "--nnu 8" "--nnu 8 --brick-waves 3" "--nnu 8 --team 2" "--nnu 8 --dataflow 3 --opt queue_mix=2" "--nnu 4" "--nnu 4 --team 0" "--nnu 1" ; do
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $args > $OUT/b.json 2> $OUT/b.err || { echo "$args failed"; tail -5 $OUT/b.err; exit 1; }
    python - <<P
import json
d=json.load(open("$OUT/b.json"))
print("$args: step %.2f ms, sweep phase %.2f ms, value %.3e" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["value"]))
P
done
