#!/bin/bash
# Round 3: the merge with the loads of up to eight accumulators in flight: parity subset, then the bench line (step - sweep phase) and the kernel's own time
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_merge
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_brick_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "not baseline_size and not config3 and not every_rank" > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
for i in 1 2; do timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/b.json 2>/dev/null; python - <<P
import json
d=json.load(open("$OUT/b.json"))
print("step %.2f ms, sweep phase %.2f ms, rest %.2f ms" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["ms_per_step"]-d["roofline"]["avg_launch_ms"]))
P
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o m -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
find $OUT/t -name "*kernel_stats.csv" -exec grep -i "merge\|set_layouts" {} \; | cut -c1-160
rm -rf $OUT/t
