#!/bin/bash
# Round 3: where an iteration of configs[3] goes with the refined block swept by fine bricks: kernel trace, per kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_c4_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o c4 -- python3 $GRAFT_REPO_ROOT/tools/bench_config4.py 128 --no-reference > $OUT/c4.log 2>&1
find $OUT/t -name "*kernel_stats.csv" -exec cp {} $OUT/c4_kernel_stats.csv \;
grep "diffuse iteration" $OUT/c4.log | tail -2
cut -d, -f1-4 $OUT/c4_kernel_stats.csv | head -30
rm -rf $OUT/t
