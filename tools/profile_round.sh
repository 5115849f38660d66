#!/bin/bash
# rocprofv3 --kernel-trace --stats of the round's commands, CSV summaries into gpurun_out/prof_<tag>/ (run through gpurun).
# usage: tools/profile_round.sh <tag>
set -e
TAG=${1:-r01c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/config4 -o config4 -- python3 $GRAFT_REPO_ROOT/tools/bench_config4.py 128 --no-reference > $OUT/config4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/loop -o loop -- python3 $GRAFT_REPO_ROOT/tools/bench_loop.py 256 > $OUT/loop.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/point -o point -- python3 $GRAFT_REPO_ROOT/tools/bench_point.py 256 512 > $OUT/point.log 2>&1
find $OUT -name "*kernel_stats.csv" | head
