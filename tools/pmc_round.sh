#!/bin/bash
# Separate rocprofv3 --pmc passes (no trace domains besides the kernel trace) for the sweep kernel of bench.py.
# usage (through gpurun): bash tools/pmc_round.sh <tag>
set -e
TAG=${1:-r01c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$C.json 2> $OUT/$C.err
done
find $OUT -name "*counter_collection.csv" | head
