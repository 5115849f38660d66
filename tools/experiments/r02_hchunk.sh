#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/hchunk
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for c in 4 2 3 6 8; do
  timeout -k 10 200 python3 tools/bench_config4.py 128 --no-reference --no-point --chunk $c > $OUT/c$c.log 2>&1
  grep "diffuse iteration [23]" $OUT/c$c.log | sed "s/^/chunk $c: /"
done
for g in 2 4; do
  timeout -k 10 200 python3 tools/bench_config4.py 128 --no-reference --no-point --group $g > $OUT/g$g.log 2>&1
  grep "diffuse iteration [23]" $OUT/g$g.log | sed "s/^/group $g: /"
done
