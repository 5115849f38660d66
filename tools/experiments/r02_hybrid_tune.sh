#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/htune
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for p in 2 3 4; do for c in 4 8; do
  timeout -k 10 200 python3 tools/bench_config4.py 128 --no-reference --no-point --pipelines $p --chunk $c > $OUT/p${p}_c$c.log 2>&1
  grep "diffuse iteration 3" $OUT/p${p}_c$c.log | sed "s/^/pipelines $p chunk $c: /"
done; done
