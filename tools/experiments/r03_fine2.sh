#!/bin/bash
# Round 3: fine-level bricks: layers per fine brick, chunk of the base bricks, pipelines.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_fine2
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for n in 128 256; do
for args in "--fine_chunk 8" "--fine_chunk 16" "--fine_chunk 32" "--chunk 8 --fine_chunk 8" "--chunk 8 --fine_chunk 16" "--pipelines 2 --fine_chunk 8" "--pipelines 4 --fine_chunk 8"; do
  timeout -k 10 600 python tools/bench_config4.py $n --no-reference $args > $OUT/l.log 2>&1; echo "n $n $args: $(grep 'diffuse iteration 3' $OUT/l.log | cut -c1-60)"
done; done
