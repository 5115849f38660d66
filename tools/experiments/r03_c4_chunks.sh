#!/bin/bash
# Round 3: configs[3] is a chain of stage latencies (FTTE_HYBRID_TIMELINE): layers per base brick x layers per fine brick x directions per group
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_c4_chunks
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for args in "" "--chunk 8" "--chunk 4" "--chunk 8 --fine_chunk 4" "--chunk 8 --fine_chunk 8" "--chunk 8 --fine_chunk 16" "--chunk 4 --fine_chunk 4" "--chunk 16 --fine_chunk 4" "--chunk 8 --fine_chunk 8 --group 2" "--chunk 8 --fine_chunk 8 --group 4"; do
  timeout -k 10 300 python tools/bench_config4.py 128 --no-reference --no-point $args > $OUT/l.log 2>&1; echo "$args: $(grep 'diffuse iteration 3' $OUT/l.log | cut -c1-80)"
done
