#!/bin/bash
# hybrid sweep as 1..4 pipelines on streams of their own
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02v
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for h in 2 3 4; do
  timeout -k 10 200 python3 tools/bench_config4.py 128 --no-reference --no-point --halves $h > $OUT/halves$h.log 2>&1
  grep "diffuse iteration [23]" $OUT/halves$h.log | sed "s/^/halves $h: /"
done
for h in 3 4; do
  timeout -k 10 200 python3 tools/bench_config4.py 128 --no-reference --no-point --halves $h --group 2 > $OUT/halves${h}_g2.log 2>&1
  grep "diffuse iteration [23]" $OUT/halves${h}_g2.log | sed "s/^/halves $h group 2: /"
done
timeout -k 10 600 python -m pytest tests/test_hybrid_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "hybrid or config3" 2>&1 | tail -3
