#!/bin/bash
# Round 3: fine-level bricks for fully refined blocks: the hybrid and configs tests, then configs[3] and the 256^3 + 64^3 case with and without.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_fine
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hybrid_gpu.py tests/test_configs_gpu.py tests/test_parity_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 300 python tools/bench_config4.py 128 --no-reference > $OUT/c4_fine.log 2>&1; grep "diffuse iteration" $OUT/c4_fine.log | tail -2
timeout -k 10 300 python tools/bench_config4.py 128 --no-reference --fine_bricks 0 > $OUT/c4_forest.log 2>&1; grep "diffuse iteration" $OUT/c4_forest.log | tail -2
timeout -k 10 600 python tools/bench_config4.py 256 --no-reference > $OUT/c4_256_fine.log 2>&1; grep "diffuse iteration" $OUT/c4_256_fine.log | tail -2
timeout -k 10 600 python tools/bench_config4.py 256 --no-reference --fine_bricks 0 > $OUT/c4_256_forest.log 2>&1; grep "diffuse iteration" $OUT/c4_256_forest.log | tail -2
