#!/bin/bash
# Round 3: the host side of the hybrid plan after the forest builder learnt to keep its arrays from direction to direction and to
# leave the interior of a fine block alone: tests, then the first call (plan + forests) and the iteration on configs[3] and 256^3 + 64^3
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_plan_time
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hybrid_gpu.py tests/test_configs_gpu.py tests/test_fortran_host_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 300 python tools/bench_config4.py 128 --no-reference > $OUT/c4.log 2>&1; grep "diffuse iteration" $OUT/c4.log | sed -n "1p;\$p"
timeout -k 10 600 python tools/bench_config4.py 256 --no-reference > $OUT/c4_256.log 2>&1; grep "diffuse iteration" $OUT/c4_256.log | sed -n "1p;\$p"
timeout -k 10 600 python tools/bench_config4.py 256 --no-reference --fine_bricks 0 > $OUT/c4_256_forest.log 2>&1; grep "diffuse iteration" $OUT/c4_256_forest.log | sed -n "1p;\$p"
