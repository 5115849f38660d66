#!/bin/bash
# Round 3: thin forest levels in one launch (option forest_fuse): parity, then configs[3] and 256^3 + 64^3 with and without
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_fuse
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hybrid_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for fuse in 0 4096 16384 65536; do
  echo "forest_fuse $fuse"
  timeout -k 10 300 python tools/bench_config4.py 128 --no-reference --forest_fuse $fuse > $OUT/c4_$fuse.log 2>&1; grep "diffuse iteration" $OUT/c4_$fuse.log | tail -2
done
for fuse in 0 16384 65536; do
  echo "256: forest_fuse $fuse"
  timeout -k 10 600 python tools/bench_config4.py 256 --no-reference --forest_fuse $fuse > $OUT/c4_256_$fuse.log 2>&1; grep "diffuse iteration" $OUT/c4_256_$fuse.log | tail -1
done
