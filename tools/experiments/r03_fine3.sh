#!/bin/bash
# Round 3: after the forest builder's changes: the randomised hybrid cross-check, the whole GPU suite, counter passes of configs[3] with fine bricks.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_fine3
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tests/fuzz_hybrid_gpu.py 120 7 > $OUT/fuzz_hybrid.txt 2>&1 || { tail -5 $OUT/fuzz_hybrid.txt; exit 1; }
tail -1 $OUT/fuzz_hybrid.txt
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
python tools/pmc_passes.py --out profiles/r03_pmc_config4_fine_bricks.json --note "configs[3] with the refined block swept by bricks of its own: 128^3 + refined 32^3 block, 8 groups, 96 directions, 4 iterations (the first builds the plan) + one star" -- tools/bench_config4.py 128 > $OUT/c4.txt 2>&1 || { tail $OUT/c4.txt; exit 1; }
cat $OUT/c4.txt
timeout -k 10 300 python tools/bench_config4.py 128 > $OUT/config4_unprofiled.log 2>&1; grep "diffuse iteration\|tracer" $OUT/config4_unprofiled.log | tail -4
timeout -k 10 600 python tools/bench_config4.py 256 --no-reference > $OUT/config4_256_unprofiled.log 2>&1; grep "leaves\|diffuse iteration" $OUT/config4_256_unprofiled.log | tail -4
