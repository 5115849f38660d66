#!/bin/bash
# Round 3: the source-function sweep with the exact path mean (ftte_segment_source): parity of everything that carries emission,
# then configs[4] on one GPU with the pair form (default) and one wavefront per brick.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_source
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "emission or source or equilibrium or emit or pair or hybrid_with" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 300 python tools/bench_config5.py 256 6 > $OUT/pair.log 2>&1; grep "iteration" $OUT/pair.log | tail -3
timeout -k 10 300 python tools/bench_config5.py 256 6 --team=0 > $OUT/solo.log 2>&1; grep "iteration" $OUT/solo.log | tail -3
timeout -k 10 300 python tools/bench_config5.py 256 6 --team=2 --pair_waves=3 > $OUT/pair3.log 2>&1; grep "iteration" $OUT/pair3.log | tail -2
