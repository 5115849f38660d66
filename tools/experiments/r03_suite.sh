#!/bin/bash
# Round 3: the whole GPU suite, smoke, the counter passes of the headline configuration (bench.py --measure-traffic) and the bench line.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_suite
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail $OUT/smoke.log; exit 1; }
tail -2 $OUT/smoke.log
