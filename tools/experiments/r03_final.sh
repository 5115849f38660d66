#!/bin/bash
# Round 3: what the driver runs at the end: the GPU suite, smoke, the default bench line.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_final
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail $OUT/smoke.log; exit 1; }
tail -2 $OUT/smoke.log
timeout -k 10 900 python bench.py --measure-traffic > $OUT/measure.log 2>&1 || { tail -20 $OUT/measure.log; exit 1; }
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json | cut -c1-2600
