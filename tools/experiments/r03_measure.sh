#!/bin/bash
# Round 3: counter passes of the headline configuration (bench.py --measure-traffic writes profiles/pmc_traffic.json), the kernel trace
# of the same command with one stream, and the bench line itself.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_measure
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py --measure-traffic > $OUT/measure.log 2>&1 || { tail -20 $OUT/measure.log; exit 1; }
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_l1 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --lanes 1 > $OUT/bench_l1.json 2> $OUT/bench_l1.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/bench_traced.err
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json | cut -c1-1500
find $OUT -name "*kernel_stats.csv"
