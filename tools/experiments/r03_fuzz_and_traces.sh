#!/bin/bash
# Round 3: the randomised cross-checks on the round's code (now with the persistent form and slot lists in the draw), the rehearsal of
# bench.py's multi-rank branch, and kernel traces of the side benches.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_fuzz
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_distributed_gloo.py -x -q -m gpu > $OUT/dist.log 2>&1 || { tail -30 $OUT/dist.log; exit 1; }
tail -2 $OUT/dist.log
timeout -k 10 500 python tests/fuzz_gpu.py 150 3 > $OUT/fuzz_uniform.txt 2>&1 || { tail -5 $OUT/fuzz_uniform.txt; exit 1; }
tail -1 $OUT/fuzz_uniform.txt
timeout -k 10 500 python tests/fuzz_hybrid_gpu.py 150 3 > $OUT/fuzz_hybrid.txt 2>&1 || { tail -5 $OUT/fuzz_hybrid.txt; exit 1; }
tail -1 $OUT/fuzz_hybrid.txt
cd /tmp && export TMPDIR=/tmp
for t in "config4 tools/bench_config4.py 128" "config5 tools/bench_config5.py 256 6" "loop tools/bench_loop.py 256" "point tools/bench_point.py 256 512"; do
  set -- $t; name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python3 $GRAFT_REPO_ROOT/$1 ${@:2} > $OUT/$name.log 2>&1
  find $OUT/$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
  grep -v "rocprofv3\|amdgpu.ids\|^W20\|^E20\|^I20" $OUT/$name.log | tail -4
done
