#!/bin/bash
# Round-2 evidence: rocprofv3 kernel-trace stats and separate PMC passes (never combined with trace domains) of bench.py and the
# side benches; raw output under gpurun_out/prof_<tag>/, summaries copied into profiles/ by tools/r02_profiles_summary.py.
# usage (through gpurun): bash tools/r02_profiles.sh <tag>
TAG=${1:-r02a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
python3 $B --steps 5 --warmup 2 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $B --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_l1 -o bench -- python3 $B --steps 5 --warmup 2 --no-cpu-baseline --lanes 1 > $OUT/bench_l1.json 2> $OUT/bench_l1.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- python3 $B --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $OUT/$C.json 2> $OUT/$C.err
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex brick_kernel --output-format csv -d $OUT/SQ -o pmc -- python3 $B --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $OUT/SQ.json 2> $OUT/SQ.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM --kernel-include-regex brick_kernel --output-format csv -d $OUT/SQ2 -o pmc -- python3 $B --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $OUT/SQ2.json 2> $OUT/SQ2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/config4 -o config4 -- python3 $GRAFT_REPO_ROOT/tools/bench_config4.py 128 > $OUT/config4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/config5 -o config5 -- python3 $GRAFT_REPO_ROOT/tools/bench_config5.py 256 12 > $OUT/config5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/loop -o loop -- python3 $GRAFT_REPO_ROOT/tools/bench_loop.py 256 > $OUT/loop.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/point -o point -- python3 $GRAFT_REPO_ROOT/tools/bench_point.py 256 512 > $OUT/point.log 2>&1
find $OUT -name "*kernel_stats.csv" | head; cat $OUT/bench_plain.json | cut -c1-400
