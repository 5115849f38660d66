#!/bin/bash
# round-2 first probe: baseline bench line, the shapes a rank would run under nu x direction sharding, and an SQ pass
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_base.json 2> $OUT/bench_base.err
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --nnu 1 --ndir 96 > $OUT/bench_nnu1.json 2> $OUT/bench_nnu1.err
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --nnu 8 --ndir 12 > $OUT/bench_ndir12.json 2> $OUT/bench_ndir12.err
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --nnu 4 --ndir 24 > $OUT/bench_nnu4.json 2> $OUT/bench_nnu4.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex sweep_kernel --output-format csv -d $OUT/sq -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq.json 2> $OUT/sq.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-include-regex sweep_kernel --output-format csv -d $OUT/sq2 -o sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq2.json 2> $OUT/sq2.err || echo "sq2 pass failed"
cat $OUT/bench_*.json
