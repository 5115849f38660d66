#!/bin/bash
# Where the brick kernel's wavefronts wait: scalar-cache and instruction-cache hit rates, and the time instructions of each kind
# are in flight (separate PMC passes, no trace domains).  usage (through gpurun): bash tools/r02_stall_counters.sh <tag>
TAG=${1:-r02f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/stall_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --lanes 1 --team 0"
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --kernel-include-regex brick_kernel --output-format csv -d $OUT/SQC -o pmc -- python3 $B $ARGS > $OUT/SQC.json 2> $OUT/SQC.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SMEM SQ_IFETCH --kernel-include-regex brick_kernel --output-format csv -d $OUT/SQ3 -o pmc -- python3 $B $ARGS > $OUT/SQ3.json 2> $OUT/SQ3.err
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex brick_kernel --output-format csv -d $OUT/SQ4 -o pmc -- python3 $B $ARGS > $OUT/SQ4.json 2> $OUT/SQ4.err
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "brick_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot):
    print(f"{k:28s} {tot[k]:.6g}")
PY
