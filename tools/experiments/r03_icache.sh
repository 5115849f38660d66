#!/bin/bash
# Round 3: does the brick kernel's code fit the instruction cache?  SQC_ICACHE counters (a pass of their own) of the sweep at eight
# groups, at one group (pair kernel and one wavefront per brick) -- hits, misses, and what the waves wait for
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
    tag=$1; shift
    rm -rf $OUT/$tag
    timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-include-regex "brick" --output-format csv -d $OUT/$tag -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lanes 1 "$@" > $OUT/$tag.log 2>&1 || { echo "$tag failed"; tail -5 $OUT/$tag.log; return; }
    python3 - <<P
import csv, glob, collections
tot=collections.defaultdict(float); n=0
for f in glob.glob("$OUT/$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Counter_Name"]]+=float(r["Counter_Value"])
req,hit,miss,dup=tot["SQC_ICACHE_REQ"],tot["SQC_ICACHE_HITS"],tot["SQC_ICACHE_MISSES"],tot["SQC_ICACHE_MISSES_DUPLICATE"]
print("%-28s icache requests %.3e  hits %.4f  misses %.4f (+ duplicates %.4f)  ifetch %.3e  wave cycles %.3e  waiting for instructions/any %.3f" % ("$tag", req, hit/req, miss/req, dup/req, tot["SQ_IFETCH"], tot["SQ_WAVE_CYCLES"], tot["SQ_WAIT_INST_ANY"]/max(tot["SQ_WAVE_CYCLES"],1)))
P
    rm -rf $OUT/$tag
}
run nnu8
run nnu1_pair --nnu 1
run nnu1_solo --nnu 1 --team 0
run nnu1_solo_ablated --nnu 1 --team 0 --opt ablate=63
