#!/bin/bash
# Round 3: counters of the one-group rank shape (what a rank of an 8-GPU run executes): bytes per update, vector-unit busy
# fraction, resident wavefronts -- stage form (pair kernel) and persistent form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_nnu1_pmc
timeout -k 10 500 python tools/pmc_passes.py --out gpurun_out/r03_nnu1_pmc/stages.json --kernels "brick" -- bench.py --nnu 1 --steps 2 --warmup 1 --no-cpu-baseline --lanes 1 > gpurun_out/r03_nnu1_pmc/stages.txt 2>&1 || { tail -20 gpurun_out/r03_nnu1_pmc/stages.txt; exit 1; }
timeout -k 10 500 python tools/pmc_passes.py --out gpurun_out/r03_nnu1_pmc/persistent.json --kernels "brick" -- bench.py --nnu 1 --steps 2 --warmup 1 --no-cpu-baseline --dataflow 3 --share 0 > gpurun_out/r03_nnu1_pmc/persistent.txt 2>&1 || { tail -20 gpurun_out/r03_nnu1_pmc/persistent.txt; exit 1; }
tail -30 gpurun_out/r03_nnu1_pmc/stages.txt; tail -30 gpurun_out/r03_nnu1_pmc/persistent.txt
