#!/bin/bash
# Round 3: ionisation equilibrium with its three statistics combined per workgroup before the atomics: parity, then the closed loop
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_chem
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_chem_gpu.py tests/test_fortran_host_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
timeout -k 10 300 python tools/bench_loop.py 256 2>&1 | tail -3
