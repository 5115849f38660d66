#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/emit
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "emis or source or iteration or config4" 2>&1 | tail -3 || exit 1
timeout -k 10 300 python3 tools/bench_config5.py 256 6 2>&1 | tee gpurun_out/emit/config5.log | grep "iteration"
