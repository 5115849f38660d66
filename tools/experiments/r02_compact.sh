#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/compact
timeout -k 10 900 python -m pytest tests/test_hybrid_gpu.py tests/test_configs_gpu.py tests/test_parity_gpu.py tests/test_fortran_host_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 300 python3 tools/bench_config4.py 128 --no-reference --no-point 2>&1 | tee gpurun_out/compact/config4.log | grep "diffuse iteration"
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
