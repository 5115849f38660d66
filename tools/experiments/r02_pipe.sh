#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pipe
timeout -k 10 900 python -m pytest tests/test_host_boundary_gpu.py tests/test_fortran_host_gpu.py tests/test_abi_and_host.py -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 600 python3 tools/pcie_rate.py 2>&1 | tee gpurun_out/pipe/pcie.log | grep -v Gloo
