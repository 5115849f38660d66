#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests/test_hybrid_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 600 python3 tools/pcie_rate.py 2>&1 | grep "config-4"
