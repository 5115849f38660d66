#!/bin/bash
# the hybrid sweep's launches replayed from a captured hipGraph against issued one by one
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hybrid_gpu.py tests/test_configs_gpu.py tests/test_host_boundary_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for g in 1 0; do
  timeout -k 10 300 python3 tools/bench_config4.py 128 --no-reference --no-point --graph $g 2>&1 | grep "diffuse iteration [13]" | sed "s/^/graph $g: /"
done
for k in 4 8 16; do
  timeout -k 10 300 python3 tools/bench_clusters.py 128 $k 2>&1 | grep "hybrid 1 iteration 3" | sed "s/^/clusters $k graph 1: /"
  timeout -k 10 300 python3 tools/bench_clusters.py 128 $k --no-graph 2>&1 | grep "hybrid 1 iteration 3" | sed "s/^/clusters $k graph 0: /"
done
