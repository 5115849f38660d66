#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hybrid_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
for opts in "" "--chunk 16" "--chunk 4" "--group 4" "--group 2" "--chunk 16 --group 4"; do
echo "== $opts"; python tools/bench_config4.py 128 --no-point $opts 2>&1 | grep "diffuse iteration 3"
done
