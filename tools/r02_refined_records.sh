#!/bin/bash
# refined cell arrays, for the record under profiles/: configs[3], the same one size up, scattered clusters, a source iteration
OUT=$GRAFT_REPO_ROOT/gpurun_out/refined
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
echo "## tools/bench_config4.py 128 --no-reference  (BASELINE configs[3])"
timeout -k 10 300 python3 tools/bench_config4.py 128 --no-reference 2>&1 | grep -v amdgpu
echo "## tools/bench_config4.py 256 --no-reference --no-point  (256^3 base, central 64^3 block refined once)"
timeout -k 10 600 python3 tools/bench_config4.py 256 --no-reference --no-point 2>&1 | grep -v amdgpu
for k in 1 4 8 16 32; do
  echo "## tools/bench_clusters.py 128 $k"
  timeout -k 10 300 python3 tools/bench_clusters.py 128 $k 2>&1 | grep -v amdgpu
done
echo "## tools/experiments/r02_emit_hybrid.py  (source iteration on the configs[3] tree)"
timeout -k 10 300 python3 tools/experiments/r02_emit_hybrid.py 2>&1 | grep -v amdgpu
} > $OUT/refined.log 2>&1
tail -5 $OUT/refined.log
