#!/bin/bash
# BASELINE configs[2] (192 directions) and configs[4] (50 source iterations) on one GPU, for the record under profiles/
OUT=$GRAFT_REPO_ROOT/gpurun_out/configs
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py --ndir 192 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/config3_192dir_one_gpu.json 2> $OUT/c3.err
python3 -c "
import json; r=json.load(open('$OUT/config3_192dir_one_gpu.json')); print('192 directions: ms/step %.2f value %.3e sweep %.2f'%(r['ms_per_step'], r['value'], r['roofline']['avg_launch_ms']))"
timeout -k 10 600 python3 tools/bench_config5.py 256 50 > $OUT/config5_50_iterations.log 2>&1
grep -v "amdgpu.ids" $OUT/config5_50_iterations.log | tail -8
