#!/bin/bash
# the default bench line several times on one box (box-to-box and run-to-run spread)
OUT=$GRAFT_REPO_ROOT/gpurun_out/repeat
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for k in 1 2 3 4; do
  timeout -k 10 150 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/r$k.json 2> $OUT/r$k.err
  python3 -c "
import json; r=json.load(open('$OUT/r$k.json')); print('run $k ms/step %.2f sweep %.2f'%(r['ms_per_step'], r['roofline']['avg_launch_ms']))"
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
