#!/bin/bash
# Round 3: where the persistent form's time goes: per-queue drain times and polls (FTTE_QUEUE_STATS), three ways of cutting the queues.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_persistent2
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export FTTE_QUEUE_STATS=1
for nnu in 8 1; do
  for mix in 0 1 2; do
    timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --nnu $nnu --dataflow 3 --opt queue_mix=$mix > $OUT/b_${nnu}_$mix.json 2> $OUT/b_${nnu}_$mix.err || { echo failed; tail -5 $OUT/b_${nnu}_$mix.err; exit 1; }
    echo "== nnu $nnu queue_mix $mix"; tail -8 $OUT/b_${nnu}_$mix.err
    python -c "
import json; d=json.load(open('$OUT/b_${nnu}_$mix.json')); print('step %.2f ms, sweep phase %.2f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  done
done
