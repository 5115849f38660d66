#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by a per cent or two): libftte.so against libftte_variant.so, alternating
cd $GRAFT_REPO_ROOT
P=radiativetransfer_amd
cp $P/libftte.so /tmp/libA.so; cp $P/libftte_variant.so /tmp/libB.so
run() { cp /tmp/lib$1.so $P/libftte.so; python bench.py --steps 8 --warmup 3 --no-cpu-baseline "${@:2}" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1: step %.2f ms, sweep phase %.2f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
for i in 1 2 3; do run A "$@"; run B "$@"; done
cp /tmp/libA.so $P/libftte.so
