#!/bin/bash
# Round 3: the latency of one brick (see r03_latency.py), solo and pair form, chunk 4 and 16, one and three directions
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_latency
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
for cfg in "0 4 3" "2 4 3" "0 4 1" "2 4 1" "0 16 3" "2 16 3"; do
    set -- $cfg
    D=$OUT/f$1_c$2_d$3
    rm -rf $D
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/experiments/r03_latency.py run $1 $2 $3 > $D.log 2>&1 || { echo "$cfg failed"; tail -5 $D.log; exit 1; }
    echo "== form $1 chunk $2 directions $3"; grep directions $D.log
    python3 $GRAFT_REPO_ROOT/tools/experiments/r03_latency.py read $D | head -14
    rm -rf $D
done
