#!/bin/bash
# Round 3: what the latency of one brick is made of -- the accesses of the layer loop switched off one after the other (option ablate)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_latency
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
for cfg in "0 16 1 0" "0 16 1 1" "0 16 1 3" "0 16 1 7" "0 16 1 23" "0 16 1 55" "0 16 1 63" "0 16 2 0" "0 16 2 63"; do
    set -- $cfg
    D=$OUT/f$1_c$2_d$3_a$4
    rm -rf $D
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/experiments/r03_latency.py run $1 $2 $3 $4 > $D.log 2>&1 || { echo "$cfg failed"; tail -5 $D.log; exit 1; }
    echo "== form $1 chunk $2 directions $3 ablate $4"; grep directions $D.log
    python3 $GRAFT_REPO_ROOT/tools/experiments/r03_latency.py read $D | head -3
    rm -rf $D
done
