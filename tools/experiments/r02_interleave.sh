#!/bin/bash
# A/B of the brick kernel's row interleave (FTTE_BRICK_INTERLEAVE) and register cap, rebuilding only ftte_brick.o on the box
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT/radiativetransfer_amd/csrc
run() { tag=$1; shift; (cd $GRAFT_REPO_ROOT && python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err); python3 - <<PY
import json
try:
    r=json.load(open("$OUT/$tag.json"))
    print("$tag", "ms/step %.2f"%r["ms_per_step"], "frac %.3f"%r["roofline"]["frac"], "sweep phase ms/step %.2f"%(r["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-300:])
PY
}
for I in 1 2 4 8; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -x hip -c ftte_brick.hip -o ftte_brick.o -DFTTE_BRICK_INTERLEAVE=$I 2> $OUT/build_$I.err
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libftte.so ftte_kernels.o ftte_brick.o ftte_api.o ftte_geometry.o ftte_amr.o ftte_point.o ftte_ingest.o
  run il${I}_w4 --brick-waves 4
  run il${I}_w3 --brick-waves 3
  run il${I}_w2 --brick-waves 2
done
