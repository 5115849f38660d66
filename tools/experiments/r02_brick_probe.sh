#!/bin/bash
# brick engine: per-stage times and an SQ pass
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $OUT/stages.txt 2>&1
import numpy as np, torch, sys
sys.path.insert(0, ".")
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic
import bench
n, nnu = 256, 8
phi, theta, w = bench.directions(96, 96, 0)
kappa_host, uvb, box = synthetic.uniform_workload(n, nnu, seed=12345, tau_median=0.1)
dev = torch.device("cuda", 0)
kappa = torch.from_numpy(kappa_host).to(dev)
J = torch.empty((nnu, n ** 3), dtype=torch.float64, device=dev)
for group in (2, 4):
    eng = rt.DiffuseTransfer(device=0)
    eng.set_uniform_grid(n, box)
    eng.set_option("engine", 2); eng.set_option("group", group)
    for it in range(3):
        eng.set_opacity_device(nnu, kappa.data_ptr())
        eng.transport_device(phi, theta, w, uvb, J.data_ptr(), 0)
        torch.cuda.synchronize()
    rec = eng.launch_records()
    print("group", group, "stages", len(rec), "sum ms %.2f" % sum(m for m, _ in rec))
    for s, (ms, upd) in enumerate(rec):
        print("  stage %2d  %.3f ms  %5.1f Mupd  %.1f Gupd/s" % (s, ms, upd / 1e6, upd / ms / 1e6))
    eng.close()
PY
cd /tmp && export TMPDIR=/tmp
for G in 2 4; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex brick_kernel --output-format csv -d $OUT/sq_g$G -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --engine 2 --group $G > $OUT/sq_g$G.json 2> $OUT/sq_g$G.err
done
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-include-regex brick_kernel --output-format csv -d $OUT/sq2_g2 -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --engine 2 --group 2 > $OUT/sq2_g2.json 2> $OUT/sq2_g2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --engine 2 --group 2 > $OUT/kt.json 2> $OUT/kt.err
cat $OUT/stages.txt | head -60
