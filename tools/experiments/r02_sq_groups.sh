#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02o
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for G in 3 8; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex brick_kernel --output-format csv -d $OUT/sq_g$G -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 --brick-waves 2 --group $G > $OUT/sq_g$G.json 2> $OUT/sq_g$G.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --kernel-include-regex brick_kernel --output-format csv -d $OUT/sq2_g$G -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 --brick-waves 2 --group $G > $OUT/sq2_g$G.json 2> $OUT/sq2_g$G.err
done
python3 - <<'PY'
import csv, collections, glob, os
OUT=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02o"
upd=256**3*96*8
for G in (3,8):
    for sub in ("sq","sq2"):
        tot=collections.defaultdict(float)
        for f in glob.glob(f"{OUT}/{sub}_g{G}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                tot[r["Counter_Name"]]+=float(r["Counter_Value"])
        print("group",G,sub,{k:"%.3g"%(v*64/upd) for k,v in sorted(tot.items())})
        if "GRBM_GUI_ACTIVE" in tot:
            cyc=tot["GRBM_GUI_ACTIVE"]/8
            print("   cycles %.3g  VALU busy %.2f  waves/SIMD %.2f  wait_any/wave %.2f wait_inst/wave %.2f active/wave %.2f"%(cyc, tot["SQ_ACTIVE_INST_VALU"]*4/1024/cyc, tot["SQ_WAVE_CYCLES"]*4/1024/cyc, tot["SQ_WAIT_ANY"]/tot["SQ_WAVE_CYCLES"], tot["SQ_WAIT_INST_ANY"]/tot["SQ_WAVE_CYCLES"], tot["SQ_ACTIVE_INST_ANY"]/tot["SQ_WAVE_CYCLES"]))
PY
