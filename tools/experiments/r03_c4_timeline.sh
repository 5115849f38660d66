#!/bin/bash
# Round 3: the timeline of one iteration of configs[3] (kernel trace, per stream): who waits for whom
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_c4_timeline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o c4 -- python3 $GRAFT_REPO_ROOT/tools/bench_config4.py 128 --no-reference > $OUT/c4.log 2>&1
find $OUT/t -name "*kernel_trace.csv" -exec cp {} $OUT/c4_kernel_trace.csv \;
rm -rf $OUT/t
grep "diffuse iteration" $OUT/c4.log | tail -1
python3 - <<P
import csv
rows=list(csv.DictReader(open("$OUT/c4_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last iteration: from the last base_cells_kernel to the last merge kernel
names=[r["Kernel_Name"] for r in rows]
last_merge=max(i for i,n in enumerate(names) if "merge_kernel" in n)
first=max(i for i,n in enumerate(names[:last_merge]) if "base_cells_kernel" in n)
first=min(i for i in range(first-3, first+1) if "base_cells" in names[i] or "to_layout" in names[i] or True)
it=rows[first:last_merge+1]
t0=int(it[0]["Start_Timestamp"])
def short(n):
    for k in ("brick_kernel<3, 0, 0, true>","brick_kernel","amr_level","amr_combine","amr_export","amr_fine_import","merge","to_layout","base_cells","cell_major","fillBuffer","copyBuffer"):
        if k in n: return k
    return n[:30]
streams={}
for r in it: streams.setdefault(r["Queue_Id"],[]).append(r)
print("iteration window %.2f ms, %d dispatches, %d streams"%((int(it[-1]["End_Timestamp"])-t0)/1e6,len(it),len(streams)))
for q,rs in streams.items():
    print("stream",q)
    run=None
    for r in rs:
        k=short(r["Kernel_Name"]); s=(int(r["Start_Timestamp"])-t0)/1e6; e=(int(r["End_Timestamp"])-t0)/1e6
        if run and run[0]==k and s-run[2]<0.05: run[2]=e; run[3]+=1; run[4]+=e-s
        else:
            if run: print("   %-28s x%4d  %7.3f .. %7.3f ms  busy %6.3f"%(run[0],run[3],run[1],run[2],run[4]))
            run=[k,s,e,1,e-s]
    if run: print("   %-28s x%4d  %7.3f .. %7.3f ms  busy %6.3f"%(run[0],run[3],run[1],run[2],run[4]))
P
