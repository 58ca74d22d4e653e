#!/bin/bash
# kernels of ONE replay of the captured optimiser step, in order, with durations and gaps.  graph_kernels.sh RAYS
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/gk
rocprofv3 --kernel-trace --output-format csv -d /tmp/gk -- python $R/scratch/prof_step.py $1 graph 40 > /tmp/gk.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/gk/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# the last replay: find the last k_sgd_process_multi and walk back to the previous one
idx=[i for i,r in enumerate(rows) if "k_sgd_process" in r["Kernel_Name"]]
a,b=idx[-2]+1,idx[-1]+1
prev=None; tot=0
for r in rows[a:b]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev)/1e3 if prev else 0.0
    print(f"{r['Kernel_Name'][:64]:64s} {(e-s)/1e3:7.1f} us  gap {gap:5.1f}")
    prev=e
print("span us:", (int(rows[b-1]["End_Timestamp"])-int(rows[a]["Start_Timestamp"]))/1e3, "kernels", b-a)
PY
tail -1 /tmp/gk.log
