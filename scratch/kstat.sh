#!/bin/bash
# kstat.sh OUTNAME RAYS MODE STEPS  -> gpurun_out/OUTNAME.csv (rocprofv3 kernel stats of scratch/prof_step.py)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -- python $R/scratch/prof_step.py $2 $3 $4 > $R/gpurun_out/$1.log 2>&1
cp $(ls /tmp/prof_$1/*/*kernel_stats.csv | head -1) $R/gpurun_out/$1.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/$1.csv")))
print("$1")
for r in rows[:9]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} pct {r['Percentage']}")
PY
tail -1 $R/gpurun_out/$1.log
