#!/bin/bash
# Per-kernel averages of N eager fused steps: bash scratch/kstats_step.sh [RAYS] [MODE] [extra prof_step args]
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/scratch/prof_step.py ${1:-1000000} ${2:-fused} 40 $3 > /tmp/o.txt 2>&1
tail -1 /tmp/o.txt
python - <<PY
import csv,glob
f=glob.glob("/tmp/ks/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:${4:-16}]:
    print("%-62s %5s %9.1f us %6s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
