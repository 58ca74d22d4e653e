#!/bin/bash
# kernel stats of a python command: bash scratch/kstats.sh OUTNAME python-args...   (run on the GPU box)
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O; NAME=$1; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$NAME
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$NAME -- python "$@" > $O/$NAME.log 2>&1
cp $(ls /tmp/ks_$NAME/*/*kernel_stats.csv | head -1) $O/$NAME.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/$NAME.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
tail -2 $O/$NAME.log
