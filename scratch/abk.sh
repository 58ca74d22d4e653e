#!/bin/bash
# abk.sh RAYS lib1 lib2 ...: average duration of k_intersect_beam in single-pass traces, per library variant
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
RAYS=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset TFRT_LIB_PATH; else export TFRT_LIB_PATH=$R/scratch/variants_live/lib_$v.so; fi
  rm -rf /tmp/abk_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$v -- python $R/scratch/pass1_time.py $RAYS > /tmp/abk_$v.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/abk_$v/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "k_intersect_beam" in r["Name"]:
        print("$v N=$RAYS beam: calls", r["Calls"], "avg_us %.1f min %.1f max %.1f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
