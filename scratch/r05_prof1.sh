#!/bin/bash
# round 5: A/B of in-place kernel variants + kernel stats of the graph step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
bash $R/scratch/ab.sh 1000000 default "$@" > $O/ab1.txt 2>&1
cat $O/ab1.txt
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/scratch/prof_step.py 1000000 graph 200 > $O/prof_step.txt 2>&1
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/step_kernel_stats.csv
head -20 $O/step_kernel_stats.csv | cut -c1-200
