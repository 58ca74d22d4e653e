#!/bin/bash
# ms per optimiser step of the bench scene at several ray counts (the shards of a strong-scaling run),
# generic / fused eager / fused HIP-graph.  bash scratch/step_times.sh > profiles/rNN_step_times.txt
R=$GRAFT_REPO_ROOT
printf "%9s %9s %9s %9s   ms/step\n" rays generic fused graph
for n in 1000000 500000 250000 125000; do
  row=""
  for m in generic fused graph; do
    v=$(python $R/scratch/prof_step.py $n $m 60 2>/dev/null | grep "ms/step" | sed 's/.*: \([0-9.]*\) ms.*/\1/')
    row="$row $(printf '%9s' $v)"
  done
  printf "%9d%s\n" $n "$row"
done
