#!/bin/bash
# ab.sh RAYS lib1 lib2 ... : step time (graph mode) per library variant in scratch/variants_live ("default" = shipped)
R=$GRAFT_REPO_ROOT
RAYS=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset TFRT_LIB_PATH; else export TFRT_LIB_PATH=$R/scratch/variants_live/lib_$v.so; fi
  echo -n "$v: "; python $R/scratch/prof_step.py $RAYS graph 60 2>&1 | grep -v amdgpu | tail -1
done
