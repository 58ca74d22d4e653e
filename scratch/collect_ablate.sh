#!/bin/bash
# VALU / SALU / LDS instruction counts of k_intersect_group for the stage-ablation builds
# (scratch/variants/lib_<name>.so).  Run on the GPU box: bash scratch/collect_ablate.sh name...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ablate; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/ab_$v
  if [ "$v" != "default" ]; then export TFRT_LIB_PATH=$R/scratch/variants/lib_$v.so; else unset TFRT_LIB_PATH; fi
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d /tmp/ab_$v --output-format csv -- python $R/scratch/prof_step.py 1000000 fused 12 > $O/o_$v.txt 2>&1
  python $R/scratch/pmc_to_json.py $O/$v.json k_intersect_group 3 /tmp/ab_$v > /dev/null
  python - <<PY
import json
d=json.load(open("$O/$v.json"))
for p in d["passes"]:
    w=p["SQ_WAVES"]
    print("$v pass",p["pass"],"per wave: VALU %.0f SALU %.0f LDS %.0f VMEM %.0f  wave_cycles %.0f wait %.0f%%"%(p["SQ_INSTS_VALU"]/w,p["SQ_INSTS_SALU"]/w,p["SQ_INSTS_LDS"]/w,p["SQ_INSTS_VMEM_RD"]/w,4*p["SQ_WAVE_CYCLES"]/w,100*p["SQ_WAIT_ANY"]/p["SQ_WAVE_CYCLES"]))
PY
done
