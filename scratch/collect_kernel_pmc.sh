#!/bin/bash
# Instruction counts of one kernel of the bench step.  bash scratch/collect_kernel_pmc.sh KERNEL_SUBSTRING LAUNCHES_PER_STEP
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kpmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kp0 /tmp/kp1
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_WR -d /tmp/kp0 --output-format csv -- python $R/scratch/prof_step.py ${RAYS:-1000000} fused 12 > $O/o0.txt 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES -d /tmp/kp1 --output-format csv -- python $R/scratch/prof_step.py ${RAYS:-1000000} fused 12 > $O/o1.txt 2>&1
python $R/scratch/pmc_to_json.py $O/$1.json $1 $2 /tmp/kp0 /tmp/kp1 > /dev/null
python - <<PY
import json
d=json.load(open("$O/$1.json"))
for p in d["passes"]:
    w=p["SQ_WAVES"]
    print("$1 launch",p["pass"],"waves %.0f"%w, {k:round(v/w,1) for k,v in p.items() if k not in ("pass","SQ_WAVES")})
PY
