#!/bin/bash
# Instruction counts and busy cycles of one kernel of the bench step.
# bash scratch/collect_kernel_pmc.sh KERNEL_SUBSTRING LAUNCHES_PER_STEP   (RAYS=..., default 1,000,000)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kpmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kp0 /tmp/kp1 /tmp/kp2
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_WR -d /tmp/kp0 --output-format csv -- python $R/scratch/prof_step.py ${RAYS:-1000000} fused 12 > $O/o0.txt 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES -d /tmp/kp1 --output-format csv -- python $R/scratch/prof_step.py ${RAYS:-1000000} fused 12 > $O/o1.txt 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS -d /tmp/kp2 --output-format csv -- python $R/scratch/prof_step.py ${RAYS:-1000000} fused 12 > $O/o2.txt 2>&1
python $R/scratch/pmc_to_json.py $O/$1.json $1 $2 /tmp/kp0 /tmp/kp1 /tmp/kp2 > /dev/null
python - <<PY
import json, csv, glob, statistics
d=json.load(open("$O/$1.json"))
for p in d["passes"]:
    w=p["SQ_WAVES"]
    print("$1 launch",p["pass"],"waves %.0f"%w, {k:round(v/w,1) for k,v in p.items() if k not in ("pass","SQ_WAVES")})
# durations of the kernel under the counter runs (kernel trace of the first run)
for f in glob.glob("/tmp/kp0/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "$1" in r["Kernel_Name"]]
    du=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
    du=du[len(du)%$2:]; du=du[(len(du)//$2//3)*$2:]
    print("durations under --pmc (us), by pass:", [round(statistics.median(du[k::$2]),1) for k in range($2)])
PY
