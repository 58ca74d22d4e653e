#!/bin/bash
# round 5: PMC counters of the in-place step's kernels (one counter set per run, kernel-trace only)
# usage: r05_pmc.sh OUTDIR_NAME  -> gpurun_out/r05/<name>_pmc_*.json
R=$GRAFT_REPO_ROOT; TAG=${1:-a}; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
STEP="python $R/scratch/prof_step.py 1000000 fused 30"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc$i; rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc$i --output-format csv -- $STEP > /tmp/o$i.txt 2>&1
  tail -1 /tmp/o$i.txt
  i=$((i+1))
done
D="/tmp/pmc0 /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 /tmp/pmc4"
python $R/scratch/pmc_to_json.py $O/${TAG}_pmc_inplace.json k_trace_inplace 1 $D > /dev/null
python $R/scratch/pmc_to_json.py $O/${TAG}_pmc_backward.json k_backward_chain 1 $D > /dev/null
python - <<PY
import json
for k in ("inplace","backward"):
    d=json.load(open("$O/${TAG}_pmc_%s.json"%k))["passes"][0]
    w=d.get("SQ_WAVES",1)
    print(k, {c: round(v/w,1) for c,v in d.items() if c.startswith("SQ_INSTS") or c in ("SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_LDS","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_INST_CYCLES_VMEM")}, "waves", w, "fetchKB", d.get("FETCH_SIZE_KB"), "writeKB", d.get("WRITE_SIZE_KB"))
PY
