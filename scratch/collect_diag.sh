#!/bin/bash
# Diagnostic PMC passes for k_intersect_group (memory pipeline, scalar cache, instruction mix).
# Run on the GPU box: bash scratch/collect_diag.sh   -> gpurun_out/diag/diag.json
# (small counter sets: a set the hardware cannot collect makes rocprofv3 abort and hang)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/diag; mkdir -p $O; rm -f $O/*.txt
cd /tmp; export TMPDIR=/tmp
STEP="python $R/scratch/prof_step.py 1000000 fused 12"
i=0; dirs=""
for set in \
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH" \
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_WAVE_CYCLES" \
 "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
 "TD_TD_BUSY_sum" \
 "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
 "TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum" \
 "SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_TC_STALL SQ_IFETCH" \
 "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" ; do
  rm -rf /tmp/dg$i
  if timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set -d /tmp/dg$i --output-format csv -- $STEP > $O/o$i.txt 2>&1; then
    dirs="$dirs /tmp/dg$i"; echo "set $i ok" >> $O/progress.txt
  else
    echo "set $i FAILED: $set" >> $O/progress.txt
  fi
  i=$((i+1))
done
python $R/scratch/pmc_to_json.py $O/diag.json k_intersect_group 3 $dirs > /dev/null
cat $O/progress.txt
