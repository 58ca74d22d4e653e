#!/bin/bash
# Collects the profile artefacts of the default bench into gpurun_out/profiles_new/ and installs the
# PMC files under profiles/ (copy the rest by hand).  Run on the GPU box: bash scratch/collect_profiles.sh [ROUND]
set -e
RN=${1:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_new; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
STEP="python $R/scratch/prof_step.py 1000000 fused 30"
# 1. kernel stats of the bench command itself (graph replays and the eager profiling leg)
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu-baseline --no-extra-legs > $O/bench_prof_line.json 2>/dev/null
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/${RN}_bench_kernel_stats.csv
# 2. PMC counters, one set per run (SQ: 8 slots; TCC: FETCH_SIZE and WRITE_SIZE apart), kernel-trace only
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc$i; rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc$i --output-format csv -- $STEP > /tmp/o$i.txt 2>&1
  i=$((i+1))
done
D="/tmp/pmc0 /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 /tmp/pmc4"
python $R/scratch/pmc_to_json.py $O/${RN}_pmc_beam.json k_intersect_beam 3 $D > /dev/null
python $R/scratch/pmc_to_json.py $O/${RN}_pmc_group.json k_intersect_group 3 $D > /dev/null
python $R/scratch/pmc_to_json.py $O/${RN}_pmc_react.json k_react3d 3 $D > /dev/null
python $R/scratch/pmc_to_json.py $O/${RN}_pmc_backward.json k_backward3d 3 $D > /dev/null
python $R/scratch/pmc_to_json.py $O/${RN}_pmc_accumulate.json k_face_accumulate 1 $D > /dev/null
# 3. the bench line itself, now that the PMC files exist (bench withholds the counter-derived
#    fields when the kernel sources have changed since the counters were collected)
cp $O/${RN}_pmc_*.json $R/profiles/
python $R/bench.py > $O/${RN}_bench_line.json 2> $O/bench_err.log
tail -c 4000 $O/${RN}_bench_line.json
