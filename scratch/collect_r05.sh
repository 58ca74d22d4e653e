#!/bin/bash
# Round 5: every measured artefact under profiles/ (run on the GPU box: bash scratch/collect_r05.sh).
# Output: gpurun_out/profiles_r05/ -- copy into profiles/ afterwards.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_r05; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
STEP="python $R/scratch/prof_step.py 1000000 fused 30"
# 1. PMC counters first (bench.py reads them), one counter set per run, kernel-trace only
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc$i; rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc$i --output-format csv -- $STEP > /tmp/o$i.txt 2>&1
  i=$((i+1))
done
D="/tmp/pmc0 /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 /tmp/pmc4"
export PMC_WORKLOAD="bench.py default (cfg4: 1,000,000 rays x 10,574 faces, f32 state), eager fused step over the in-place trace (ONE k_trace_inplace launch per step: all three passes), rocprofv3 --kernel-trace --pmc, one counter set per run"
python $R/scratch/pmc_to_json.py $O/r05_pmc_inplace.json "k_trace_inplace<" 1 $D > /dev/null
python $R/scratch/pmc_to_json.py $O/r05_pmc_backward.json k_backward_chain 1 $D > /dev/null
unset PMC_WORKLOAD
cp $O/r05_pmc_*.json $R/profiles/
echo "pmc done"
# 2. kernel stats of the bench command itself
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu-baseline --no-extra-legs > $O/bench_prof_line.json 2>/dev/null
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/r05_bench_kernel_stats.csv
echo "kernel stats done"
# 3. the bench line (with the PMC files in place) and the other configurations
cd $R
python bench.py > $O/r05_bench_line.json 2> $O/bench_err.log
echo "bench done"
for c in cfg2 cfg3 cfg5a cfg5b; do python bench.py --config $c > $O/r05_bench_$c.json 2> $O/bench_$c.err; echo "$c done"; done
# 4. step times by ray count
( printf "%9s %9s %9s %9s %9s   ms/step\n" rays generic rowwise fused graph
  for n in 1000000 500000 250000 125000; do
    row=""
    for m in generic rowwise fused graph; do
      v=$(python scratch/prof_step.py $n $m 100 2>/dev/null | grep "ms/step" | sed 's/.*: \([0-9.]*\) ms.*/\1/')
      row="$row $(printf '%9s' $v)"
    done
    printf "%9d%s\n" $n "$row"
  done ) > $O/r05_step_times.txt
cat $O/r05_step_times.txt
# 5. per-kernel tables of the other steps
bash scratch/kstats_step.sh 1000000 graph random 20 > $O/r05_random_source_kernel_stats.txt 2>&1
bash scratch/kstats_step.sh 1000000 generic "" 24 > $O/r05_generic_step_kernel_stats.txt 2>&1
bash scratch/kstats_step.sh 1000000 rowwise "" 24 > $O/r05_rowwise_step_kernel_stats.txt 2>&1
# 6. the in-place kernel from the inside (tuning build, never shipped)
if [ -f scratch/variants_live/lib_ticks.so ]; then
  TFRT_LIB_PATH=scratch/variants_live/lib_ticks.so python scratch/inplace_ticks.py 1000000 > $O/r05_wave_timeline.txt 2>&1
fi
ls -la $O
