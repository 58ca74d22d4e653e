#!/bin/bash
# Collects the committed profile artefacts of the default bench into gpurun_out/profiles_new/.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_new; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/bench_line.json 2> $O/bench_err.log
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu-baseline --no-extra-legs > $O/bench_prof_line.json 2>/dev/null
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
export ONE_PASS=1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc; rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc --output-format csv -- python $R/scratch/perf_group.py group > /tmp/o.txt 2>&1
  python $R/scratch/pmc_median.py /tmp/pmc k_intersect_group >> $O/pmc_group.txt
done
python $R/scratch/perf_group.py group | tail -1 >> $O/pmc_group.txt
cat $O/pmc_group.txt
