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
python $R/scratch/pmc_to_json.py $O/${RN}_pmc_backward.json k_backward_chain 1 $D > /dev/null
python $R/scratch/pmc_to_json.py $O/${RN}_pmc_accumulate.json k_face_accumulate 1 $D > /dev/null
# 3. the bench line itself, now that the PMC files exist (bench withholds the counter-derived
#    fields when the kernel sources have changed since the counters were collected)
cp $O/${RN}_pmc_*.json $R/profiles/
python $R/bench.py > $O/${RN}_bench_line.json 2> $O/bench_err.log
tail -c 4000 $O/${RN}_bench_line.json
# 4. the 2-D kernels on cfg5b (4M rays x 320 primitives, 4 passes, forward + reverse sweep)
STEP2="python $R/scratch/perf_2d.py 4000000"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/p2d$i; rocprofv3 --kernel-trace --pmc $set -d /tmp/p2d$i --output-format csv -- $STEP2 > /tmp/o2d$i.txt 2>&1
  i=$((i+1))
done
D2="/tmp/p2d0 /tmp/p2d1 /tmp/p2d2 /tmp/p2d3"
export PMC_WORKLOAD="cfg5b: 2-D, 4,000,000 rays x (256 segments + 64 arcs), 4 passes, f32 state, ops.trace2d forward (+ reverse sweep in the second half of the run), rocprofv3 --kernel-trace --pmc, one counter set per run"
python $R/scratch/pmc_to_json.py $O/${RN}_pmc_intersect2d.json k_intersect2d 4 $D2 > /dev/null
python $R/scratch/pmc_to_json.py $O/${RN}_pmc_react2d.json k_react2d 4 $D2 > /dev/null
python $R/scratch/pmc_to_json.py $O/${RN}_pmc_backward2d.json k_backward2d 4 $D2 > /dev/null
rm -rf /tmp/ks2d; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2d -- $STEP2 > $O/perf2d.txt 2>&1
cp $(ls /tmp/ks2d/*/*kernel_stats.csv | head -1) $O/${RN}_cfg5b_kernel_stats.csv
# 5. k_intersect_group on a NATURAL-order 1M-ray step (what a caller without an order runs)
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/png$i; TFRT_COHERENT=0 rocprofv3 --kernel-trace --pmc $set -d /tmp/png$i --output-format csv -- $STEP > /tmp/ong$i.txt 2>&1
  i=$((i+1))
done
export PMC_WORKLOAD="cfg4 in NATURAL ray order (OpticalEngine(coherent=False)): 1,000,000 rays x 10,574 faces, f32 state, eager fused step, rocprofv3 --kernel-trace --pmc, one counter set per run"
python $R/scratch/pmc_to_json.py $O/${RN}_pmc_group.json "k_intersect_group<" 3 /tmp/png0 /tmp/png1 /tmp/png2 /tmp/png3 > /dev/null
cp $O/${RN}_pmc_group.json $R/profiles/
unset PMC_WORKLOAD
