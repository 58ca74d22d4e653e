#!/bin/bash
# The measured files collect_profiles.sh does not make: step times by ray count, per-kernel tables of the
# random-source and generic steps, the other BASELINE configurations, the beam kernel's wave timeline.
RN=${1:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_more; mkdir -p $O
cd $R
bash scratch/step_times.sh > $O/${RN}_step_times.txt
bash scratch/kstats_step.sh 1000000 graph random 30 > $O/${RN}_random_source_kernel_stats.txt 2>&1
bash scratch/kstats_step.sh 1000000 generic "" 30 > $O/${RN}_generic_step_kernel_stats.txt 2>&1
for c in cfg2 cfg3 cfg5a cfg5b; do python bench.py --config $c > $O/${RN}_bench_$c.json 2> $O/bench_$c.err; done
( for p in 1 2 3; do TFRT_LIB_PATH=scratch/variants_live/lib_ticks.so python scratch/wave_times.py 1000000 $p; done
  TFRT_LIB_PATH=scratch/variants_live/lib_tune.so python scratch/beam_stats.py 1000000
  TFRT_LIB_PATH=scratch/variants_live/lib_ticks.so python scratch/beam_stats.py 1000000
  echo "--- source re-drawn every step"
  TFRT_LIB_PATH=scratch/variants_live/lib_tune.so python scratch/beam_stats.py 1000000 random ) > $O/${RN}_wave_timeline.txt 2>&1
ls -la $O
