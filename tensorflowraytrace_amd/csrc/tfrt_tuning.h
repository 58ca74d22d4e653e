// Tuning builds only (-DTFRT_TUNING; scratch/build_variants.py): funnel counters of
// k_intersect_group.  Never part of the shipped library: __graft_entry__.build() and
// tensorflowraytrace_amd/_build.py do not define TFRT_TUNING.
//   [0] level-0 tests, [1] (ray, supercluster) pairs, [2] queued clusters, [3] member-sphere hits,
//   [4] pairs past the float32 screen, [5] float64 decisions that hit.
// k_intersect_beam: [8] wavefronts, [9] left to the grouped kernel because the directions spread,
//   [10] / [11] / [12] because more than BEAM_SLIST superclusters / BEAM_CLIST clusters /
//   BEAM_FLIST faces were touched, [13] candidate faces, [14] pairs past the screen, [15] decisions.
#pragma once

__device__ unsigned long long g_group_stats[16];
#define TFRT_STAT(k, v) \
  do { if (lane_id() == 0) atomicAdd(&g_group_stats[k], (unsigned long long)(v)); } while (0)

// read (and clear) the counters
#define TFRT_TUNING_EXPORTS                                                                       \
  int tfrt_debug_group_stats(unsigned long long* out16) {                                         \
    unsigned long long zero[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                                      \
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_group_stats), sizeof(zero)) != hipSuccess)        \
      return -1;                                                                                 \
    return hipMemcpyToSymbol(HIP_SYMBOL(g_group_stats), zero, sizeof(zero)) == hipSuccess ? 0    \
                                                                                          : -1;  \
  }
