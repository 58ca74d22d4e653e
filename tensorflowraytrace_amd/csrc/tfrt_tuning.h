// Tuning builds only (-DTFRT_TUNING; scratch/build_variants.py): funnel counters of
// k_intersect_group.  Never part of the shipped library: __graft_entry__.build() and
// tensorflowraytrace_amd/_build.py do not define TFRT_TUNING.
//   [0] level-0 tests, [1] (ray, supercluster) pairs, [2] queued clusters, [3] member-sphere hits,
//   [4] pairs past the float32 screen, [5] float64 decisions that hit.
// k_intersect_beam: [8] wavefronts, [9] left to the grouped kernel because the directions spread,
//   [10] / [11] / [12] because more than BEAM_SLIST superclusters / BEAM_CLIST clusters /
//   BEAM_FLIST faces were touched, [13] candidate faces (member spheres touched), [14] (ray, face)
//   pairs queued for the float64 decision, [15] first ray of the last wavefront left over,
//   [27] faces past face_frame, [28] decision batches, [29] faces walked, [30] bundles tried.
#pragma once

__device__ unsigned long long g_group_stats[32];
#define TFRT_STAT(k, v) \
  do { if (lane_id() == 0) atomicAdd(&g_group_stats[k], (unsigned long long)(v)); } while (0)

// stage clocks of k_intersect_beam (-DTFRT_TICKS on top of -DTFRT_TUNING; the counters above are
// switched off then: their contended atomics would be what is measured): [16 + k] = shader-clock
// ticks wavefronts spent in stage k, accumulated over 256 slots per stage to keep the atomics apart
__device__ unsigned long long g_ticks[16][256];
// ... and each wavefront's first and last instant on the 100 MHz wall clock (last launch wins)
__device__ unsigned long long g_wave_t0[65536], g_wave_t1[65536], g_wave_info[65536];
#ifdef TFRT_TICKS
#define TFRT_WAVE_BEGIN const unsigned long long _w0 = wall_clock64(); unsigned _wi[4] = {0u, 0u, 0u, 0u}
#define TFRT_WAVE_NOTE(k, v) _wi[k] += (unsigned)(v)
#define TFRT_WAVE_END(qw)                                                    \
  do {                                                                       \
    if (lane_id() == 0) {                                                    \
      g_wave_t0[(qw) & 65535] = _w0;                                         \
      g_wave_t1[(qw) & 65535] = wall_clock64();                              \
      g_wave_info[(qw) & 65535] = (unsigned long long)(_wi[0] & 0xFFFFu) << 48 | \
          (unsigned long long)(_wi[1] & 0xFFFFu) << 32 | (unsigned long long)(_wi[2] & 0xFFFFu) << 16 | \
          (unsigned long long)(_wi[3] & 0xFFFFu);                            \
    }                                                                        \
  } while (0)
#undef TFRT_STAT
#define TFRT_STAT(k, v) do { } while (0)
#define TFRT_TICK_INIT unsigned long long _tick = clock64(); const int _tslot = (blockIdx.x * 4 + (threadIdx.x >> 6)) & 255
#define TFRT_TICK(k)                                                         \
  do {                                                                       \
    const unsigned long long _n = clock64();                                 \
    if (lane_id() == 0) atomicAdd(&g_ticks[k][_tslot], _n - _tick);          \
    _tick = clock64();                                                       \
  } while (0)
#else
#define TFRT_TICK_INIT do { } while (0)
#define TFRT_TICK(k) do { } while (0)
#define TFRT_WAVE_BEGIN do { } while (0)
#define TFRT_WAVE_END(qw) do { } while (0)
#define TFRT_WAVE_NOTE(k, v) do { } while (0)
#endif

// read (and clear) the counters
#define TFRT_TUNING_EXPORTS                                                                       \
  int tfrt_debug_group_stats(unsigned long long* out32) {                                         \
    static unsigned long long zero[32] = {0};                                                     \
    static unsigned long long ticks[16][256], tz[16][256];                                        \
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_group_stats), sizeof(zero)) != hipSuccess)        \
      return -1;                                                                                 \
    if (hipMemcpyFromSymbol(ticks, HIP_SYMBOL(g_ticks), sizeof(ticks)) != hipSuccess) return -1;  \
    for (int k = 0; k < 16; ++k)                                                                  \
      for (int j = 0; j < 256; ++j) out32[16 + k] += ticks[k][j];                                 \
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ticks), tz, sizeof(tz)) != hipSuccess) return -1;          \
    return hipMemcpyToSymbol(HIP_SYMBOL(g_group_stats), zero, sizeof(zero)) == hipSuccess ? 0    \
                                                                                          : -1;  \
  }                                                                                              \
  int tfrt_debug_wave_times(unsigned long long* t0, unsigned long long* t1,                     \
                            unsigned long long* info) { /* 65536 each */                         \
    if (hipMemcpyFromSymbol(t0, HIP_SYMBOL(g_wave_t0), 65536 * 8) != hipSuccess) return -1;       \
    if (hipMemcpyFromSymbol(info, HIP_SYMBOL(g_wave_info), 65536 * 8) != hipSuccess) return -1;   \
    return hipMemcpyFromSymbol(t1, HIP_SYMBOL(g_wave_t1), 65536 * 8) == hipSuccess ? 0 : -1;      \
  }
