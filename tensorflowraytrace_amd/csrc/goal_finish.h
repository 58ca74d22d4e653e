// Second stage of the built-in error sum (tfrt_goal_error3d): the per-workgroup partial sums in a
// fixed order, by ONE workgroup of BLOCK threads.  Shared by k_goal_finish (tfrt_error.hip) and
// k_sgd_process_multi (tfrt_update.hip), which can do it in a spare workgroup of its own launch
// (tfrt_goal_error3d_deferred / tfrt_sgd_process_multi_finish: one dependent launch less per
// optimiser step).
#pragma once
#include "tfrt_common.h"

namespace tfrt {

struct GoalFields {
  int32_t n;
  int32_t row[6];  // row of the ray block (0..5: x_start .. z_end) compared with goal column c
};

__device__ __forceinline__ void goal_finish_block(const tfrt_goal_pending& g) {
#pragma clang fp contract(off)
  __shared__ double wsum[WAVES];
  // (eight loads in flight per thread: one workgroup reads up to n_rays / 64 partial sums, and a
  // chain of dependent load-add pairs took 16 us for 15,625 of them; the order stays fixed.  The
  // counts of an in-place trace that compacted nothing -- the sweep counted finished rays and
  // passes itself -- ride along in the same rounds: a second loop of their own cost 5 us more)
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long c0 = 0, c1 = 0;
  const bool counted = g.partial_counts != nullptr;
  const int2* pc = reinterpret_cast<const int2*>(g.partial_counts);
  int b = threadIdx.x;
  if (counted) {   // (two loops, no branch between the loads of a round: a conditional load inside
                   // the round made every load wait for the one before it, 8 -> 22 us)
    for (; b + 7 * BLOCK < g.n_partial; b += 8 * BLOCK) {
      double v[8];
      int2 w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = g.partial[b + k * BLOCK];
        w[k] = pc[b + k * BLOCK];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a[k] += v[k];
        c0 += w[k].x;
        c1 += w[k].y;
      }
    }
  } else {
    for (; b + 7 * BLOCK < g.n_partial; b += 8 * BLOCK) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = g.partial[b + k * BLOCK];
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += v[k];
    }
  }
  for (int k = 0; b < g.n_partial; b += BLOCK, ++k) {
    a[k] += g.partial[b];
    if (counted) {
      const int2 w = pc[b];
      c0 += w.x;
      c1 += w.y;
    }
  }
  double s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  if (lane_id() == 0) wsum[threadIdx.x >> 6] = s;
  __shared__ long long wcnt[WAVES][2];
  if (counted) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      c0 += __shfl_xor(c0, d, 64);
      c1 += __shfl_xor(c1, d, 64);
    }
    if (lane_id() == 0) {
      wcnt[threadIdx.x >> 6][0] = c0;
      wcnt[threadIdx.x >> 6][1] = c1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < WAVES; ++w) tot += wsum[w];
    long long n_fin, tests = 0;
    bool have_tests = false;
    if (g.partial_counts != nullptr) {
      long long passes = 0;
      n_fin = 0;
      for (int w = 0; w < WAVES; ++w) {
        n_fin += wcnt[w][0];
        passes += wcnt[w][1];
      }
      tests = passes * (long long)g.n_faces;
      have_tests = true;
      if (g.counts_tail != nullptr) {
        g.counts_tail[1] = (int32_t)n_fin;
        g.counts_tail[4] = (int32_t)(uint32_t)((unsigned long long)tests & 0xFFFFFFFFull);
        g.counts_tail[5] = (int32_t)(uint32_t)((unsigned long long)tests >> 32);
      }
    } else {
      n_fin = *g.n_finished;
      if (g.tests_lo_hi != nullptr) {
        tests = (long long)((unsigned long long)(uint32_t)g.tests_lo_hi[0] |
                            ((unsigned long long)(uint32_t)g.tests_lo_hi[1] << 32));
        have_tests = true;
      }
    }
    const double terms = (double)n_fin * (double)g.n_fields;
    g.error_out[0] = tot;
    g.error_out[1] = terms;
    // reduce_mean of optimizer.py:257 (no finished ray: the mean of nothing is NaN there too)
    g.error_out[2] = terms > 0.0 ? tot / terms : __builtin_nan("");
    if (g.tests_total != nullptr && have_tests) *g.tests_total += tests;
  }
}

}  // namespace tfrt
