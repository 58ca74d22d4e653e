// Shared device helpers for the tfrt HIP kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tfrt_hip.h"
#include "trace_math.h"

namespace tfrt {

constexpr int BLOCK = 256;       // 4 waves of 64
constexpr int WAVES = BLOCK / 64;

template <typename T>
__device__ __forceinline__ double ldd(const T* p, int64_t i) {
  return static_cast<double>(p[i]);
}

// One ray of a SoA ray block (ROWS rows of `stride` elements).
template <typename T>
__device__ __forceinline__ void load_ray3(const T* rays, int64_t stride, int64_t i, double s[3],
                                          double e[3]) {
  s[0] = ldd(rays, i);
  s[1] = ldd(rays, stride + i);
  s[2] = ldd(rays, 2 * stride + i);
  e[0] = ldd(rays, 3 * stride + i);
  e[1] = ldd(rays, 4 * stride + i);
  e[2] = ldd(rays, 5 * stride + i);
}

template <typename T>
__device__ __forceinline__ void store_ray3(T* rays, int64_t stride, int64_t i, const double s[3],
                                           const double e[3]) {
  rays[i] = static_cast<T>(s[0]);
  rays[stride + i] = static_cast<T>(s[1]);
  rays[2 * stride + i] = static_cast<T>(s[2]);
  rays[3 * stride + i] = static_cast<T>(e[0]);
  rays[4 * stride + i] = static_cast<T>(e[1]);
  rays[5 * stride + i] = static_cast<T>(e[2]);
}

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// number of set bits of `mask` strictly below this lane
// (v_mbcnt_lo + v_mbcnt_hi: the hardware's masked bit count below the lane, two instructions;
// `popc(mask & lanes_below)` compiles to four)
__device__ __forceinline__ int rank_below(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Ordering point between LDS accesses of ONE wave that communicate across lanes.  The LDS
// executes a wave's instructions in issue order, so no wait is needed; this only keeps the
// compiler from moving memory accesses across it.
__device__ __forceinline__ void wave_fence() {
  __asm__ volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

inline int cdiv(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Exclusive scan of per-ray-block histograms (rows of NB counters, NB = 4 or 8) over a grid of
// 1024-thread workgroups, one row per thread: coalesced int4 loads and stores, a wave scan by
// shuffles and one exchange through LDS.  `off` receives the offsets *within the row's
// workgroup*; the workgroup that finishes last (a ticket) turns the per-workgroup sums `rowtot`
// into the bases `rowbase` (the consumer adds rowbase[(row >> 10) * NB + c]) and gets the column
// sums in `total` (LDS, NB ints): the function returns true there and false everywhere else.
// One workgroup scanning everything could not keep enough loads in flight: 54 us for the
// 15.6k rows of a 4M-ray pass.  `ticket` must be 0 on entry and is left 0.
template <int NB>
__device__ __forceinline__ bool scan_rows_grid(const int32_t* __restrict__ cnt,
                                               int32_t* __restrict__ off, int nrows,
                                               int32_t* __restrict__ rowtot,
                                               int32_t* __restrict__ rowbase,
                                               unsigned int* __restrict__ ticket, int* total) {
  static_assert(NB == 4 || NB == 8, "rows are one or two int4");
  constexpr int Q = NB / 4;
  __shared__ int wsum[16][NB];
  __shared__ int wbase[16][NB];
  __shared__ int is_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 1024 + (int)threadIdx.x;
  const int4* cnt4 = reinterpret_cast<const int4*>(cnt);
  int4* off4 = reinterpret_cast<int4*>(off);
  int v[NB], pre[NB];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int4 x = r < nrows ? cnt4[(int64_t)r * Q + q] : make_int4(0, 0, 0, 0);
    v[4 * q] = x.x;
    v[4 * q + 1] = x.y;
    v[4 * q + 2] = x.z;
    v[4 * q + 3] = x.w;
  }
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    int x = v[c];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(x, d, 64);
      if (lane >= d) x += o;
    }
    pre[c] = x - v[c];  // exclusive within the wave
    if (lane == 63) wsum[wave][c] = x;
  }
  __syncthreads();
  if (threadIdx.x < NB) {
    int run = 0;
    for (int w = 0; w < 16; ++w) {
      wbase[w][threadIdx.x] = run;
      run += wsum[w][threadIdx.x];
    }
    rowtot[blockIdx.x * NB + threadIdx.x] = run;
  }
  __syncthreads();
  if (r < nrows) {
#pragma unroll
    for (int q = 0; q < Q; ++q)
      off4[(int64_t)r * Q + q] =
          make_int4(wbase[wave][4 * q] + pre[4 * q], wbase[wave][4 * q + 1] + pre[4 * q + 1],
                    wbase[wave][4 * q + 2] + pre[4 * q + 2], wbase[wave][4 * q + 3] + pre[4 * q + 3]);
  }
  // the last workgroup to arrive sees every rowtot (release / acquire through the ticket)
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(ticket, 1u);
    is_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return false;
  __threadfence();
  if (threadIdx.x < NB) {
    int run = 0;
    for (unsigned int b = 0; b < gridDim.x; ++b) {
      rowbase[b * NB + threadIdx.x] = run;
      run += __atomic_load_n(&rowtot[b * NB + threadIdx.x], __ATOMIC_RELAXED);
    }
    total[threadIdx.x] = run;
  }
  if (threadIdx.x == 0) *ticket = 0u;  // ready for the next pass
  __syncthreads();
  return true;
}

}  // namespace tfrt
