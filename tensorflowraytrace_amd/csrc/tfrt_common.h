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
__device__ __forceinline__ int rank_below(unsigned long long mask) {
  return __popcll(mask & ((1ull << lane_id()) - 1ull));
}

inline int cdiv(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

}  // namespace tfrt
