// Bandwidth of the per-ray kernels' access pattern: R rows read + R rows written per ray, one 4-byte
// element per lane and row (SoA, what k_react3d / k_backward3d do) against 16-byte accesses (4 rays
// per lane) and against one 32-byte record per ray (AoS).   hipcc --offload-arch=gfx950 -O3 soa_copy.hip -o soa_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int R>
__global__ void k_soa4(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = in[r * n + i];
#pragma unroll
  for (int r = 0; r < R; ++r) out[r * n + i] = v[r] * 1.0001f;
}
template <int R>
__global__ void k_soa16(const float4* __restrict__ in, float4* __restrict__ out, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = in[r * n4 + i];
#pragma unroll
  for (int r = 0; r < R; ++r) { v[r].x *= 1.0001f; out[r * n4 + i] = v[r]; }
}
// AoS: R floats per ray as R/4 float4
template <int Q>
__global__ void k_aos(const float4* __restrict__ in, float4* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float4 v[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) v[q] = in[i * Q + q];
#pragma unroll
  for (int q = 0; q < Q; ++q) { v[q].x *= 1.0001f; out[i * Q + q] = v[q]; }
}
int main() {
  const int64_t n = 1 << 20;
  constexpr int R = 8;
  float *a, *b;
  hipMalloc(&a, n * R * 4 * 8); hipMalloc(&b, n * R * 4 * 8);
  hipMemset(a, 0, n * R * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto f, const char* name, double bytes) {
    for (int k = 0; k < 3; ++k) f();
    hipEventRecord(e0);
    for (int k = 0; k < 20; ++k) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-28s %7.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
  };
  const double bytes = 2.0 * n * R * 4;
  time([&] { hipLaunchKernelGGL((k_soa4<R>), dim3(n / 256), dim3(256), 0, 0, a, b, n); }, "SoA 8 rows, 4 B per lane", bytes);
  time([&] { hipLaunchKernelGGL((k_soa16<R>), dim3(n / 4 / 256), dim3(256), 0, 0, (float4*)a, (float4*)b, n / 4); }, "SoA 8 rows, 16 B per lane", bytes);
  time([&] { hipLaunchKernelGGL((k_aos<2>), dim3(n / 256), dim3(256), 0, 0, (float4*)a, (float4*)b, n); }, "AoS 32 B per ray", bytes);
  // the same with 4M rays (where launch ramp matters less)
  const int64_t n4m = n * 4;
  const double bytes4 = 2.0 * n4m * R * 4;
  time([&] { hipLaunchKernelGGL((k_soa4<R>), dim3(n4m / 256), dim3(256), 0, 0, a, b, n4m); }, "4M: SoA 8 rows, 4 B", bytes4);
  time([&] { hipLaunchKernelGGL((k_soa16<R>), dim3(n4m / 4 / 256), dim3(256), 0, 0, (float4*)a, (float4*)b, n4m / 4); }, "4M: SoA 8 rows, 16 B", bytes4);
  time([&] { hipLaunchKernelGGL((k_aos<2>), dim3(n4m / 256), dim3(256), 0, 0, (float4*)a, (float4*)b, n4m); }, "4M: AoS 32 B per ray", bytes4);
  return 0;
}
