// VALU issue-rate microbenchmark for gfx950: independent chains, 8 accumulators per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[8];
  float2_t p[8];
  half2_t h = {(_Float16)seed, (_Float16)(seed * 0.5f)};
  half2_t g = {(_Float16)(seed * 0.25f), (_Float16)(seed * 0.125f)};
  for (int i = 0; i < 8; ++i) { a[i] = seed * (i + 1) + threadIdx.x; p[i] = {a[i], a[i] * 0.5f}; }
  const float b = seed * 1.0001f, c = seed * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) a[i] = __builtin_fmaf(a[i], b, c);
        if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], (float2_t){b, b}, (float2_t){c, c});
        if (MODE == 2) a[i] = __builtin_amdgcn_fdot2(h, g, a[i], false);
        if (MODE == 3) a[i] = __builtin_fminf(a[i], b + i);
        if (MODE == 4) a[i] = a[i] * b;
        if (MODE == 5) { asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(h), "v"(g)); }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
double run(const char* name, double flop_per_op, int waves_per_simd = 8) {
  float* d; hipMalloc(&d, 4);
  const int iters = 2000, blocks = 256 * waves_per_simd;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double wave_instr = (double)blocks * 4 * iters * 64;   // per wave: iters*64 instrs
  double per_simd_per_s = wave_instr / 1024 / (ms * 1e-3);
  printf("%-22s %8.3f ms  %.3e wave-instr/s/SIMD  (cycles/instr @2.0GHz = %.2f, @2.4 = %.2f)  %.1f TFLOP/s\n", name, ms,
         per_simd_per_s, 2.0e9 / per_simd_per_s, 2.4e9 / per_simd_per_s, wave_instr * 64 * flop_per_op / (ms * 1e-3) / 1e12);
  hipFree(d);
  return ms;
}

int main() {
  run<0>("v_fma_f32", 2);
  run<1>("v_pk_fma_f32", 4);
  run<2>("v_dot2_f32_f16", 4);
  run<3>("v_min_f32", 1);
  run<4>("v_mul_f32", 1);
  run<5>("v_dot2c_f32_f16", 4);
  // the same FMA stream with fewer waves per SIMD (256 blocks of 4 waves = 1 wave per SIMD)
  for (int w : {1, 2, 3, 4, 5, 6}) {
    char name[64];
    snprintf(name, sizeof name, "v_fma_f32, %d waves/SIMD", w);
    run<0>(name, 2, w);
  }
  return 0;
}
