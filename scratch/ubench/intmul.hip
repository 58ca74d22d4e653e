// Integer multiply issue rates on gfx950 (what Philox4x32 is made of): 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed * (i + 1) + threadIdx.x; b[i] = a[i] ^ 0x9E3779B9u; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) { const uint64_t p = (uint64_t)0xD2511F53u * a[i]; a[i] = (uint32_t)(p >> 32) ^ b[i]; b[i] = (uint32_t)p; }  // as compiled
        if (MODE == 1) { uint32_t lo, hi; asm volatile("v_mul_lo_u32 %0, %2, %3\n\tv_mul_hi_u32 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(a[i]), "v"(0xD2511F53u)); a[i] = hi ^ b[i]; b[i] = lo; }
        if (MODE == 2) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u)); }
        if (MODE == 3) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u)); }
        if (MODE == 4) { a[i] = a[i] ^ b[i]; b[i] += a[i]; }
      }
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + b[i];
  if (s == 12345u) out[0] = s;
}
template <int MODE>
void run(const char* name, double instr_per_op) {
  uint32_t* d; hipMalloc(&d, 4);
  const int iters = 500, blocks = 256 * 8;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double ops = (double)blocks * 4 * iters * 64;      // per wave: iters * 64 ops
  const double per_simd = ops / 1024 / (ms * 1e-3);
  printf("%-34s %8.3f ms  %.3e ops/s/SIMD  cycles per op @2.4 GHz = %.1f\n", name, ms, per_simd, 2.4e9 / per_simd);
  hipFree(d);
}
int main() {
  run<0>("philox half-round (compiler)", 1);
  run<1>("philox half-round (mul_lo+mul_hi)", 1);
  run<2>("v_mul_hi_u32", 1);
  run<3>("v_mul_lo_u32", 1);
  run<4>("xor + add", 1);
  return 0;
}
