// Micro-benchmark (tuning aid, not product code): issue rate of the bf16 / fp8 MFMA shapes on gfx950, operands in registers,
// random bit patterns, 8 independent accumulators per wave, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int MODE>
__global__ __launch_bounds__(512) void k(const uint32_t* seed, float* out, int iters) {
  const int tid = threadIdx.x;
  uint32_t s[8];
  for (int i = 0; i < 8; ++i) s[i] = seed[(tid * 8 + i) & 4095];
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  union { uint32_t u[4]; bf16x8 b; } a16, b16;
  for (int i = 0; i < 4; ++i) { a16.u[i] = (s[i] & 0x3fff3fffu) | 0x3c003c00u; b16.u[i] = (s[i + 4] & 0x3fff3fffu) | 0x3c003c00u; }
  long a8 = ((long)(s[0] & 0x3f3f3f3f) << 32) | (s[1] & 0x3f3f3f3f), b8 = ((long)(s[2] & 0x3f3f3f3f) << 32) | (s[3] & 0x3f3f3f3f);
  i32x8 a32, b32;
  for (int i = 0; i < 8; ++i) { a32[i] = s[i] & 0x3f3f3f3f; b32[i] = s[7 - i] & 0x3f3f3f3f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16.b, b16.b, acc[i], 0, 0, 0);
      if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a8, b8, acc[i], 0, 0, 0);
      if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a32, b32, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + tid] = r;
}

int main() {
  uint32_t h[4096];
  uint32_t x = 12345;
  for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = x; }
  uint32_t* seed; float* out;
  CK(hipMalloc(&seed, sizeof(h))); CK(hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 256 * 2 * 512 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  const double flop[3] = {2.0 * 16 * 16 * 32, 2.0 * 16 * 16 * 32, 2.0 * 16 * 16 * 128};
  const char* name[3] = {"bf16 16x16x32", "fp8 16x16x32 (legacy)", "fp8 16x16x128 f8f6f4"};
  for (int mode = 0; mode < 3; ++mode)
    for (int blocks = 256; blocks <= 512; blocks += 256) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, seed, out, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, seed, out, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, seed, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 1) printf("%-24s %d waves/SIMD: %.2f ms, %.0f TFLOP/s\n", name[mode], blocks / 128, ms, flop[mode] * 8.0 * iters * blocks * 8 / (ms * 1e-3) / 1e12);
      }
    }
  return 0;
}
