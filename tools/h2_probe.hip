// Probe for the compensated two-plane f16 mode (SG_PREC_F16X2): x = hi + lo with hi = f16(x), lo = f16(x - hi), products
// A_hi.W_hi + A_hi.W_lo + A_lo.W_hi accumulated in f32 by v_mfma_f32_16x16x32_f16.
// Questions answered on the hardware (no ISA manual at hand):
//   1. does v_cvt_f16_f32 produce f16 subnormals (lo is unscaled, so small lo values live there)?
//   2. does the f16 MFMA honour subnormal A / B inputs or flush them?
//   3. what does the three-product form deliver against an f64 dot product on random data (K = 1024), next to plain f16?
//   hipcc --offload-arch=gfx950 -O2 tools/h2_probe.hip -o tools/h2_probe && tools/h2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe_subnormal(float* out) {
  const int l = threadIdx.x;
  // 1: conversion of 3e-6 (below the f16 normal range 6.1e-5)
  const _Float16 tiny = (_Float16)3.0e-6f;
  if (l == 0) out[0] = (float)tiny;
  // 2: A = 2^-20 everywhere (subnormal in f16), B = 1024 -> every product 2^-10, 32 products per output: expect 32 * 2^-10 = 0.03125
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1024.0f; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  if (l == 0) out[1] = acc[0];
  // the same with the subnormal on the B port
  acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc, 0, 0, 0);
  if (l == 0) out[2] = acc[0];
}

// one wave: C[16,16] = A[16,K] . B[16,K]^T with the split form (mode 1) or plain f16 (mode 0); A, B f32 in HBM
__global__ void probe_dot(const float* A, const float* B, int K, float* C, int mode) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 32) {
    f16x8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
      const float av = A[r * K + k0 + 8 * g + j], bv = B[r * K + k0 + 8 * g + j];
      ah[j] = (_Float16)av; al[j] = (_Float16)(av - (float)ah[j]);
      bh[j] = (_Float16)bv; bl[j] = (_Float16)(bv - (float)bh[j]);
    }
    // operands swapped as in the library's GEMMs (B rows on the first port): lane ends up with 4 consecutive columns of one row
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc, 0, 0, 0);
    if (mode) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acc, 0, 0, 0);
    }
  }
  // D layout: col = lane & 15 (= A row, second port), row = (lane >> 4) * 4 + e (= B row, first port)
  for (int e = 0; e < 4; ++e) C[r * 16 + g * 4 + e] = acc[e];
}

int main() {
  float* d; hipMalloc(&d, 64);
  probe_subnormal<<<1, 64>>>(d);
  float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  printf("cvt(3e-6) -> %.9g (0 = flushed)\n", h[0]);
  printf("mfma f16 subnormal on port 1: %.9g, on port 2: %.9g (expect 0.03125; 0 = flushed)\n", h[1], h[2]);
  for (int trial = 0; trial < 3; ++trial) {
    const int K = 1024;
    const float wscale = trial == 0 ? 1.0f : trial == 1 ? 0.03f : 8.0f;      // activations O(1) against weights of O(1) / O(0.03) / O(8)
    std::vector<float> A(16 * K), B(16 * K);
    srand(123 + trial);
    auto rnd = []() { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.0f; };   // ~N(0,1)
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd() * wscale;
    float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
      probe_dot<<<1, 64>>>(dA, dB, K, dC, mode);
      float C[256]; hipMemcpy(C, dC, 1024, hipMemcpyDeviceToHost);
      double worst = 0, worst32 = 0, norm = 0;
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double ref = 0; float f32 = 0.f;
        for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[j * K + k]; f32 = fmaf(A[i * K + k], B[j * K + k], f32); }
        worst = fmax(worst, fabs(C[i * 16 + j] - ref)); worst32 = fmax(worst32, fabs(f32 - ref)); norm = fmax(norm, fabs(ref));
      }
      printf("w scale %g  %s: max|err| = %.3e (f32 fmaf chain: %.3e), max|ref| = %.3g\n", wscale, mode ? "split 3-MFMA" : "plain f16   ", worst, worst32, norm);
    }
  }
  return 0;
}
