// f32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32): the PARITY-mode engine.
// The instruction is bit-for-bit a k-ordered fmaf chain, so results agree with the reference's
// fp32 CPU path up to summation order.  Fully general: any M, N, K, any strides for B
// (W[N,K] "transposed" form or [K,N] row-major, e.g. P.V with V[key,dv]), two-level batching
// (image, head).  Used for every linear in SG_PREC_F32 and for the materialised attention
// (scores = q.k^T, ctx = P.v) of that mode; throughput mode never touches this file.
#include "common.h"

namespace sg {

constexpr int FM = 128, FN = 128, FK = 16, FPAD = 4;

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args a) {
  __shared__ float sA[FK][FM + FPAD];
  __shared__ float sB[FK][FN + FPAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int tiles_n = (a.N + FN - 1) / FN;
  const int m0 = (blockIdx.x / tiles_n) * FM, n0 = (blockIdx.x % tiles_n) * FN;
  const int zo = blockIdx.y / a.inner, zi = blockIdx.y % a.inner;
  const float* A = a.A + zo * a.sAo + zi * a.sAi;
  const float* B = a.B + zo * a.sBo + zi * a.sBi;
  float* C = a.C + zo * a.sCo + zi * a.sCi;
  const float* R = a.residual ? a.residual + zo * a.sCo + zi * a.sCi : nullptr;
  const bool b_n_fast = (a.sbn == 1 && a.sbk != 1);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int k0 = 0; k0 < a.K; k0 += FK) {
    // A tile: k fastest across threads (A is K-contiguous)
    {
      const int k = tid & 15;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = (tid >> 4) + 16 * i;
        float v = 0.f;
        if (m0 + m < a.M && k0 + k < a.K) v = A[(int64_t)(m0 + m) * a.lda + k0 + k];
        sA[k][m] = v;
      }
    }
    if (b_n_fast) {
      const int n = tid & 127;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = (tid >> 7) + 2 * i;
        float v = 0.f;
        if (n0 + n < a.N && k0 + k < a.K) v = B[(int64_t)(k0 + k) * a.sbk + (int64_t)(n0 + n) * a.sbn];
        sB[k][n] = v;
      }
    } else {
      const int k = tid & 15;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int n = (tid >> 4) + 16 * i;
        float v = 0.f;
        if (n0 + n < a.N && k0 + k < a.K) v = B[(int64_t)(k0 + k) * a.sbk + (int64_t)(n0 + n) * a.sbn];
        sB[k][n] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < FK / 2; ++ks) {
      const int k = 2 * ks + (lane >> 5);
      float fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = sA[k][wave_m * 64 + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = sB[k][wave_n * 64 + j * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j], fa[i], acc[i][j], 0, 0, 0);   // D[n][m]
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wave_m * 64 + i * 32 + (lane & 31);
    if (m >= a.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wave_n * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n >= a.N) continue;
        float x = acc[i][j][r] * a.alpha;
        if (a.bias) x += a.bias[n];
        if (a.act == ACT_QUICK_GELU) x = quick_gelu_exact(x);
        else if (a.act == ACT_GELU) x = erf_gelu(x);
        if (R) x += R[(int64_t)m * a.ldr + n];
        C[(int64_t)m * a.ldc + n] = x;
      }
    }
  }
}

int gemm_f32(const GemmF32Args& a, hipStream_t s) {
  SG_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.batch > 0 && a.inner > 0, "gemm_f32: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  const int64_t tiles = cdiv(a.M, FM) * cdiv(a.N, FN);
  SG_REQUIRE(tiles < (1ll << 31) && a.batch < 65536, "gemm_f32: grid too large");
  prof_begin(PROF_GEMM_F32, 2.0 * a.M * (double)a.N * a.K * a.batch, s);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)tiles, (unsigned)a.batch), dim3(256), 0, s, a);
  prof_end(PROF_GEMM_F32, s);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

}  // namespace sg
