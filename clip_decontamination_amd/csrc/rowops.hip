// Row-wise (HBM-bound) ops of the ViT: LayerNorm, token assembly (+class token, +positional
// embedding, ln_pre), positional-embedding resize, f32->bf16 packing, row L2-normalisation and the
// materialised softmax of parity mode.  One wavefront (64 lanes) per row, float4 loads, shuffle
// reductions -- no LDS, no atomics.
#include "rowops.h"

namespace sg {

constexpr int LN_MAX_VEC = 8;      // float4 per lane kept in registers: D <= 64 * 4 * 8 = 2048

// elements e .. e+3 (e % 4 == 0) of the row starting at `row`
template <typename OutT>
__device__ __forceinline__ void store4(OutT* row, int e, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* row, int e, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(row + e) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* row, int e, float a, float b, float c, float d) {
  uint2 o; o.x = pack_bf2(a, b); o.y = pack_bf2(c, d);
  *reinterpret_cast<uint2*>(row + e) = o;
}
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* row, int e, float a, float b, float c, float d) {
  uint2 o; o.x = pack_h2(a, b); o.y = pack_h2(c, d);
  *reinterpret_cast<uint2*>(row + e) = o;
}
template <> __device__ __forceinline__ void store4<h2_t>(h2_t* row, int e, float a, float b, float c, float d) {   // half a storage group: 8 B hi, 8 B lo
  uint2 hi, lo;
  split_h2(a, b, hi.x, lo.x); split_h2(c, d, hi.y, lo.y);
  uint16_t* g = reinterpret_cast<uint16_t*>(row) + ((e >> 3) << 4) + (e & 7);
  *reinterpret_cast<uint2*>(g) = hi; *reinterpret_cast<uint2*>(g + 8) = lo;
}

// Normalise the row held in v[] (nvec float4 per lane) -- reference LayerNorm/LayerNormFp32
// (open_clip/transformer.py:17-32): biased variance, eps inside the sqrt, f32 arithmetic.
template <typename OutT>
__device__ __forceinline__ void ln_row(float4 (&v)[LN_MAX_VEC], int D, int lane, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float eps, OutT* __restrict__ out) {
  const int nv = D >> 2;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i)
    if (lane + 64 * i < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum_dpp(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i)
    if (lane + 64 * i < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i) {
    const int idx = lane + 64 * i;
    if (idx < nv) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * idx);
      const float4 b = *reinterpret_cast<const float4*>(beta + 4 * idx);
      store4<OutT>(out, 4 * idx, (v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                   (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
    }
  }
}

template <typename OutT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, OutT* __restrict__ y, int64_t ldy,
                                                        int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float4 v[LN_MAX_VEC];
  const int nv = D >> 2;
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i)
    if (lane + 64 * i < nv) v[i] = *reinterpret_cast<const float4*>(xr + 4 * (lane + 64 * i));
  ln_row<OutT>(v, D, lane, gamma, beta, eps, y + row * ldy);
}

int layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy, int y_is_bf16,
              int64_t rows, int D, float eps, hipStream_t s) {
  SG_REQUIRE(D % 4 == 0 && D <= 64 * 4 * LN_MAX_VEC, "layernorm: D=%d must be a multiple of 4 and <= %d", D, 64 * 4 * LN_MAX_VEC);
  SG_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "layernorm: row strides must be multiples of 4");
  if (rows == 0) return SG_OK;
  dim3 grid((unsigned)cdiv(rows, 4));
  if (y_is_bf16 == HK_F16X2) {
    SG_REQUIRE(D % 8 == 0 && ldy % 8 == 0, "layernorm: two-plane f16 output needs D and the row stride to be multiples of 8");
    hipLaunchKernelGGL(layernorm_kernel<h2_t>, grid, dim3(256), 0, s, x, ldx, gamma, beta, (h2_t*)y, ldy, rows, D, eps);
  } else if (y_is_bf16 == HK_F16) hipLaunchKernelGGL(layernorm_kernel<f16_t>, grid, dim3(256), 0, s, x, ldx, gamma, beta, (f16_t*)y, ldy, rows, D, eps);
  else if (y_is_bf16) hipLaunchKernelGGL(layernorm_kernel<bf16_t>, grid, dim3(256), 0, s, x, ldx, gamma, beta, (bf16_t*)y, ldy, rows, D, eps);
  else hipLaunchKernelGGL(layernorm_kernel<float>, grid, dim3(256), 0, s, x, ldx, gamma, beta, (float*)y, ldy, rows, D, eps);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// ---- fp8 (OCP e4m3) row quantisation ------------------------------------------------------------------------------------------
// q[r,:] = e4m3(v[r,:] / scale[r]),  scale[r] = max|v[r,:]| / 448  (per-token dynamic scale; per-output-channel for weights).
// v_cvt_pk_fp8_f32 rounds to nearest even; |v| / scale <= 448 by construction, so nothing saturates.
constexpr float FP8_MAX = 448.0f;
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}

// LayerNorm whose output goes straight to fp8 + a per-row scale (the A operand of the fp8 QKV / fc GEMMs): one wave per row
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, uint8_t* __restrict__ y, int64_t ldy,
                                                            float* __restrict__ scale, int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float4 v[LN_MAX_VEC];
  const int nv = D >> 2;
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i)
    if (lane + 64 * i < nv) v[i] = *reinterpret_cast<const float4*>(xr + 4 * (lane + 64 * i));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i)
    if (lane + 64 * i < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum_dpp(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i)
    if (lane + 64 * i < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) / (float)D + eps);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i) {
    const int idx = lane + 64 * i;
    if (idx < nv) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * idx);
      const float4 b = *reinterpret_cast<const float4*>(beta + 4 * idx);
      v[i] = make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y, (v[i].z - mean) * rstd * g.z + b.z,
                         (v[i].w - mean) * rstd * g.w + b.w);
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
    }
  }
  amax = wave_max_dpp(amax);
  const float sc = amax > 0.f ? amax / FP8_MAX : 1.0f;
  const float inv = 1.0f / sc;
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i) {
    const int idx = lane + 64 * i;
    if (idx < nv) *reinterpret_cast<uint32_t*>(y + row * ldy + 4 * idx) = pack_fp8x4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
  }
  if (lane == 0) scale[row] = sc;
}

int layernorm_fp8(const float* x, int64_t ldx, const float* gamma, const float* beta, uint8_t* y, int64_t ldy, float* scale, int64_t rows,
                  int D, float eps, hipStream_t s) {
  SG_REQUIRE(D % 4 == 0 && D <= 64 * 4 * LN_MAX_VEC, "layernorm_fp8: D=%d must be a multiple of 4 and <= %d", D, 64 * 4 * LN_MAX_VEC);
  SG_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "layernorm_fp8: row strides must be multiples of 4");
  if (rows == 0) return SG_OK;
  hipLaunchKernelGGL(layernorm_fp8_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, scale, rows, D, eps);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// rows of f32 or bf16 -> fp8 + per-row scale (weights at load time; the GELU output in front of the fp8 proj GEMM): one wave per row
template <typename T>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const T* __restrict__ x, int64_t ldx, uint8_t* __restrict__ y, int64_t ldy,
                                                                float* __restrict__ scale, int64_t rows, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * ldx;
  float amax = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
#pragma unroll
    for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(to_f32<T>(xr[c + e])));
  }
  amax = wave_max_dpp(amax);
  const float sc = amax > 0.f ? amax / FP8_MAX : 1.0f;
  const float inv = 1.0f / sc;
  for (int c = lane * 4; c < D; c += 256)
    *reinterpret_cast<uint32_t*>(y + row * ldy + c) =
        pack_fp8x4(to_f32<T>(xr[c]) * inv, to_f32<T>(xr[c + 1]) * inv, to_f32<T>(xr[c + 2]) * inv, to_f32<T>(xr[c + 3]) * inv);
  if (lane == 0) scale[row] = sc;
}

int quantize_rows_fp8(const void* x, int x_is_bf16, int64_t ldx, uint8_t* y, int64_t ldy, float* scale, int64_t rows, int D, hipStream_t s) {
  SG_REQUIRE(D % 4 == 0 && ldy % 4 == 0, "quantize_rows_fp8: D=%d and the output stride must be multiples of 4", D);
  if (rows == 0) return SG_OK;
  dim3 grid((unsigned)cdiv(rows, 4));
  if (x_is_bf16 == HK_F16) hipLaunchKernelGGL(quantize_rows_fp8_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)x, ldx, y, ldy, scale, rows, D);
  else if (x_is_bf16) hipLaunchKernelGGL(quantize_rows_fp8_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ldx, y, ldy, scale, rows, D);
  else hipLaunchKernelGGL(quantize_rows_fp8_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ldx, y, ldy, scale, rows, D);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// x[b,t,:] = ln_pre( (t == 0 ? class_embedding : patch_embed[b, t-1, :]) + pos[t, :] )
// reference open_clip/transformer.py:565-574
__global__ __launch_bounds__(256) void embed_assemble_kernel(const float* __restrict__ patches, int64_t ldp,
                                                             const float* __restrict__ cls_emb, const float* __restrict__ pos,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ x, int B, int N, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * N) return;
  const int b = (int)(row / N), t = (int)(row % N);
  const float* src = (t == 0) ? cls_emb : patches + ((int64_t)b * (N - 1) + (t - 1)) * ldp;
  const float* pr = pos + (int64_t)t * D;
  float4 v[LN_MAX_VEC];
  const int nv = D >> 2;
#pragma unroll
  for (int i = 0; i < LN_MAX_VEC; ++i)
    if (lane + 64 * i < nv) {
      const float4 a = *reinterpret_cast<const float4*>(src + 4 * (lane + 64 * i));
      const float4 p = *reinterpret_cast<const float4*>(pr + 4 * (lane + 64 * i));
      v[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
  ln_row<float>(v, D, lane, gamma, beta, eps, x + row * D);
}

int embed_assemble(const float* patches, int64_t ldp, const float* cls_emb, const float* pos, const float* gamma,
                   const float* beta, float* x, int B, int N, int D, float eps, hipStream_t s) {
  SG_REQUIRE(D % 4 == 0 && D <= 64 * 4 * LN_MAX_VEC && ldp % 4 == 0, "embed_assemble: unsupported D=%d ldp=%lld", D, (long long)ldp);
  hipLaunchKernelGGL(embed_assemble_kernel, dim3((unsigned)cdiv((int64_t)B * N, 4)), dim3(256), 0, s, patches, ldp, cls_emb, pos,
                     gamma, beta, x, B, N, D, eps);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// ---- positional-embedding resize ------------------------------------------------------------------
// mode 0: F.interpolate(scale_factor=(g+0.1)/g0, mode='bicubic') -- open_clip/transformer.py:777-795
//         (A = -0.75, align_corners=False, source index not clamped, taps clamped)
// mode 1: F.interpolate(size=, mode='bicubic', antialias=True)   -- gem/gem_utils.py:12-43
//         (separable PIL-style filter, A = -0.5, window truncated at the border and renormalised)
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__device__ __forceinline__ int taps_plain(int dst, float scale, int in, int* idx, float* w) {
  const float A = -0.75f;
  const float src = scale * ((float)dst + 0.5f) - 0.5f;
  const float fl = floorf(src);
  const float t = src - fl;
  const int i0 = (int)fl;
  w[0] = cubic2(t + 1.f, A); w[1] = cubic1(t, A); w[2] = cubic1(1.f - t, A); w[3] = cubic2(2.f - t, A);
#pragma unroll
  for (int k = 0; k < 4; ++k) { int j = i0 - 1 + k; idx[k] = j < 0 ? 0 : (j > in - 1 ? in - 1 : j); }
  return 4;
}

constexpr int AA_MAX_TAPS = 24;
__device__ __forceinline__ float aa_filter(float x) {
  const float A = -0.5f;
  x = fabsf(x);
  if (x < 1.f) return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * A;
  return 0.f;
}
__device__ __forceinline__ int taps_aa(int dst, float scale, int in, int* idx, float* w) {
  const float support = (scale >= 1.f) ? 2.f * scale : 2.f;
  const float invscale = (scale >= 1.f) ? 1.f / scale : 1.f;
  const float center = scale * ((float)dst + 0.5f);
  int xmin = (int)(center - support + 0.5f); if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5f); if (xmax > in) xmax = in;
  int n = xmax - xmin; if (n > AA_MAX_TAPS) n = AA_MAX_TAPS;
  float tot = 0.f;
  for (int j = 0; j < n; ++j) { w[j] = aa_filter(((float)(j + xmin) - center + 0.5f) * invscale); tot += w[j]; idx[j] = xmin + j; }
  if (tot != 0.f) for (int j = 0; j < n; ++j) w[j] /= tot;
  return n;
}

__global__ void posembed_resize_kernel(const float* __restrict__ pos, int g0, int D, int gh, int gw, float scale_h, float scale_w,
                                       int mode, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)(gh * gw + 1) * D;
  if (i >= total) return;
  const int d = (int)(i % D), t = (int)(i / D);
  if (t == 0) { out[i] = pos[d]; return; }                       // class position kept
  const int oy = (t - 1) / gw, ox = (t - 1) % gw;
  int iy[AA_MAX_TAPS], ix[AA_MAX_TAPS];
  float wy[AA_MAX_TAPS], wx[AA_MAX_TAPS];
  const int ny = mode ? taps_aa(oy, scale_h, g0, iy, wy) : taps_plain(oy, scale_h, g0, iy, wy);
  const int nx = mode ? taps_aa(ox, scale_w, g0, ix, wx) : taps_plain(ox, scale_w, g0, ix, wx);
  float acc = 0.f;
  for (int a = 0; a < ny; ++a) {
    float row = 0.f;
    for (int b = 0; b < nx; ++b) row += wx[b] * pos[(int64_t)(1 + iy[a] * g0 + ix[b]) * D + d];
    acc += wy[a] * row;
  }
  out[i] = acc;
}

int posembed_resize(const float* pos, int g0, int D, int gh, int gw, int antialias, float* out, hipStream_t s) {
  float sh, sw;
  if (antialias) { sh = (float)g0 / (float)gh; sw = (float)g0 / (float)gw; }
  else {   // scale = 1 / scale_factor, scale_factor = (g + 0.1) / g0 evaluated in double as Python does
    sh = (float)(1.0 / (((double)gh + 0.1) / (double)g0));
    sw = (float)(1.0 / (((double)gw + 0.1) / (double)g0));
  }
  if (antialias) SG_REQUIRE(2.f * fmaxf(fmaxf(sh, sw), 1.f) * 2.f + 2.f <= AA_MAX_TAPS, "posembed_resize: downscale %f too large", sh);
  const int64_t total = (int64_t)(gh * gw + 1) * D;
  hipLaunchKernelGGL(posembed_resize_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, pos, g0, D, gh, gw, sh, sw, antialias, out);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// ---- f32 -> bf16 pack with optional zero padding of the row (weights, patch matrix K padding) ------
template <typename OutT>
__global__ void pack_half_kernel(const float* __restrict__ src, int64_t rows, int cols, int64_t ld_src, OutT* __restrict__ dst,
                                 int cols_pad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols_pad) return;
  const int64_t r = i / cols_pad; const int c = (int)(i % cols_pad);
  dst[i] = from_f32<OutT>(c < cols ? src[r * ld_src + c] : 0.f);
}
__global__ void pack_f32_kernel(const float* __restrict__ src, int64_t rows, int cols, int64_t ld_src, float* __restrict__ dst,
                                int cols_pad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols_pad) return;
  const int64_t r = i / cols_pad; const int c = (int)(i % cols_pad);
  dst[i] = c < cols ? src[r * ld_src + c] : 0.f;
}
// two-plane f16: one thread per storage group (8 elements -> 32 bytes)
__global__ void pack_h2_kernel(const float* __restrict__ src, int64_t rows, int cols, int64_t ld_src, h2_t* __restrict__ dst, int cols_pad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int gpr = cols_pad >> 3;
  if (i >= rows * gpr) return;
  const int64_t r = i / gpr; const int c0 = (int)(i % gpr) * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = c0 + j < cols ? src[r * ld_src + c0 + j] : 0.f;
  store_h2x8(dst + r * cols_pad + c0, v);
}
int pack_rows(const float* src, int64_t rows, int cols, int64_t ld_src, void* dst, int cols_pad, int to_bf16, hipStream_t s) {
  const int64_t total = rows * cols_pad;
  if (total == 0) return SG_OK;
  if (to_bf16 == HK_F16X2) {
    SG_REQUIRE(cols_pad % 8 == 0, "pack_rows: two-plane f16 rows are multiples of 8 elements (got %d)", cols_pad);
    hipLaunchKernelGGL(pack_h2_kernel, dim3((unsigned)cdiv(total / 8, 256)), dim3(256), 0, s, src, rows, cols, ld_src, (h2_t*)dst, cols_pad);
  } else if (to_bf16 == HK_F16) hipLaunchKernelGGL(pack_half_kernel<f16_t>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, src, rows, cols, ld_src, (f16_t*)dst, cols_pad);
  else if (to_bf16) hipLaunchKernelGGL(pack_half_kernel<bf16_t>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, src, rows, cols, ld_src, (bf16_t*)dst, cols_pad);
  else hipLaunchKernelGGL(pack_f32_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, src, rows, cols, ld_src, (float*)dst, cols_pad);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// ---- LayerNorm folded into the neighbouring GEMMs -----------------------------------------------------------------------------------
// Chan's combination of the per-slice (sum, centred sum of squares): exact to f32 rounding whatever the mean is.
// st is slice-major [S][rows][2] (what the producing epilogue writes as full lines): one thread per row, coalesced 8-byte loads.
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ st, int64_t rows, int S, float eps, float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float2* p = reinterpret_cast<const float2*>(st) + r;
  float tot = 0.f;
  for (int i = 0; i < S; ++i) tot += p[(int64_t)i * rows].x;
  const float mean = tot / (float)(S * 64);
  float m2 = 0.f;
  for (int i = 0; i < S; ++i) { const float2 v = p[(int64_t)i * rows]; const float d = v.x * (1.0f / 64.0f) - mean; m2 += v.y + 64.0f * d * d; }
  reinterpret_cast<float2*>(out)[r] = make_float2(mean, 1.0f / sqrtf(m2 / (float)(S * 64) + eps));
}
int ln_stats_finalize(const float* slice_stats, int64_t rows, int D, float eps, float* mean_rstd, hipStream_t s) {
  SG_REQUIRE(D % 64 == 0 && rows > 0, "ln_stats_finalize: D=%d must be a multiple of 64", D);
  hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((unsigned)cdiv(rows, 256)), dim3(256), 0, s, slice_stats, rows, D / 64, eps, mean_rstd);
  SG_LAUNCH_CHECK();
  return SG_OK;
}
// one wave per output channel n
template <typename OutT>
__global__ __launch_bounds__(256) void fold_ln_weight_kernel(const float* __restrict__ W, int N, int K, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias, OutT* __restrict__ Wp,
                                                             float* __restrict__ c, float* __restrict__ bias_f) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float cs = 0.f, bs = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = W[(int64_t)n * K + k];
    const OutT h = from_f32<OutT>(w * gamma[k]);
    Wp[(int64_t)n * K + k] = h;
    cs += to_f32<OutT>(h);
    bs += beta[k] * w;
  }
  cs = wave_sum(cs); bs = wave_sum(bs);
  if (lane == 0) { c[n] = cs; bias_f[n] = (bias ? bias[n] : 0.f) + bs; }
}
// two-plane f16: W' = hi + lo carries 22 bits; c sums exactly the two planes the GEMM multiplies with
__global__ __launch_bounds__(256) void fold_ln_weight_h2_kernel(const float* __restrict__ W, int N, int K, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ bias, h2_t* __restrict__ Wp,
                                                                float* __restrict__ c, float* __restrict__ bias_f) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float cs = 0.f, bs = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = W[(int64_t)n * K + k];
    st_elem<h2_t>(Wp + (int64_t)n * K, k, w * gamma[k]);
    cs += ld_elem<h2_t>(Wp + (int64_t)n * K, k);                          // the same thread's own store
    bs += beta[k] * w;
  }
  cs = wave_sum(cs); bs = wave_sum(bs);
  if (lane == 0) { c[n] = cs; bias_f[n] = (bias ? bias[n] : 0.f) + bs; }
}
int fold_ln_weight(const float* W, int N, int K, const float* gamma, const float* beta, const float* bias, int hk, void* Wp, float* c,
                   float* bias_f, hipStream_t s) {
  SG_REQUIRE(hk == HK_BF16 || hk == HK_F16 || hk == HK_F16X2, "fold_ln_weight: 2-byte and two-plane compute dtypes only");
  const dim3 grid((unsigned)cdiv(N, 4));
  if (hk == HK_F16X2) {
    SG_REQUIRE(K % 8 == 0, "fold_ln_weight: two-plane f16 rows are multiples of 8 elements");
    hipLaunchKernelGGL(fold_ln_weight_h2_kernel, grid, dim3(256), 0, s, W, N, K, gamma, beta, bias, (h2_t*)Wp, c, bias_f);
  } else if (hk == HK_F16) hipLaunchKernelGGL(fold_ln_weight_kernel<f16_t>, grid, dim3(256), 0, s, W, N, K, gamma, beta, bias, (f16_t*)Wp, c, bias_f);
  else hipLaunchKernelGGL(fold_ln_weight_kernel<bf16_t>, grid, dim3(256), 0, s, W, N, K, gamma, beta, bias, (bf16_t*)Wp, c, bias_f);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// transpose-pack: dst[c][r] = src[r][c]   (proj [D,E] -> W[E,D] so `x @ proj` becomes a W[N,K]^T GEMM)
__global__ void transpose_pack_kernel(const float* __restrict__ src, int rows, int cols, void* __restrict__ dst, int to_bf16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int c = (int)(i / rows), r = (int)(i % rows);
  const float v = src[(int64_t)r * cols + c];
  if (to_bf16 == HK_F16X2) st_elem<h2_t>((h2_t*)dst + (int64_t)c * rows, r, v);      // rows % 8 == 0 (checked on the host)
  else if (to_bf16 == HK_F16) ((f16_t*)dst)[i] = f2h(v); else if (to_bf16) ((bf16_t*)dst)[i] = f2bf(v); else ((float*)dst)[i] = v;
}
int transpose_pack(const float* src, int rows, int cols, void* dst, int to_bf16, hipStream_t s) {
  if (to_bf16 == HK_F16X2) SG_REQUIRE(rows % 8 == 0, "transpose_pack: two-plane f16 rows are multiples of 8 elements (got %d)", rows);
  hipLaunchKernelGGL(transpose_pack_kernel, dim3((unsigned)cdiv((int64_t)rows * cols, 256)), dim3(256), 0, s, src, rows, cols, dst, to_bf16);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// ---- row L2 normalisation: y = x / max(||x||, eps)  (F.normalize, eps 1e-12) --------------------------
// Rows are (outer, inner) indexed: row r -> x + (r / inner) * so + (r % inner) * si, length D contiguous.
template <typename InT, typename OutT>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const InT* __restrict__ x, int64_t so, int64_t si, int inner,
                                                          OutT* __restrict__ y, int64_t yo, int64_t yi, int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const InT* xr = x + (row / inner) * so + (row % inner) * si;
  OutT* yr = y + (row / inner) * yo + (row % inner) * yi;
  float ss = 0.f;
  for (int i = lane; i < D; i += 64) { const float v = ld_elem<InT>(xr, i); ss += v * v; }
  const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), eps);
  for (int i = lane; i < D; i += 64) st_elem<OutT>(yr, i, ld_elem<InT>(xr, i) * inv);
}
// Short rows (head vectors: D <= 128, D % 8 == 0): LPR = 8 or 16 lanes per row, 8 elements per lane, the sum of squares over the row's lanes on
// DPP adds -- 8 / 4 rows per wave instead of one row per wave with a single element per lane and six ds_bpermute round trips (GEM's q / k / v
// normalisations: 665 us per call at 119 tiles of ViT-L/14, 10.6 % of the BASELINE config-3 step).
template <typename InT, typename OutT, int LPR>
__global__ __launch_bounds__(256) void l2norm_rows_short_kernel(const InT* __restrict__ x, int64_t so, int64_t si, int inner,
                                                                OutT* __restrict__ y, int64_t yo, int64_t yi, int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63, sub = lane / LPR, l = lane % LPR;
  constexpr int RPW = 64 / LPR;
  int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
  const bool live = row < rows;
  row = live ? row : rows - 1;
  const InT* xr = x + (row / inner) * so + (row % inner) * si;
  OutT* yr = y + (row / inner) * yo + (row % inner) * yi;
  const bool mine = 8 * l < D;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = mine ? ld_elem<InT>(xr, 8 * l + e) : 0.f;
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
  ss = LPR == 8 ? sum8_dpp(ss) : sum16_dpp(ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), eps);
  if (live && mine) {
#pragma unroll
    for (int e = 0; e < 8; ++e) st_elem<OutT>(yr, 8 * l + e, v[e] * inv);
  }
}
template <typename InT, typename OutT>
static void l2norm_launch_short(const InT* x, int64_t so, int64_t si, int inner, OutT* y, int64_t yo, int64_t yi, int64_t rows, int D, float eps, hipStream_t s) {
  if (D <= 64) hipLaunchKernelGGL((l2norm_rows_short_kernel<InT, OutT, 8>), dim3((unsigned)cdiv(rows, 32)), dim3(256), 0, s, x, so, si, inner, y, yo, yi, rows, D, eps);
  else hipLaunchKernelGGL((l2norm_rows_short_kernel<InT, OutT, 16>), dim3((unsigned)cdiv(rows, 16)), dim3(256), 0, s, x, so, si, inner, y, yo, yi, rows, D, eps);
}
template <typename InT>
static void l2norm_launch_out(const InT* x, int64_t so, int64_t si, int inner, void* y, int y_kind, int64_t yo, int64_t yi, int64_t rows, int D,
                              float eps, dim3 grid, hipStream_t s) {
  if (D <= 128 && D % 8 == 0 && rows >= 4096) {             // many short rows
    if (y_kind == HK_F16X2) l2norm_launch_short<InT, h2_t>(x, so, si, inner, (h2_t*)y, yo, yi, rows, D, eps, s);
    else if (y_kind == HK_F16) l2norm_launch_short<InT, f16_t>(x, so, si, inner, (f16_t*)y, yo, yi, rows, D, eps, s);
    else if (y_kind == HK_BF16) l2norm_launch_short<InT, bf16_t>(x, so, si, inner, (bf16_t*)y, yo, yi, rows, D, eps, s);
    else l2norm_launch_short<InT, float>(x, so, si, inner, (float*)y, yo, yi, rows, D, eps, s);
    return;
  }
  if (y_kind == HK_F16X2) hipLaunchKernelGGL((l2norm_rows_kernel<InT, h2_t>), grid, dim3(256), 0, s, x, so, si, inner, (h2_t*)y, yo, yi, rows, D, eps);
  else if (y_kind == HK_F16) hipLaunchKernelGGL((l2norm_rows_kernel<InT, f16_t>), grid, dim3(256), 0, s, x, so, si, inner, (f16_t*)y, yo, yi, rows, D, eps);
  else if (y_kind == HK_BF16) hipLaunchKernelGGL((l2norm_rows_kernel<InT, bf16_t>), grid, dim3(256), 0, s, x, so, si, inner, (bf16_t*)y, yo, yi, rows, D, eps);
  else hipLaunchKernelGGL((l2norm_rows_kernel<InT, float>), grid, dim3(256), 0, s, x, so, si, inner, (float*)y, yo, yi, rows, D, eps);
}
int l2norm_rows(const void* x, int x_bf16, int64_t so, int64_t si, int inner, void* y, int y_bf16, int64_t yo, int64_t yi,
                int64_t rows, int D, float eps, hipStream_t s) {
  if (rows == 0) return SG_OK;
  dim3 grid((unsigned)cdiv(rows, 4));
  if (x_bf16 == HK_F16X2 || y_bf16 == HK_F16X2)
    SG_REQUIRE(so % 8 == 0 && si % 8 == 0 && yo % 8 == 0 && yi % 8 == 0, "l2norm_rows: two-plane f16 rows must start on multiples of 8 elements");
  if (x_bf16 == HK_F16X2) l2norm_launch_out<h2_t>((const h2_t*)x, so, si, inner, y, y_bf16, yo, yi, rows, D, eps, grid, s);
  else if (x_bf16 == HK_F16) l2norm_launch_out<f16_t>((const f16_t*)x, so, si, inner, y, y_bf16, yo, yi, rows, D, eps, grid, s);
  else if (x_bf16 == HK_BF16) l2norm_launch_out<bf16_t>((const bf16_t*)x, so, si, inner, y, y_bf16, yo, yi, rows, D, eps, grid, s);
  else l2norm_launch_out<float>((const float*)x, so, si, inner, y, y_bf16, yo, yi, rows, D, eps, grid, s);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// ---- materialised softmax of parity mode ----------------------------------------------------------------
// scores [rows, N] (row stride ld) hold RAW dot products.  Per row r (token i = r % N of image b = r / (H*N)):
//   mode 0: p = softmax(scale * s + w * bias)            (bias row from sim[b, i-1, :] shifted by the CLS column)
//   mode 1: p = softmax( softmax(scale * s) + w * bias ) (the 'Experimental' double softmax, transformer.py:896-902)
// out (+)= p.  lse (optional) receives log-sum-exp of (scale*s [+ w*bias]) for mode 0.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ scores, int64_t ld, int64_t rows, int N, int H,
                                                           const float* __restrict__ scale_per_image, float scale,
                                                           const float* __restrict__ bias, float bias_w, int64_t bias_bstride,
                                                           const float* __restrict__ bias_rn, const float* __restrict__ bias_cn,
                                                           int mode, int accumulate, float* __restrict__ out, float* __restrict__ lse, int causal) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int i = (int)(row % N);
  const int64_t b = row / ((int64_t)H * N);
  const float sc = scale_per_image ? scale_per_image[b] : scale;
  const float* sr = scores + row * ld;
  float* orow = out + row * ld;
  const int n = N - 1;
  const int Nk = causal ? i + 1 : N;                      // causal: only keys <= query take part; the rest get probability 0
  const float* brow = (bias && i > 0) ? bias + b * bias_bstride + (int64_t)(i - 1) * n : nullptr;   // bias[b, i-1, :]
  const int hd = (int)((row / N) % H);
  const float* cn = bias_cn ? bias_cn + (b * H + hd) * (int64_t)N : nullptr;
  if (bias_rn) bias_w *= bias_rn[(b * H + hd) * (int64_t)N + i];
  float mx = -INFINITY;
  for (int j = lane; j < Nk; j += 64) {
    float v = sr[j] * sc;
    if (mode == 0 && brow && j > 0) v += bias_w * brow[j - 1] * (cn ? cn[j] : 1.f);
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < Nk; j += 64) {
    float v = sr[j] * sc;
    if (mode == 0 && brow && j > 0) v += bias_w * brow[j - 1] * (cn ? cn[j] : 1.f);
    sum += expf(v - mx);
  }
  sum = wave_sum(sum);
  if (lse && lane == 0) lse[row] = mx + logf(sum);
  const float inv = 1.0f / sum;
  if (mode == 0) {
    for (int j = lane; j < Nk; j += 64) {
      float v = sr[j] * sc;
      if (brow && j > 0) v += bias_w * brow[j - 1] * (cn ? cn[j] : 1.f);
      const float p = expf(v - mx) * inv;
      orow[j] = accumulate ? orow[j] + p : p;
    }
    if (!accumulate) for (int j = Nk + lane; j < N; j += 64) orow[j] = 0.f;
    return;
  }
  float mx2 = -INFINITY;
  for (int j = lane; j < Nk; j += 64) {
    float p = expf(sr[j] * sc - mx) * inv;
    if (brow && j > 0) p += bias_w * brow[j - 1] * (cn ? cn[j] : 1.f);
    mx2 = fmaxf(mx2, p);
  }
  mx2 = wave_max(mx2);
  float sum2 = 0.f;
  for (int j = lane; j < Nk; j += 64) {
    float p = expf(sr[j] * sc - mx) * inv;
    if (brow && j > 0) p += bias_w * brow[j - 1] * (cn ? cn[j] : 1.f);
    sum2 += expf(p - mx2);
  }
  sum2 = wave_sum(sum2);
  const float inv2 = 1.0f / sum2;
  for (int j = lane; j < Nk; j += 64) {
    float p = expf(sr[j] * sc - mx) * inv;
    if (brow && j > 0) p += bias_w * brow[j - 1] * (cn ? cn[j] : 1.f);
    const float q = expf(p - mx2) * inv2;
    orow[j] = accumulate ? orow[j] + q : q;
  }
}

int softmax_rows(const float* scores, int64_t ld, int64_t rows, int N, int H, const float* scale_per_image, float scale,
                 const float* bias, float bias_w, int64_t bias_bstride, const float* bias_rn, const float* bias_cn, int mode,
                 int accumulate, float* out, float* lse, hipStream_t s, int causal) {
  if (rows == 0) return SG_OK;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, scores, ld, rows, N, H, scale_per_image,
                     scale, bias, bias_w, bias_bstride, bias_rn, bias_cn, mode, accumulate, out, lse, causal);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// ---- Gaussian neighbourhood bias of NACLIP / NOnly / GAV (reference open_clip/transformer.py:797-820,909-917) ------------------
// omega[(y,x),(y',x')] = exp(-((y-y')^2 + (x-x')^2) / (2 std^2)) over the patch grid (the CLS row / column is zero and not stored)
__global__ void gaussian_bias_kernel(int gh, int gw, float inv2s2, float* __restrict__ omega) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = gh * gw;
  if (i >= (int64_t)n * n) return;
  const int a = (int)(i / n), b = (int)(i % n);
  const float dy = (float)(a / gw - b / gw), dx = (float)(a % gw - b % gw);
  omega[i] = expf(-(dy * dy + dx * dx) * inv2s2);
}
int gaussian_bias(int gh, int gw, float std, float* omega, hipStream_t s) {
  const int64_t total = (int64_t)gh * gw * gh * gw;
  hipLaunchKernelGGL(gaussian_bias_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, gh, gw, 1.0f / (2.0f * std * std), omega);
  SG_LAUNCH_CHECK();
  return SG_OK;
}
// out[b,h,t] = || x[b,t,h,:] ||   (x: one of the q/k/v slices of the packed qkv)
template <typename T>
__global__ __launch_bounds__(256) void head_norms_kernel(const T* __restrict__ x, int64_t sb, int64_t st, int N, int H, int dh, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // (b, t, h)
  const int b = blockIdx.y;
  if (row >= (int64_t)N * H) return;
  const int t = (int)(row / H), hd = (int)(row % H);
  const T* p = x + (int64_t)b * sb + (int64_t)t * st + hd * dh;
  float ss = 0.f;
  for (int i = lane; i < dh; i += 64) { const float v = ld_elem<T>(p, i); ss += v * v; }
  ss = wave_sum(ss);
  if (lane == 0) out[((int64_t)b * H + hd) * N + t] = sqrtf(ss);
}
int head_norms(const void* x, int is_bf16, int64_t sb, int64_t st, int B, int N, int H, int dh, float* out, hipStream_t s) {
  dim3 grid((unsigned)cdiv((int64_t)N * H, 4), (unsigned)B);
  if (is_bf16 == HK_F16X2) hipLaunchKernelGGL(head_norms_kernel<h2_t>, grid, dim3(256), 0, s, (const h2_t*)x, sb, st, N, H, dh, out);
  else if (is_bf16 == HK_F16) hipLaunchKernelGGL(head_norms_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)x, sb, st, N, H, dh, out);
  else if (is_bf16) hipLaunchKernelGGL(head_norms_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, sb, st, N, H, dh, out);
  else hipLaunchKernelGGL(head_norms_kernel<float>, grid, dim3(256), 0, s, (const float*)x, sb, st, N, H, dh, out);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// ---- small element-wise helpers -------------------------------------------------------------------------
__global__ void axpby_kernel(float* __restrict__ y, const float* __restrict__ x, float a, float b, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i] + b * y[i];
}
int axpby(float* y, const float* x, float a, float b, int64_t n, hipStream_t s) {
  if (n == 0) return SG_OK;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, y, x, a, b, n);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// inv_temp[b] = mean_t ||x[b,t,:]|| * scale   (gem/gem_utils.py:79-81)
__global__ __launch_bounds__(256) void gem_inv_temp_kernel(const float* __restrict__ x, int N, int D, float scale, float* __restrict__ out) {
  __shared__ float part[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int t = wave; t < N; t += 4) {
    const float* r = x + ((int64_t)b * N + t) * D;
    float ss = 0.f;
    for (int i = lane; i < D; i += 64) ss += r[i] * r[i];
    acc += sqrtf(wave_sum(ss));
  }
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[b] = (part[0] + part[1] + part[2] + part[3]) / (float)N * scale;
}
int gem_inv_temp(const float* x, int B, int N, int D, float scale, float* out, hipStream_t s) {
  hipLaunchKernelGGL(gem_inv_temp_kernel, dim3(B), dim3(256), 0, s, x, N, D, scale, out);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

}  // namespace sg
