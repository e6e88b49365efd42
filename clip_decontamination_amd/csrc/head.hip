// Segmentation head: everything after the vision tower.
//   cosine_logits : segmentor.py:309-336 (CLS normalise, cls_logits, similarity-weighted global debias),
//                   :374-379 (L2 normalise, tokens @ T^T, + lambda * cls_logits)
//   stitch        : segmentor.py:388-391 bilinear upsample of each tile's logits, :436-447 un-pad,
//                   overlap-add, count-normalise -- fused, write-once per canvas pixel (the reference
//                   re-writes the whole canvas once per tile)
//   resize        : F.interpolate(mode='bilinear', align_corners=False) (segmentor.py:449)
//   postprocess   : segmentor.py:475-489
// All HBM-bound; one pass over the data each.
#include "rowops.h"

namespace sg {

// ---- cosine logits -------------------------------------------------------------------------------------------
// 16 lanes per token, four tokens per wave at a time, CL_TPW tokens per wave.  T (Q x E) is staged once per workgroup in LDS.
constexpr int CL_TPW = 8, CL_TPB = 4 * CL_TPW, CL_MAXV = 32;   // E <= 64 * CL_MAXV
template <int NV>      // NV = compile-time bound on E / 64 (features of a token held in NV registers per lane)
__global__ __launch_bounds__(256) void cosine_logits_kernel(const float* __restrict__ tokens, const float* __restrict__ cls,
                                                            const float* __restrict__ text, int n, int E, int Q, float debias,
                                                            float lambda, float* __restrict__ logits) {
  extern __shared__ float sm[];
  float* sT = sm;                    // [Q][E]
  float* sC = sm + (size_t)Q * E;    // [E] unit-norm CLS
  float* sCL = sC + E;               // [Q] cls logits
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool use_cls = cls != nullptr;
  for (int i = tid; i < Q * E; i += 256) sT[i] = text[i];
  if (use_cls) {
    for (int i = tid; i < E; i += 256) sC[i] = cls[(int64_t)b * E + i];
    __syncthreads();
    if (wave == 0) {                                       // cls /= ||cls||  (segmentor.py:310)
      float ss = 0.f;
      for (int i = lane; i < E; i += 64) ss += sC[i] * sC[i];
      const float nrm = sqrtf(wave_sum(ss));
      for (int i = lane; i < E; i += 64) sC[i] = sC[i] / nrm;
    }
    __syncthreads();
    for (int q = wave; q < Q; q += 4) {                    // cls_logits = cls @ T^T  (:311)
      float d = 0.f;
      for (int i = lane; i < E; i += 64) d += sC[i] * sT[q * E + i];
      d = wave_sum(d);
      if (lane == 0) sCL[q] = d;
    }
  }
  __syncthreads();
  // Four tokens per wave at a time (16 lanes each, CL_TPW tokens per wave in all): the staging above is paid once per CL_TPB tokens,
  // a token's features are read from HBM once (one float4 per lane per 64 channels) and kept in registers for the three passes
  // (norm / debias, renormalise, Q dot products), and every reduction is a 4-step shuffle inside the 16-lane group instead of a
  // 6-step wave reduction per token.
  const int sub = lane >> 4, sl = lane & 15;
  auto group_sum = [](float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  const bool deb = use_cls && debias != 0.f;
  for (int tt = 0; tt < CL_TPW; tt += 4) {
    const int t = blockIdx.x * CL_TPB + wave * CL_TPW + tt + sub;
    if (blockIdx.x * CL_TPB + wave * CL_TPW + tt >= n) break;             // wave-uniform: none of the four tokens exists
    const float* f = tokens + ((int64_t)b * n + (t < n ? t : n - 1)) * E;
    float4 x[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = 4 * sl + 64 * k;
      x[k] = i < E ? *reinterpret_cast<const float4*>(f + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // similarity-weighted debias (:322-336): f' = f - cls * (cos(f, cls) * factor); cls already unit norm,
    // the reference renormalises it once more (a no-op up to rounding) -- reproduced for fidelity.
    float ff = 0.f, fc = 0.f, cc = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = 4 * sl + 64 * k;
      if (i < E) {
        ff += (x[k].x * x[k].x + x[k].y * x[k].y) + (x[k].z * x[k].z + x[k].w * x[k].w);
        if (use_cls) {
          const float4 c4 = *reinterpret_cast<const float4*>(sC + i);
          fc += (x[k].x * c4.x + x[k].y * c4.y) + (x[k].z * c4.z + x[k].w * c4.w);
          cc += (c4.x * c4.x + c4.y * c4.y) + (c4.z * c4.z + c4.w * c4.w);
        }
      }
    }
    ff = group_sum(ff);
    float w = 0.f;
    if (deb) {
      fc = group_sum(fc); cc = group_sum(cc);
      w = (fc / (sqrtf(ff) * sqrtf(cc))) * debias;
    }
    float nn = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = 4 * sl + 64 * k;
      if (i < E) {
        if (deb) {
          const float4 c4 = *reinterpret_cast<const float4*>(sC + i);
          x[k].x -= c4.x * w; x[k].y -= c4.y * w; x[k].z -= c4.z * w; x[k].w -= c4.w * w;
        }
        nn += (x[k].x * x[k].x + x[k].y * x[k].y) + (x[k].z * x[k].z + x[k].w * x[k].w);
      }
    }
    const float inv = 1.0f / sqrtf(group_sum(nn));
    for (int q = 0; q < Q; ++q) {
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = 4 * sl + 64 * k;
        if (i < E) {
          const float4 t4 = *reinterpret_cast<const float4*>(sT + q * E + i);
          d += ((x[k].x * inv) * t4.x + (x[k].y * inv) * t4.y) + ((x[k].z * inv) * t4.z + (x[k].w * inv) * t4.w);
        }
      }
      d = group_sum(d);
      if (sl == 0 && t < n) {
        if (use_cls && lambda != 0.f) d += sCL[q] * lambda;
        logits[((int64_t)b * Q + q) * n + t] = d;
      }
    }
  }
}

// ---- bilinear helpers (align_corners=False, ATen area_pixel_compute_source_index) ------------------------------------
__device__ __forceinline__ void bilinear_tap(int dst, int in, int out, int& i0, int& i1, float& l0, float& l1) {
  if (in == out) { i0 = i1 = dst; l0 = 1.f; l1 = 0.f; return; }            // same-size resize is an exact identity
  const float scale = (float)in / (float)out;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src; i0 = i0 > in - 1 ? in - 1 : i0;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0; l0 = 1.f - l1;
}

// ---- stitch -------------------------------------------------------------------------------------------------------
// Write-once: a pixel's contributions are summed in registers in raster order of the tiles (the reference's `preds[...] += `
// order, so the f32 sum is bit-identical), divided by the count and stored a single time.  The row test of a tile is wave-uniform
// (one canvas row per wave) and runs on the scalar unit; queries go through the registers in chunks of ST_QC.
constexpr int ST_QC = 8, ST_MAXC = 64;
__global__ __launch_bounds__(256) void stitch_kernel(const float* __restrict__ tile_logits, const int32_t* __restrict__ windows,
                                                     int T, int Q, int gh, int gw, int up_h, int up_w, int pad_t, int pad_l,
                                                     int H, int W, float* __restrict__ canvas) {
  // candidate tiles of this 64 x 4 pixel block, compacted IN RASTER ORDER by wave 0 (ballot + prefix popcount): the per-pixel
  // loop then visits the handful of overlapping tiles instead of testing all T windows
  __shared__ int s_list[ST_MAXC];
  __shared__ int s_count;
  const int lane = threadIdx.x & 63;
  const int bx0 = blockIdx.x * 64, by0 = blockIdx.y * 4;
  if ((threadIdx.x >> 6) == 0) {
    int base = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
      const int t = t0 + lane;
      bool hit = false;
      if (t < T) {
        const int y1 = windows[t * 4 + 0], y2 = windows[t * 4 + 1], x1 = windows[t * 4 + 2], x2 = windows[t * 4 + 3];
        hit = y1 < by0 + 4 && y2 > by0 && x1 < bx0 + 64 && x2 > bx0;
      }
      const unsigned long long m = __ballot(hit);
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
      if (hit && pos < ST_MAXC) s_list[pos] = t;
      base += __popcll(m);
    }
    if (lane == 0) s_count = base;
  }
  __syncthreads();
  const int n_cand = s_count;
  const bool use_list = n_cand <= ST_MAXC;                         // pathological overlap (stride << crop): test every window
  const int n_iter = use_list ? n_cand : T;
  const int x = bx0 + lane;
  const int y = __builtin_amdgcn_readfirstlane(by0 + (threadIdx.x >> 6));
  if (y >= H) return;
  const int64_t plane = (int64_t)H * W;
  const int64_t tile_sz = (int64_t)gh * gw;
  for (int q0 = 0; q0 < Q; q0 += ST_QC) {
    float acc[ST_QC];
#pragma unroll
    for (int k = 0; k < ST_QC; ++k) acc[k] = 0.f;
    float cnt = 0.f;
    for (int it = 0; it < n_iter; ++it) {                              // raster order = the reference's add order
      const int t = use_list ? s_list[it] : it;
      const int y1 = windows[t * 4 + 0], y2 = windows[t * 4 + 1];
      if (y < y1 || y >= y2) continue;                                 // wave-uniform: the whole wave skips the tile
      const int x1 = windows[t * 4 + 2], x2 = windows[t * 4 + 3];
      if (x < x1 || x >= x2 || x >= W) continue;
      int ya, yb, xa, xb; float wy0, wy1, wx0, wx1;
      bilinear_tap(y - y1 + pad_t, gh, up_h, ya, yb, wy0, wy1);
      bilinear_tap(x - x1 + pad_l, gw, up_w, xa, xb, wx0, wx1);
      const float* base = tile_logits + ((int64_t)t * Q + q0) * tile_sz;
#pragma unroll
      for (int k = 0; k < ST_QC; ++k) {
        if (q0 + k < Q) {
          const float* p = base + (int64_t)k * tile_sz;
          const float top = p[ya * gw + xa] * wx0 + p[ya * gw + xb] * wx1;
          const float bot = p[yb * gw + xa] * wx0 + p[yb * gw + xb] * wx1;
          acc[k] += top * wy0 + bot * wy1;
        }
      }
      cnt += 1.f;
    }
    if (x < W) {
#pragma unroll
      for (int k = 0; k < ST_QC; ++k)
        if (q0 + k < Q) canvas[(q0 + k) * plane + (int64_t)y * W + x] = cnt > 0.f ? acc[k] / cnt : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ src, int C, int h, int w,
                                                              float* __restrict__ dst, int H, int W) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  int ya, yb, xa, xb; float wy0, wy1, wx0, wx1;
  bilinear_tap(y, h, H, ya, yb, wy0, wy1);
  bilinear_tap(x, w, W, xa, xb, wx0, wx1);
  for (int c = 0; c < C; ++c) {
    const float* p = src + (int64_t)c * h * w;
    const float top = p[ya * w + xa] * wx0 + p[ya * w + xb] * wx1;
    const float bot = p[yb * w + xa] * wx0 + p[yb * w + xb] * wx1;
    dst[((int64_t)c * H + y) * W + x] = top * wy0 + bot * wy1;
  }
}

// ---- postprocess ----------------------------------------------------------------------------------------------------
constexpr int PP_MAX_Q = 64;
// QMAX is a compile-time bound on Q so that v[] lives in registers (a runtime-sized v[64] goes to scratch memory)
template <int QMAX>
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ logits, const int32_t* __restrict__ query_idx,
                                                          int Q, int K, int64_t HW, float logit_scale, float prob_thd, int bg_idx,
                                                          float* __restrict__ probs, int64_t* __restrict__ labels) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  float v[QMAX];
  float mx = -INFINITY;
#pragma unroll
  for (int q = 0; q < QMAX; ++q)
    if (q < Q) { v[q] = logits[q * HW + i] * logit_scale; mx = fmaxf(mx, v[q]); }
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < QMAX; ++q)
    if (q < Q) { v[q] = expf(v[q] - mx); sum += v[q]; }
  float best = -INFINITY; int arg = 0;
  for (int c = 0; c < K; ++c) {
    float pc;
    if (K == Q) {
      pc = 0.f;
#pragma unroll
      for (int q = 0; q < QMAX; ++q) if (q == c) pc = v[q] / sum;       // static indexing keeps v[] in registers
    } else {
      // (probabilities * one_hot).max over queries (segmentor.py:484-486): zeros take part in the max
      bool any_other = false;
      float m = -INFINITY;
#pragma unroll
      for (int q = 0; q < QMAX; ++q)
        if (q < Q) { if (query_idx[q] == c) m = fmaxf(m, v[q] / sum); else any_other = true; }
      pc = any_other ? fmaxf(m, 0.f) : m;
    }
    if (probs) probs[c * HW + i] = pc;
    if (pc > best) { best = pc; arg = c; }                                // first maximum wins (torch argmax)
  }
  if (best < prob_thd) arg = bg_idx;
  labels[i] = arg;
}

// ---- label / confidence images (segmentor.py:501-531, 580-608) ---------------------------------------------------------
// mask  = palette[clip(label, 0, K-1)]                                (_colorize_mask)
// heat  = (g, 0, 255 - g), g = uint8(clip(nan_to_num(max_k probs), 0, 1) * 255)   (_to_colormap, the branch without OpenCV)
__global__ __launch_bounds__(256) void render_maps_kernel(const int64_t* __restrict__ labels, const float* __restrict__ probs,
                                                          const uint8_t* __restrict__ palette, int K, int64_t HW,
                                                          uint8_t* __restrict__ mask_rgb, uint8_t* __restrict__ heat_rgb) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  if (mask_rgb) {
    int64_t l = labels[i];
    l = l < 0 ? 0 : (l > K - 1 ? K - 1 : l);
    mask_rgb[3 * i + 0] = palette[3 * l + 0]; mask_rgb[3 * i + 1] = palette[3 * l + 1]; mask_rgb[3 * i + 2] = palette[3 * l + 2];
  }
  if (heat_rgb) {
    float c = -INFINITY;
    for (int k = 0; k < K; ++k) { const float v = probs[(int64_t)k * HW + i]; c = (v > c || v != v) ? v : c; }   // torch.max propagates NaN
    if (c != c) c = 0.f;
    c = fminf(fmaxf(c, 0.f), 1.f);
    const uint8_t g = (uint8_t)(c * 255.0f);
    heat_rgb[3 * i + 0] = g; heat_rgb[3 * i + 1] = 0; heat_rgb[3 * i + 2] = (uint8_t)(255 - g);
  }
}

}  // namespace sg

using namespace sg;

extern "C" int sg_render_maps(const int64_t* labels, const float* probs, const uint8_t* palette, int K, int H, int W, uint8_t* mask_rgb,
                              uint8_t* heat_rgb, sg_stream s) {
  SG_REQUIRE(K > 0 && H > 0 && W > 0, "sg_render_maps: bad shape");
  SG_REQUIRE(!mask_rgb || (labels && palette), "sg_render_maps: the mask needs labels and a palette");
  SG_REQUIRE(!heat_rgb || probs, "sg_render_maps: the heat map needs the class probabilities");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(render_maps_kernel, dim3((unsigned)cdiv(HW, 256)), dim3(256), 0, as_stream(s), labels, probs, palette, K, HW, mask_rgb, heat_rgb);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

namespace sg {
// Large-n form without the global debias (per-pixel logits behind the upsampler): logits = (x / |x|) . T^T is a [n, E] x [E, 16] GEMM.
// f32-grade on the f16 matrix pipe: T as two f16 planes in LDS, the f32 rows split into hi + lo on the fly, three v_mfma_f32_16x16x32_f16
// per product into one f32 accumulator (the same scheme as SG_PREC_F16X2's GEMMs), |x|^2 in f32 from the loaded values.  A wave owns 64
// consecutive pixels per round; a lane ends with 4 consecutive pixels of one query.  Bound by the read of x (the lane-per-token kernel
// above runs this shape at 1.5 TB/s on the vector pipe: 5.9 ms per 8 tiles of 592 x 592 at E = 768).
constexpr int CLM_ROUNDS = 4;
__global__ __launch_bounds__(256, 2) void cosine_logits_mfma_kernel(const float* __restrict__ tokens, const float* __restrict__ cls,
                                                                    const float* __restrict__ text, int n, int E, int Q, float lambda,
                                                                    float* __restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) char clm_sm[];
  const int ldt = E + 8;
  uint16_t* sTh = reinterpret_cast<uint16_t*>(clm_sm);                   // [16][E + 8] f16
  uint16_t* sTl = sTh + 16 * ldt;
  float* sN = reinterpret_cast<float*>(sTl + 16 * ldt);                   // [4 waves][64]
  float* sCL = sN + 4 * 64;                                               // [16] lambda * cls logits of this image
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16 * E; i += 256) {
    const int q = i / E, c = i % E;
    const float v = q < Q ? text[(int64_t)q * E + c] : 0.f;
    const f16_t hi = f2h(v);
    sTh[q * ldt + c] = hi.bits;
    sTl[q * ldt + c] = f2h(v - h2f(hi)).bits;
  }
  if (tid < 16) sCL[tid] = 0.f;
  __syncthreads();
  if (cls != nullptr && lambda != 0.f) {                                   // lambda * (cls / |cls|) . T[q]   (segmentor.py:310-311, 379), f32
    const float* cr = cls + (int64_t)b * E;
    float ss = 0.f;
    for (int i = lane; i < E; i += 64) ss += cr[i] * cr[i];
    const float inv = 1.0f / sqrtf(wave_sum(ss));
    for (int q = wave; q < Q; q += 4) {
      float d = 0.f;
      for (int i = lane; i < E; i += 64) d += cr[i] * text[(int64_t)q * E + i];
      d = wave_sum(d);
      if (lane == 0) sCL[q] = lambda * d * inv;
    }
    __syncthreads();
  }
  const int r = lane & 15, g = lane >> 4;
  float* myN = sN + wave * 64;
  const float* xb = tokens + (int64_t)b * n * E;
  float* lb = logits + (int64_t)b * Q * n;
  for (int round = 0; round < CLM_ROUNDS; ++round) {
    const int pix0 = ((blockIdx.x * CLM_ROUNDS + round) * 4 + wave) * 64;
    if (pix0 >= n) return;                                                // wave-uniform; no workgroup barrier below
    const float* xr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int pr = pix0 + t * 16 + r;
      pr = pr < n ? pr : n - 1;
      xr[t] = xb + (int64_t)pr * E + 8 * g;
    }
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float nx[4] = {0.f, 0.f, 0.f, 0.f};
    float4 cur[4][2], nxt[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) { cur[t][0] = *reinterpret_cast<const float4*>(xr[t]); cur[t][1] = *reinterpret_cast<const float4*>(xr[t] + 4); }
    const int nk = E / 32;
    for (int ks = 0; ks < nk; ++ks) {
      const int kn = ks + 1 < nk ? ks + 1 : ks;
#pragma unroll
      for (int t = 0; t < 4; ++t) { nxt[t][0] = *reinterpret_cast<const float4*>(xr[t] + 32 * kn); nxt[t][1] = *reinterpret_cast<const float4*>(xr[t] + 32 * kn + 4); }
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sTh + r * ldt + 32 * ks + 8 * g);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(sTl + r * ldt + 32 * ks + 8 * g);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float v[8] = {cur[t][0].x, cur[t][0].y, cur[t][0].z, cur[t][0].w, cur[t][1].x, cur[t][1].y, cur[t][1].z, cur[t][1].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) nx[t] = __builtin_fmaf(v[e], v[e], nx[t]);
        uint4 hi, lo;
        split_h2x8(v, hi, lo);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, hi), al = __builtin_bit_cast(bf16x8, lo);
        acc[t] = mfma_16x16x32<true>(ah, bh, acc[t]);                      // D[pixel i][query j]: lane = j + 16 (i / 4), 4 consecutive pixels
        acc[t] = mfma_16x16x32<true>(al, bh, acc[t]);
        acc[t] = mfma_16x16x32<true>(ah, bl, acc[t]);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) { cur[t][0] = nxt[t][0]; cur[t][1] = nxt[t][1]; }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { nx[t] += __shfl_xor(nx[t], 16, 64); nx[t] += __shfl_xor(nx[t], 32, 64); }
    if (g == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) myN[t * 16 + r] = 1.0f / fmaxf(sqrtf(nx[t]), 1e-12f);   // F.normalize's eps
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (r < Q) {
      const float cl = sCL[r];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int pix = pix0 + t * 16 + 4 * g;
        if (pix >= n) continue;
        const float4 inv = *reinterpret_cast<const float4*>(myN + t * 16 + 4 * g);
        const float o[4] = {acc[t][0] * inv.x + cl, acc[t][1] * inv.y + cl, acc[t][2] * inv.z + cl, acc[t][3] * inv.w + cl};
        float* dst = lb + (int64_t)r * n + pix;
        if (pix + 3 < n && (n & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else {
          for (int e = 0; e < 4; ++e) if (pix + e < n) dst[e] = o[e];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}
}  // namespace sg

extern "C" int sg_cosine_logits(const float* tokens, const float* cls, const float* text, int B, int n, int E, int Q,
                                float global_debias_factor, float cls_token_lambda, float* logits, sg_stream s) {
  SG_REQUIRE(tokens && text && logits, "sg_cosine_logits: null pointer");
  SG_REQUIRE(B > 0 && n > 0 && E > 0 && Q > 0 && B < 65536, "sg_cosine_logits: bad shape B=%d n=%d E=%d Q=%d", B, n, E, Q);
  SG_REQUIRE(cls || (global_debias_factor == 0.f && cls_token_lambda == 0.f), "sg_cosine_logits: cls required for debias / lambda");
  const size_t lds = ((size_t)Q * E + E + Q) * sizeof(float);
  SG_REQUIRE(lds <= 160 * 1024, "sg_cosine_logits: Q*E=%d exceeds LDS", Q * E);
  SG_REQUIRE(E <= 64 * CL_MAXV && E % 4 == 0, "sg_cosine_logits: E=%d must be a multiple of 4 and <= %d", E, 64 * CL_MAXV);
  SG_REQUIRE((((uintptr_t)tokens) & 15) == 0, "sg_cosine_logits: tokens must be 16-byte aligned");
  const int nv = (E + 63) / 64;
  auto kern = nv <= 8 ? cosine_logits_kernel<8> : (nv <= 12 ? cosine_logits_kernel<12> : (nv <= 16 ? cosine_logits_kernel<16> : cosine_logits_kernel<CL_MAXV>));
  if (lds > 48 * 1024) SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(n, CL_TPB), (unsigned)B), dim3(256), lds, as_stream(s), tokens, cls, text,
                     n, E, Q, global_debias_factor, cls_token_lambda, logits);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// The per-pixel logits of the EXACT tower mode (SG_PREC_F16X2): the same contract as sg_cosine_logits without the global debias, on the f16 matrix
// pipe with every operand as two f16 planes (f32-grade; like everything in that mode, magnitudes beyond +-131 008 saturate -- sg_cosine_logits
// itself stays plain f32 arithmetic with no such bound).  Shapes the matrix-pipe form does not take (Q > 16, E % 32 != 0, small n) go to sg_cosine_logits.
extern "C" int sg_cosine_logits_two_plane(const float* tokens, const float* cls, const float* text, int B, int n, int E, int Q,
                                          float cls_token_lambda, float* logits, sg_stream s) {
  SG_REQUIRE(tokens && text && logits, "sg_cosine_logits_two_plane: null pointer");
  SG_REQUIRE(B > 0 && n > 0 && E > 0 && Q > 0 && B < 65536, "sg_cosine_logits_two_plane: bad shape B=%d n=%d E=%d Q=%d", B, n, E, Q);
  SG_REQUIRE(cls || cls_token_lambda == 0.f, "sg_cosine_logits_two_plane: cls required for lambda");
  const size_t ldsm = (size_t)2 * 16 * (E + 8) * 2 + (4 * 64 + 16) * sizeof(float);
  if (!(Q <= 16 && E % 32 == 0 && n >= 4096 && (((uintptr_t)logits) & 15) == 0 && (((uintptr_t)tokens) & 15) == 0 && ldsm <= 160 * 1024))
    return sg_cosine_logits(tokens, cls, text, B, n, E, Q, 0.f, cls_token_lambda, logits, s);
  if (ldsm > 48 * 1024) SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sg::cosine_logits_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm));
  hipLaunchKernelGGL(sg::cosine_logits_mfma_kernel, dim3((unsigned)cdiv(n, 4 * 64 * sg::CLM_ROUNDS), (unsigned)B), dim3(256), ldsm, as_stream(s), tokens, cls, text,
                     n, E, Q, cls_token_lambda, logits);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_stitch(const float* tile_logits, const int32_t* windows, int T, int Q, int gh, int gw, int up_h, int up_w,
                         int pad_t, int pad_l, int H, int W, float* canvas, sg_stream s) {
  SG_REQUIRE(tile_logits && windows && canvas, "sg_stitch: null pointer");
  SG_REQUIRE(T > 0 && Q > 0 && gh > 0 && gw > 0 && H > 0 && W > 0, "sg_stitch: bad shape");
  SG_REQUIRE(cdiv(H, 4) < 65536, "sg_stitch: canvas too tall for one launch");
  hipLaunchKernelGGL(stitch_kernel, dim3((unsigned)cdiv(W, 64), (unsigned)cdiv(H, 4)), dim3(256), 0, as_stream(s), tile_logits, windows,
                     T, Q, gh, gw, up_h, up_w, pad_t, pad_l, H, W, canvas);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_resize_bilinear(const float* src, int C, int h, int w, float* dst, int H, int W, sg_stream s) {
  SG_REQUIRE(src && dst && C > 0 && h > 0 && w > 0 && H > 0 && W > 0, "sg_resize_bilinear: bad arguments");
  SG_REQUIRE(cdiv(H, 4) < 65536, "sg_resize_bilinear: too tall");
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)cdiv(W, 64), (unsigned)cdiv(H, 4)), dim3(256), 0, as_stream(s), src, C, h, w, dst, H, W);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_postprocess(const float* logits, const int32_t* query_idx, int Q, int K, int H, int W, float logit_scale,
                              float prob_thd, int bg_idx, float* probs, int64_t* labels, sg_stream s) {
  SG_REQUIRE(logits && query_idx && labels, "sg_postprocess: null pointer");
  SG_REQUIRE(Q > 0 && Q <= PP_MAX_Q && K > 0 && K <= Q, "sg_postprocess: Q=%d K=%d unsupported (Q <= %d)", Q, K, PP_MAX_Q);
  const int64_t HW = (int64_t)H * W;
  if (Q <= 16)
    hipLaunchKernelGGL(postprocess_kernel<16>, dim3((unsigned)cdiv(HW, 256)), dim3(256), 0, as_stream(s), logits, query_idx, Q, K, HW,
                       logit_scale, prob_thd, bg_idx, probs, labels);
  else
    hipLaunchKernelGGL(postprocess_kernel<PP_MAX_Q>, dim3((unsigned)cdiv(HW, 256)), dim3(256), 0, as_stream(s), logits, query_idx, Q, K, HW,
                       logit_scale, prob_thd, bg_idx, probs, labels);
  SG_LAUNCH_CHECK();
  return SG_OK;
}
