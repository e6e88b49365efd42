// SimFeatUp joint bilateral upsampler (reference simfeatup_dev/upsamplers.py:202-325) on gfx950.
//
// Data layout: features are pixel-major / channels-last ([B, h*w, C] f32) end to end -- the patch tokens of
// the ViT already are -- so every tap of the adaptive convolution reads a contiguous channel vector and
// the final 1x1 conv and the cosine logits are plain row-major GEMM / row ops.
// One 2x stage (JBULearnedRange.forward, :253-275):
//   guidance pool (adaptive_avg_pool2d)                       jbu_pool_kernel
//   range_proj: conv1x1(3->32) . GELU . conv1x1(32->32)       jbu_range_proj_kernel
//   range kernel: softmax_p(temp * <proj[window p], proj[centre]>) over the reflect-padded d x d window,
//     * spatial gaussian, / sum.clamp(1e-7)                   jbu_kernel_tiled_kernel (8x8 pixel block, window in LDS, taps on lanes)
//   fixup: K += 0.1 * conv1x1(GELU(conv1x1([K, guidance])))   two GEMMs over [pixels, d^2(+3)]: f32 MFMA (parity) / bf16 MFMA, padded (throughput)
//   hr = bicubic 2x (A=-0.75, align_corners=False)            jbu_bicubic_kernel (f32, or bf16 in throughput mode)
//   adaptive conv with reflect padding                        jbu_adaptive_conv_kernel (parity: LDS-staged window + weights, VALU)
//                                                             jbu_adaptive_conv_mfma_kernel (throughput: windowed GEMM on the matrix cores)
// then  out = x + 0.1 * conv1x1_CxC(x)  (JBUOne/JBUStack.forward :301,325) as one GEMM with a residual epilogue.
#include <string>
#include <vector>
#include "rowops.h"

namespace sg {

constexpr int KEY_DIM = 32;
constexpr int AC_T = 8;                 // side of the pixel block the windowed kernels (range kernel, adaptive conv) work on

__device__ __forceinline__ int reflect_idx(int u, int size) {       // F.pad(mode='reflect'), pad < size
  if (u < 0) u = -u;
  if (u >= size) u = 2 * (size - 1) - u;
  return u;
}

// ---- tile planes: crop + normalise + zero pad a window of the scene -> [T,3,up_h,up_w] f32 (the `img` the reference hands
// to the upsampler, segmentor.py:371) ------------------------------------------------------------------------------------------
__constant__ float j_mean[3] = {122.771f, 116.746f, 104.094f};
__constant__ float j_std[3] = {68.501f, 66.632f, 70.323f};
__global__ void extract_tiles_kernel(sg_tile_batch t, int up_h, int up_w, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)3 * up_h * up_w;
  if (i >= per * t.n_tiles) return;
  const int tile = (int)(i / per);
  const int rem = (int)(i % per);
  const int c = rem / (up_h * up_w), y = (rem / up_w) % up_h, x = rem % up_w;
  const int ty = y - t.pad_t, tx = x - t.pad_l;
  float v = 0.f;
  if (ty >= 0 && ty < t.tile_h && tx >= 0 && tx < t.tile_w) {
    const int sy = t.windows[tile * 4 + 0] + ty, sx = t.windows[tile * 4 + 2] + tx;
    const int64_t off = t.scene_index ? (int64_t)t.scene_index[tile] * t.scene_stride : 0;
    if (t.format == SG_IMG_F32_NCHW) v = reinterpret_cast<const float*>(t.scene)[off + ((int64_t)c * t.scene_h + sy) * t.scene_w + sx];
    else v = ((float)reinterpret_cast<const uint8_t*>(t.scene)[off + ((int64_t)sy * t.scene_w + sx) * 3 + c] - j_mean[c]) / j_std[c];
  }
  out[i] = v;
}

// ---- guidance pool: F.adaptive_avg_pool2d([B,3,H,W] -> (oh,ow)), written pixel-major [B, oh*ow, 3] ------------------------
__global__ void jbu_pool_kernel(const float* __restrict__ g, int B, int H, int W, int oh, int ow, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * oh * ow * 3) return;
  const int c = (int)(i % 3);
  const int ox = (int)((i / 3) % ow), oy = (int)((i / 3 / ow) % oh), b = (int)(i / 3 / ow / oh);
  const int y0 = (int)(((int64_t)oy * H) / oh), y1 = (int)((((int64_t)oy + 1) * H + oh - 1) / oh);
  const int x0 = (int)(((int64_t)ox * W) / ow), x1 = (int)((((int64_t)ox + 1) * W + ow - 1) / ow);
  const float* p = g + ((int64_t)b * 3 + c) * H * W;
  float s = 0.f;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) s += p[(int64_t)y * W + x];
  out[i] = s / (float)((y1 - y0) * (x1 - x0));
}

// ---- range_proj (upsamplers.py:213-218): per pixel 3 -> 32 (GELU) -> 32 ---------------------------------------------------
__global__ __launch_bounds__(256) void jbu_range_proj_kernel(const float* __restrict__ gs, int64_t pixels, const float* __restrict__ w0,
                                                             const float* __restrict__ b0, const float* __restrict__ w3,
                                                             const float* __restrict__ b3, float* __restrict__ proj) {
  __shared__ float sw0[KEY_DIM * 3], sb0[KEY_DIM], sw3[KEY_DIM * KEY_DIM], sb3[KEY_DIM];
  for (int i = threadIdx.x; i < KEY_DIM * 3; i += 256) sw0[i] = w0[i];
  for (int i = threadIdx.x; i < KEY_DIM * KEY_DIM; i += 256) sw3[i] = w3[i];
  if (threadIdx.x < KEY_DIM) { sb0[threadIdx.x] = b0[threadIdx.x]; sb3[threadIdx.x] = b3[threadIdx.x]; }
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= pixels) return;
  const float g0 = gs[p * 3], g1 = gs[p * 3 + 1], g2 = gs[p * 3 + 2];
  float hdn[KEY_DIM];
#pragma unroll
  for (int o = 0; o < KEY_DIM; ++o) hdn[o] = erf_gelu(sw0[o * 3] * g0 + sw0[o * 3 + 1] * g1 + sw0[o * 3 + 2] * g2 + sb0[o]);
#pragma unroll 4
  for (int o = 0; o < KEY_DIM; ++o) {
    float a = sb3[o];
#pragma unroll
    for (int k = 0; k < KEY_DIM; ++k) a += sw3[o * KEY_DIM + k] * hdn[k];
    proj[p * KEY_DIM + o] = a;
  }
}

// ---- range * spatial kernel (upsamplers.py:230-262): the d*d taps of a pixel on the lanes of a wave -----------------------------
// X[pix][0..d2) = normalised combined kernel, X[pix][d2..d2+3) = guidance (the fixup conv's input rows).
// A workgroup owns 8 x 8 pixels and stages the (8+2r)^2 window
// of projected guidance vectors once (41 kB at r = 5) instead of fetching 121 x 128 B per pixel through L1/L2; a wave then walks its
// 16 pixels with the d*d taps on the lanes, reading centre and neighbour vectors from LDS (row stride 36 floats: conflict-free b128).
constexpr int JK_LD = KEY_DIM + 4;
// RT = the window radius at compile time (3: JBUStack, 5: JBUOne; 0 = run time), FAST = the throughput-mode arithmetic and bf16 operand copy.
template <int RT, bool FAST>
__global__ __launch_bounds__(256) void jbu_kernel_tiled_kernel(const float* __restrict__ proj, const float* __restrict__ gs, int H, int W, int r_rt,
                                                               const float* __restrict__ range_temp, const float* __restrict__ sigma,
                                                               float* __restrict__ X, bf16_t* __restrict__ X16, int ldx16) {
  extern __shared__ __attribute__((aligned(16))) float jk_sm[];
  const int r = RT > 0 ? RT : r_rt;
  const int d = 2 * r + 1, d2 = d * d, ldx = d2 + 3, WT = AC_T + 2 * r;
  const int tiles_x = (W + AC_T - 1) / AC_T;
  const int ty0 = (blockIdx.x / tiles_x) * AC_T, tx0 = (blockIdx.x % tiles_x) * AC_T;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t img = (int64_t)b * H * W;
  for (int i0 = 0; i0 < WT * WT * (KEY_DIM / 4); i0 += 256 * 4) {          // four 16-byte pieces per thread in flight per round
    float4 pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int i = i0 + tid + u * 256;
      i = i < WT * WT * (KEY_DIM / 4) ? i : WT * WT * (KEY_DIM / 4) - 1;
      const int pos = i / (KEY_DIM / 4), q = i % (KEY_DIM / 4);
      int sy = ty0 + pos / WT - r, sx = tx0 + pos % WT - r;
      sy = sy > H - 1 + r ? H - 1 + r : sy; sx = sx > W - 1 + r ? W - 1 + r : sx;    // ragged last tile: stay inside the padded image
      sy = reflect_idx(sy, H); sx = reflect_idx(sx, W);
      pv[u] = *reinterpret_cast<const float4*>(proj + (img + (int64_t)sy * W + sx) * KEY_DIM + 4 * q);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + tid + u * 256;
      if (i < WT * WT * (KEY_DIM / 4)) *reinterpret_cast<float4*>(jk_sm + (i / (KEY_DIM / 4)) * JK_LD + 4 * (i % (KEY_DIM / 4))) = pv[u];
    }
  }
  __syncthreads();
  const float temp = fminf(fmaxf(expf(range_temp[0]), 1e-4f), 1e4f);
  const float sg = sigma[0];
  const float step = 2.0f / (float)(d - 1);
  // tap geometry of this lane (two taps per lane), independent of the pixel
  int toff[2]; float sp[2]; bool tv[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int t = lane + 64 * s2;
    tv[s2] = t < d2;
    const int ti = tv[s2] ? t / d : 0, tj = tv[s2] ? t % d : 0;
    toff[s2] = ti * WT + tj;
    const float fi = -1.0f + (float)ti * step, fj = -1.0f + (float)tj * step;
    sp[s2] = tv[s2] ? expf(-(fi * fi + fj * fj) / (2.0f * sg * sg)) : 0.f;
  }
  // Two pixels per round (pxl and pxl + 8: same column, next row pair) so that their dependent chains -- LDS reads, dot products, two
  // cross-lane reductions, exponentials -- interleave; the second reduction carries (sum e, sum e*spatial) together:
  //   K = e*sp / sum(e) / max(sum(e*sp) / sum(e), 1e-7)  -- evaluated exactly in that order (upsamplers.py:257-262).
  for (int pp = 0; pp < 8; ++pp) {
    float val[2][2]; int64_t pixi[2]; bool live[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int pxl = wave * 16 + pp + 8 * e, py = pxl >> 3, px = pxl & 7;
      const int y = ty0 + py, x = tx0 + px;
      live[e] = y < H && x < W;                                // wave-uniform
      pixi[e] = img + (int64_t)(y < H ? y : H - 1) * W + (x < W ? x : W - 1);
      const float* cqp = jk_sm + ((py + r) * WT + (px + r)) * JK_LD;
      float4 cq[KEY_DIM / 4];
#pragma unroll
      for (int k = 0; k < KEY_DIM / 4; ++k) cq[k] = *reinterpret_cast<const float4*>(cqp + 4 * k);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        val[e][s2] = -INFINITY;
        if (tv[s2]) {
          const float* q = jk_sm + (py * WT + px + toff[s2]) * JK_LD;
          float dot = 0.f;
#pragma unroll
          for (int k = 0; k < KEY_DIM / 4; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(q + 4 * k);
            dot += v.x * cq[k].x + v.y * cq[k].y + v.z * cq[k].z + v.w * cq[k].w;
          }
          val[e][s2] = temp * dot;
        }
      }
    }
    float mx[2] = {fmaxf(val[0][0], val[0][1]), fmaxf(val[1][0], val[1][1])};
    mx[0] = wave_max_dpp(mx[0]); mx[1] = wave_max_dpp(mx[1]);   // DPP + v_readlane: no LDS-queue round trips (six ds_bpermute per value before)
    float ex[2][2], s1[2], s2v[2];
    constexpr bool fast = FAST;                                // throughput mode: hardware exp2 / rcp (1 ulp) -- the result is rounded to bf16 anyway
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (fast) {
        ex[e][0] = __builtin_amdgcn_exp2f((val[e][0] - mx[e]) * 1.4426950408889634f);
        ex[e][1] = __builtin_amdgcn_exp2f((val[e][1] - mx[e]) * 1.4426950408889634f);
      } else {
        ex[e][0] = expf(val[e][0] - mx[e]); ex[e][1] = expf(val[e][1] - mx[e]);        // exp(-inf) = 0 for the unused lanes
      }
      s1[e] = ex[e][0] + ex[e][1];
    }
    s1[0] = wave_sum_dpp(s1[0]); s1[1] = wave_sum_dpp(s1[1]);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float inv = fast ? __builtin_amdgcn_rcpf(s1[e]) : 1.0f / s1[e];
      ex[e][0] = ex[e][0] * inv * sp[0]; ex[e][1] = ex[e][1] * inv * sp[1];
      s2v[e] = ex[e][0] + ex[e][1];
    }
    s2v[0] = wave_sum_dpp(s2v[0]); s2v[1] = wave_sum_dpp(s2v[1]);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (!live[e]) continue;
      const float nrm = fmaxf(s2v[e], 1e-7f);
      const int64_t pix = pixi[e];
      float* xr = X + pix * ldx;
      float k0, k1;
      if (fast) { const float rn = __builtin_amdgcn_rcpf(nrm); k0 = ex[e][0] * rn; k1 = ex[e][1] * rn; }
      else { k0 = ex[e][0] / nrm; k1 = ex[e][1] / nrm; }
      if (lane < d2) xr[lane] = k0;
      if (lane + 64 < d2) xr[lane + 64] = k1;
      if (lane < 3) xr[d2 + lane] = gs[pix * 3 + lane];
      if (FAST) {                                              // [taps | guidance | zero padding] as the bf16 A operand of the fixup GEMM
        bf16_t* x16 = X16 + pix * ldx16;
        if (lane < ldx16) x16[lane] = f2bf(lane < d2 ? k0 : (lane < d2 + 3 ? gs[pix * 3 + (lane - d2)] : 0.f));
        const int t2 = lane + 64;
        if (t2 < ldx16) x16[t2] = f2bf(t2 < d2 ? k1 : (t2 < d2 + 3 ? gs[pix * 3 + (t2 - d2)] : 0.f));
      }
    }
  }
}

// Throughput-mode form of the same kernel with the key dot products on the matrix pipe (round 3).  In the form above every lane reads
// its tap's 32-dimensional key vector from LDS for every pixel (24 ds_read_b128 per pixel and wave: the kernel was LDS-bound).  Here the
// window's key vectors are staged as f16 (11 bits against the bf16 the weights end in), and for one row of 8 pixels a wave computes
//     S[pos][pixel] = key[pos] . key[centre(pixel)]      pos = the D x WT window positions the row's taps can touch (198 at r = 5)
// as NT tiles of ONE v_mfma_f32_16x16x32_f16 each (K = KEY_DIM = 32), parks S as [8 pixels][positions] f32 in LDS and then walks the pixels
// with the taps on the lanes as before -- one ds_read_b32 per tap instead of eight ds_read_b128.  Same softmax / normalisation arithmetic,
// same outputs (X f32 rows, X16 bf16 operand rows).
template <int R, bool EXACT = false>
struct JkmCfg {
  static constexpr int D = 2 * R + 1, D2 = D * D, WT = AC_T + 2 * R, NWIN = WT * WT;
  static constexpr int NROWPOS = (D - 1) * WT + D + 7;                 // window positions (relative to the row's first) the taps of a pixel row touch
  static constexpr int NT = (NROWPOS + 15) / 16, SP = NT * 16, LDS_S = SP + 4;
  static constexpr int KLD = (EXACT ? 2 : 1) * KEY_DIM + 8;            // f16 key row stride (80 bytes; EXACT: [32 hi | 32 lo] + 8 = 144 bytes)
  static constexpr size_t LDS = (size_t)NWIN * KLD * 2 + (size_t)4 * 8 * LDS_S * 4;
};
// EXACT (the upsampler under SG_PREC_F16X2): the keys as two f16 planes and three MFMAs per product (f32-grade scores), expf / true division as in
// the parity kernel, outputs = the f32 rows plus the fixup GEMM's operand rows in TWO-PLANE f16 ([8 hi | 8 lo] groups, X16 then points to h2_t rows of
// ldx16 elements) -- the separate pack pass of that mode is gone.
template <int R, bool EXACT = false>
__global__ __launch_bounds__(256, 2) void jbu_kernel_mfma_kernel(const float* __restrict__ proj, const float* __restrict__ gs, int H, int W,
                                                                 const float* __restrict__ range_temp, const float* __restrict__ sigma,
                                                                 float* __restrict__ X, bf16_t* __restrict__ X16, int ldx16, int half_only) {
  using J = JkmCfg<R, EXACT>;
  constexpr int D = J::D, D2 = J::D2, LDX = D2 + 3, WT = J::WT, NWIN = J::NWIN, NT = J::NT, LDS_S = J::LDS_S, KLD = J::KLD;
  static_assert(KEY_DIM == 32, "one MFMA k-step per product");
  extern __shared__ __attribute__((aligned(16))) char jkm_sm[];
  uint16_t* win = reinterpret_cast<uint16_t*>(jkm_sm);                   // [NWIN][KLD] f16
  float* sS = reinterpret_cast<float*>(jkm_sm + (size_t)NWIN * KLD * 2);  // [4 waves][8][LDS_S]
  const int tiles_x = (W + AC_T - 1) / AC_T;
  const int ty0 = (blockIdx.x / tiles_x) * AC_T, tx0 = (blockIdx.x % tiles_x) * AC_T;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t img = (int64_t)b * H * W;
  for (int i0 = 0; i0 < NWIN * (KEY_DIM / 4); i0 += 256 * 4) {           // four 16-byte pieces per thread in flight per round
    float4 pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int i = i0 + tid + u * 256;
      i = i < NWIN * (KEY_DIM / 4) ? i : NWIN * (KEY_DIM / 4) - 1;
      const int pos = i / (KEY_DIM / 4), q = i % (KEY_DIM / 4);
      int sy = ty0 + pos / WT - R, sx = tx0 + pos % WT - R;
      sy = sy > H - 1 + R ? H - 1 + R : sy; sx = sx > W - 1 + R ? W - 1 + R : sx;    // ragged last tile: stay inside the padded image
      sy = reflect_idx(sy, H); sx = reflect_idx(sx, W);
      pv[u] = *reinterpret_cast<const float4*>(proj + (img + (int64_t)sy * W + sx) * KEY_DIM + 4 * q);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + tid + u * 256;
      if (i < NWIN * (KEY_DIM / 4)) {
        uint16_t* wr = win + (i / (KEY_DIM / 4)) * KLD + 4 * (i % (KEY_DIM / 4));
        if constexpr (EXACT) {
          uint2 hv, lv;
          split_h2(pv[u].x, pv[u].y, hv.x, lv.x); split_h2(pv[u].z, pv[u].w, hv.y, lv.y);
          *reinterpret_cast<uint2*>(wr) = hv; *reinterpret_cast<uint2*>(wr + KEY_DIM) = lv;
        } else *reinterpret_cast<uint2*>(wr) = make_uint2(pack_h2(pv[u].x, pv[u].y), pack_h2(pv[u].z, pv[u].w));
      }
    }
  }
  __syncthreads();
  const float temp = fminf(fmaxf(expf(range_temp[0]), 1e-4f), 1e4f);
  const float sg = sigma[0];
  const float step = 2.0f / (float)(D - 1);
  int toff[2]; float sp[2]; bool tv[2];                                   // tap geometry of this lane (two taps per lane), independent of the pixel
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int t = lane + 64 * s2;
    tv[s2] = t < D2;
    const int ti = tv[s2] ? t / D : 0, tj = tv[s2] ? t % D : 0;
    toff[s2] = ti * WT + tj;
    const float fi = -1.0f + (float)ti * step, fj = -1.0f + (float)tj * step;
    sp[s2] = tv[s2] ? expf(-(fi * fi + fj * fj) / (2.0f * sg * sg)) : 0.f;
  }
  float* myS = sS + wave * 8 * LDS_S;
  const int li = lane & 15, lg = lane >> 4;
  for (int round = 0; round < 2; ++round) {
    const int py = wave * 2 + round;
    // ---- S[pos][pixel] for the 8 pixels of row py ----
    {
      const bf16x8 cfrag = *reinterpret_cast<const bf16x8*>(win + ((py + R) * WT + (li & 7) + R) * KLD + 8 * lg);   // second port: row = pixel (8..15 repeat 0..7)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int pos = py * WT + nt * 16 + li;
        pos = pos < NWIN ? pos : NWIN - 1;
        const bf16x8 pfrag = *reinterpret_cast<const bf16x8*>(win + pos * KLD + 8 * lg);                             // first port: row = window position
        f32x4 acc = mfma_16x16x32<true>(pfrag, cfrag, (f32x4){0.f, 0.f, 0.f, 0.f});   // lane: pixel li, positions nt*16 + 4 lg .. + 3
        if constexpr (EXACT) {
          const bf16x8 pl = *reinterpret_cast<const bf16x8*>(win + pos * KLD + KEY_DIM + 8 * lg);
          const bf16x8 cl = *reinterpret_cast<const bf16x8*>(win + ((py + R) * WT + (li & 7) + R) * KLD + KEY_DIM + 8 * lg);
          acc = mfma_16x16x32<true>(pl, cfrag, acc);
          acc = mfma_16x16x32<true>(pfrag, cl, acc);
        }
        if (li < 8) *reinterpret_cast<float4*>(myS + li * LDS_S + nt * 16 + 4 * lg) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                // a wave's LDS instructions execute in order: ordering only
    __builtin_amdgcn_wave_barrier();
    // ---- the pixels of the row, two at a time (px, px + 4), taps on the lanes ----
    for (int pp = 0; pp < 4; ++pp) {
      float val[2][2]; int64_t pixi[2]; bool live[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int px = pp + 4 * e;
        const int y = ty0 + py, x = tx0 + px;
        live[e] = y < H && x < W;                                         // wave-uniform
        pixi[e] = img + (int64_t)(y < H ? y : H - 1) * W + (x < W ? x : W - 1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) val[e][s2] = tv[s2] ? temp * myS[px * LDS_S + px + toff[s2]] : -INFINITY;
      }
      float mx[2] = {fmaxf(val[0][0], val[0][1]), fmaxf(val[1][0], val[1][1])};
      mx[0] = wave_max_dpp(mx[0]); mx[1] = wave_max_dpp(mx[1]);
      float ex[2][2], s1[2], s2v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if constexpr (EXACT) { ex[e][0] = expf(val[e][0] - mx[e]); ex[e][1] = expf(val[e][1] - mx[e]); }      // exp(-inf) = 0 for the unused lanes
        else {
          ex[e][0] = __builtin_amdgcn_exp2f((val[e][0] - mx[e]) * 1.4426950408889634f);
          ex[e][1] = __builtin_amdgcn_exp2f((val[e][1] - mx[e]) * 1.4426950408889634f);
        }
        s1[e] = ex[e][0] + ex[e][1];
      }
      s1[0] = wave_sum_dpp(s1[0]); s1[1] = wave_sum_dpp(s1[1]);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float inv = EXACT ? 1.0f / s1[e] : __builtin_amdgcn_rcpf(s1[e]);
        ex[e][0] = ex[e][0] * inv * sp[0]; ex[e][1] = ex[e][1] * inv * sp[1];
        s2v[e] = ex[e][0] + ex[e][1];
      }
      s2v[0] = wave_sum_dpp(s2v[0]); s2v[1] = wave_sum_dpp(s2v[1]);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (!live[e]) continue;
        const float nrm = fmaxf(s2v[e], 1e-7f);
        float k0, k1;
        if constexpr (EXACT) { k0 = ex[e][0] / nrm; k1 = ex[e][1] / nrm; }
        else { const float rn = __builtin_amdgcn_rcpf(nrm); k0 = ex[e][0] * rn; k1 = ex[e][1] * rn; }
        const int64_t pix = pixi[e];
        // [taps | guidance | zero padding]: the 2-byte A operand of the fixup GEMM.  half_only (the all-f16 fixup chain of the low-res path): f16
        // rows that also serve as that chain's residual, and no f32 rows at all (752 -> 256 bytes written per pixel)
        const float v0 = lane < D2 ? k0 : (lane < D2 + 3 ? gs[pix * 3 + (lane - D2)] : 0.f);
        const int t2 = lane + 64;
        const float v1 = t2 < D2 ? k1 : (t2 < D2 + 3 ? gs[pix * 3 + (t2 - D2)] : 0.f);
        bf16_t* x16 = X16 + pix * ldx16;
        if constexpr (EXACT) {                                            // f32 rows (the fixup residual) + two-plane operand rows
          float* xr = X + pix * LDX;
          if (lane < D2) xr[lane] = k0;
          if (lane + 64 < D2) xr[lane + 64] = k1;
          if (lane < 3) xr[D2 + lane] = gs[pix * 3 + lane];
          h2_t* x2r = reinterpret_cast<h2_t*>(X16) + pix * ldx16;
          if (lane < ldx16) st_elem<h2_t>(x2r, lane, v0);
          if (t2 < ldx16) st_elem<h2_t>(x2r, t2, v1);
        } else if (half_only) {
          if (lane < ldx16) x16[lane] = f2h(v0).bits;
          if (t2 < ldx16) x16[t2] = f2h(v1).bits;
        } else {
          float* xr = X + pix * LDX;
          if (lane < D2) xr[lane] = k0;
          if (lane + 64 < D2) xr[lane + 64] = k1;
          if (lane < 3) xr[D2 + lane] = gs[pix * 3 + lane];
          if (lane < ldx16) x16[lane] = f2bf(v0);
          if (t2 < ldx16) x16[t2] = f2bf(v1);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();                                      // S is rewritten by the next round
  }
}

// ---- bicubic 2x (torch.nn.Upsample(size, mode='bicubic', align_corners=False), A = -0.75) on [B,h,w,C] -> [B,oh,ow,C] ----------
__device__ __forceinline__ float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__device__ __forceinline__ void cubic_taps(int dst, int in, int out, int* idx, float* w) {
  const float A = -0.75f;
  const float src = ((float)in / (float)out) * ((float)dst + 0.5f) - 0.5f;
  const float fl = floorf(src);
  const float t = src - fl;
  const int i0 = (int)fl;
  w[0] = cc2(t + 1.f, A); w[1] = cc1(t, A); w[2] = cc1(1.f - t, A); w[3] = cc2(2.f - t, A);
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int j = i0 - 1 + k; idx[k] = j < 0 ? 0 : (j > in - 1 ? in - 1 : j); }
}
template <typename OutT>      // bf16_t in throughput mode: the matrix-core adaptive convolution consumes bf16 features
__global__ __launch_bounds__(256) void jbu_bicubic_kernel(const float* __restrict__ src, int B, int h, int w, int C, int oh, int ow,
                                                          OutT* __restrict__ dst) {
  const int C4 = C >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * oh * ow * C4) return;
  const int c4 = (int)(i % C4);
  const int ox = (int)((i / C4) % ow), oy = (int)((i / C4 / ow) % oh), b = (int)(i / C4 / ow / oh);
  int iy[4], ix[4]; float wy[4], wx[4];
  cubic_taps(oy, h, oh, iy, wy);
  cubic_taps(ox, w, ow, ix, wx);
  const float* base = src + (int64_t)b * h * w * C + 4 * c4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(base + ((int64_t)iy[a] * w + ix[k]) * C);
      row.x += wx[k] * v.x; row.y += wx[k] * v.y; row.z += wx[k] * v.z; row.w += wx[k] * v.w;
    }
    acc.x += wy[a] * row.x; acc.y += wy[a] * row.y; acc.z += wy[a] * row.z; acc.w += wy[a] * row.w;
  }
  OutT* o = dst + ((int64_t)(b * oh + oy) * ow + ox) * C + 4 * c4;
  if constexpr (sizeof(OutT) == 4) *reinterpret_cast<float4*>(o) = acc;
  else { uint2 q; q.x = pack_bf2(acc.x, acc.y); q.y = pack_bf2(acc.z, acc.w); *reinterpret_cast<uint2*>(o) = q; }
}

// ---- adaptive convolution, channels-last, reflect padding folded into the window staging ---------------------------------------
// Workgroup = 8x8 output pixels x 32 channels.  LDS holds the (8+2r)^2 reflect-indexed window of those 32 channels (row stride
// 36 floats: conflict-free 16-byte reads across a row of pixels) and the 64 pixels' d*d weights (tap-major).
constexpr int AC_CC = 32, AC_LD = 36;
__global__ __launch_bounds__(256) void jbu_adaptive_conv_kernel(const float* __restrict__ hr, const float* __restrict__ Kf, int ldk, int H, int W,
                                                                int C, int r, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int d = 2 * r + 1, d2 = d * d, WT = AC_T + 2 * r;
  float* sWin = sm;                                  // [WT*WT][AC_LD]
  float* sKw = sm + WT * WT * AC_LD;                 // [d2][64]
  const int tiles_x = (W + AC_T - 1) / AC_T;
  const int ty0 = (blockIdx.x / tiles_x) * AC_T, tx0 = (blockIdx.x % tiles_x) * AC_T;
  const int c0 = blockIdx.y * AC_CC, b = blockIdx.z, tid = threadIdx.x;
  const float* hb = hr + (int64_t)b * H * W * C;
  for (int i = tid; i < WT * WT * (AC_CC / 4); i += 256) {
    const int pos = i / (AC_CC / 4), q = i % (AC_CC / 4);
    const int wy = pos / WT, wx = pos % WT;
    int sy = ty0 + wy - r, sx = tx0 + wx - r;
    sy = sy > H - 1 + r ? H - 1 + r : sy; sx = sx > W - 1 + r ? W - 1 + r : sx;      // ragged last tile: stay inside the padded image
    sy = reflect_idx(sy, H); sx = reflect_idx(sx, W);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 + 4 * q < C) v = *reinterpret_cast<const float4*>(hb + ((int64_t)sy * W + sx) * C + c0 + 4 * q);
    *reinterpret_cast<float4*>(sWin + pos * AC_LD + 4 * q) = v;
  }
  for (int i = tid; i < d2 * 64; i += 256) {
    const int pxl = i / d2, t = i % d2;                                              // coalesced along the taps of one pixel
    int y = ty0 + (pxl >> 3), x = tx0 + (pxl & 7);
    y = y < H ? y : H - 1; x = x < W ? x : W - 1;
    sKw[t * 64 + pxl] = Kf[(((int64_t)b * H + y) * W + x) * ldk + t];
  }
  __syncthreads();
  const int pxl = tid & 63, cg = tid >> 6, py = pxl >> 3, px = pxl & 7;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      const float wv = sKw[(i * d + j) * 64 + pxl];
      const float* p = sWin + ((py + i) * WT + (px + j)) * AC_LD + cg * 8;
      const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
      a0.x += wv * v0.x; a0.y += wv * v0.y; a0.z += wv * v0.z; a0.w += wv * v0.w;
      a1.x += wv * v1.x; a1.y += wv * v1.y; a1.z += wv * v1.z; a1.w += wv * v1.w;
    }
  const int y = ty0 + py, x = tx0 + px, c = c0 + cg * 8;
  if (y < H && x < W) {
    float* o = out + (((int64_t)b * H + y) * W + x) * C + c;
    if (c < C) *reinterpret_cast<float4*>(o) = a0;
    if (c + 4 < C) *reinterpret_cast<float4*>(o + 4) = a1;
  }
}

// ---- adaptive convolution on the matrix cores (throughput mode) -----------------------------------------------------------------
// For an 8 x 8 block of output pixels the d*d taps of every pixel fall inside one (8+2r)^2 window, so
//     out[64 px, C] = F[64, WT^2] . Win[WT^2, C]       F[p, (py+i)*WT + (px+j)] = Kf[p, i*d+j], zero elsewhere
// is a dense GEMM (37 % of F is non-zero at r = 5) that v_mfma_f32_16x16x32_bf16 runs ~20x faster than the 121-tap VALU loop.
// LDS: F [64][KP+8] bf16, built once per pixel block (zero fill + scatter of the taps); Win^T [128 ch][KP+8] bf16 per channel chunk
// (window positions contiguous = the MFMA K dimension), restaged for every chunk of 128 channels.  Operands are swapped so that a
// lane owns 4 consecutive channels of one pixel (16-byte stores).  Features and kernel weights are rounded to bf16 (weights are in
// [0,1] and sum to 1): throughput mode only, the f32 parity mode keeps the VALU kernel above.
constexpr int ACM_CC = 128;
__global__ __launch_bounds__(256) void jbu_adaptive_conv_mfma_kernel(const bf16_t* __restrict__ hr, const float* __restrict__ Kf, int ldk, int H,
                                                                     int W, int C, int r, int KP, float* __restrict__ out, bf16_t* __restrict__ out16) {
  extern __shared__ __attribute__((aligned(16))) char acm_sm[];
  const int d = 2 * r + 1, d2 = d * d, WT = AC_T + 2 * r, NPOS = WT * WT, LDK = KP + 8;
  bf16_t* sF = reinterpret_cast<bf16_t*>(acm_sm);            // [64][LDK]
  bf16_t* sW = sF + 64 * LDK;                                // [ACM_CC][LDK]
  const int tiles_x = (W + AC_T - 1) / AC_T;
  // XCD-aware order: workgroups of one XCD (blockIdx.x % 8) walk a contiguous raster range of pixel blocks, so the halo rows /
  // columns shared by neighbouring blocks (5x read amplification at r = 5) are served by that XCD's L2
  const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + seq;
  const int ty0 = (blk / tiles_x) * AC_T, tx0 = (blk % tiles_x) * AC_T;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* hb = hr + (int64_t)b * H * W * C;
  // F: zero, then scatter the taps
  for (int i = tid; i < 64 * LDK / 8; i += 256) reinterpret_cast<uint4*>(sF)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  for (int i = tid; i < d2 * 64; i += 256) {
    const int pxl = i / d2, t = i % d2;                                              // coalesced along the taps of one pixel
    int y = ty0 + (pxl >> 3), x = tx0 + (pxl & 7);
    y = y < H ? y : H - 1; x = x < W ? x : W - 1;
    const int ti = t / d, tj = t % d;
    sF[pxl * LDK + ((pxl >> 3) + ti) * WT + (pxl & 7) + tj] = f2bf(Kf[(((int64_t)b * H + y) * W + x) * ldk + t]);
  }
  for (int c0 = 0; c0 < C; c0 += ACM_CC) {
    __syncthreads();                                         // F complete / previous chunk's MFMAs done with sW
    // Win^T: thread = (8 channels, position pair); two neighbouring positions go out as one 32-bit word per channel
    const int npair = KP / 2;
    for (int i = tid; i < npair * (ACM_CC / 8); i += 256) {
      const int q = i % (ACM_CC / 8), pp = i / (ACM_CC / 8);
      uint4 v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int pos = 2 * pp + e;
        v[e] = make_uint4(0, 0, 0, 0);
        if (pos < NPOS && c0 + 8 * q < C) {
          int sy = ty0 + pos / WT - r, sx = tx0 + pos % WT - r;
          sy = sy > H - 1 + r ? H - 1 + r : sy; sx = sx > W - 1 + r ? W - 1 + r : sx;  // ragged last tile: stay inside the padded image
          sy = reflect_idx(sy, H); sx = reflect_idx(sx, W);
          v[e] = *reinterpret_cast<const uint4*>(hb + ((int64_t)sy * W + sx) * C + c0 + 8 * q);
        }
      }
      uint32_t* dst = reinterpret_cast<uint32_t*>(sW + (8 * q) * LDK + 2 * pp);
      const uint32_t a[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, bq[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) {                       // word w2 holds channels 2 w2 (low half) and 2 w2 + 1 (high half)
        dst[(2 * w2) * (LDK / 2)] = (a[w2] & 0xffffu) | (bq[w2] << 16);
        dst[(2 * w2 + 1) * (LDK / 2)] = (a[w2] >> 16) | (bq[w2] & 0xffff0000u);
      }
    }
    __syncthreads();
    // wave w: channels [32w, 32w+32) of the chunk x all 64 pixels
    f32x4 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < KP; k0 += 32) {
      bf16x8 fa[4], fw[2];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[mi] = *reinterpret_cast<const bf16x8*>(sF + (mi * 16 + (lane & 15)) * LDK + k0 + (lane >> 4) * 8);
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) fw[nj] = *reinterpret_cast<const bf16x8*>(sW + (wave * 32 + nj * 16 + (lane & 15)) * LDK + k0 + (lane >> 4) * 8);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nj], fa[mi], acc[mi][nj], 0, 0, 0);
    }
    // lane: pixel mi*16 + (lane & 15), channels c0 + 32 wave + 16 nj + 4 (lane >> 4) .. +4
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int pxl = mi * 16 + (lane & 15);
      const int y = ty0 + (pxl >> 3), x = tx0 + (pxl & 7);
      if (y < H && x < W) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          const int c = c0 + wave * 32 + nj * 16 + (lane >> 4) * 4;
          if (c < C) {
            const int64_t o = (((int64_t)b * H + y) * W + x) * C + c;
            *reinterpret_cast<float4*>(out + o) = make_float4(acc[mi][nj][0], acc[mi][nj][1], acc[mi][nj][2], acc[mi][nj][3]);
            if (out16) {                                     // last stage: the bf16 A operand of the final 1x1 conv, no separate pack pass
              uint2 q; q.x = pack_bf2(acc[mi][nj][0], acc[mi][nj][1]); q.y = pack_bf2(acc[mi][nj][2], acc[mi][nj][3]);
              *reinterpret_cast<uint2*>(out16 + o) = q;
            }
          }
        }
      }
    }
  }
}


// ---- adaptive convolution on the LOW-RES source (throughput mode, radius 5 / 3) -------------------------------------------------
// hr = bicubic2x(src) is linear in src, so   out[p] = sum_t K[p,t] hr[p+t] = sum_s Keff[p,s] src[s]   with
//     Keff = Wy^T . K . Wx        (Wy / Wx: the 4-tap bicubic rows of the reflect-padded hi-res tap rows / columns of the pixel)
// over a LW x LW low-res window (12 x 12 at r = 5, 10 x 10 at r = 3) instead of the (8+2r)^2 hi-res window: the bicubic kernel, the
// hi-res tensor and half of the GEMM's K dimension disappear (K 352 -> 160 at r = 5).  Per 8 x 8 pixel block:
//   1. Wx / Wy tables (dense [8][D][16] f32) from the same cubic_taps() arithmetic as jbu_bicubic_kernel, then packed to f16 PAIRS along
//      the tap index ([8][D/2][16] dwords); K rows of the 64 pixels -> LDS as f16 rows (pairs along the tap column)
//   2. Keff per pixel on the VALU with v_dot2c_f32_f16 (two products per instruction, f32 accumulation): 4 threads per pixel, thread q owns
//      window columns 3q..3q+2: T = K . Wx (registers), T re-paired along its rows, Keff = Wy^T . T
//      -> F [64 px][KP] bf16, K slot = q * QW + 3 ly + c for window position (ly, 3q + c): one contiguous run per thread, written with
//      8 / 16-byte LDS stores                                    [1.5 k dot2 per pixel, shared by all C channels; f16 operands carry 11 bits
//      against the 8 of the bf16 F they end in]
//      (history: with f32 operands and the arithmetic SLP-packed into v_pk_fma_f32 this kernel was not reproducible from run to run; the
//      unit is built with -fno-slp-vectorize, see build.py and DESIGN.md section 4 'JBU reproducibility'; tests/test_gpu_repro.py)
//   3. per 128-channel chunk: window [K slot][128 ch] bf16 copied as it lies in HBM (16-byte pieces, no transposition); the MFMA operand
//      (8 consecutive window positions of one channel per lane) is fetched with two ds_read_b64_tr_b16 (hardware transpose);
//      out[64, 128] = F . Win on v_mfma_f32_16x16x32_bf16, operands swapped so a lane owns 4 consecutive channels of one pixel.
// 65 KB of LDS: two blocks per CU overlap each other's staging and MFMA phases.
template <int R> struct LowCfg {
  static constexpr int D = 2 * R + 1, D2 = D * D;
  static constexpr int DPH = (D + 1) / 2, DP = 2 * DPH; // tap pairs per K row / table column (the odd tap is paired with a zero)
  static constexpr int LW = R == 5 ? 12 : 10;          // low-res window side
  static constexpr int LWP = 16;                        // table row stride: x rows hold thread q's 3 columns at [4q, 4q+3) (16-byte aligned reads), y rows are plain
  static constexpr int OFF = R == 5 ? 4 : 3;            // window origin = block origin / 2 - OFF
  static constexpr int QW = R == 5 ? 36 : 32;          // K slots of one column-owner thread: slot q * QW + 3 ly + c <-> window (ly, 3q + c)
  static constexpr int NPOS = 4 * QW;
  static constexpr int KP = (NPOS + 31) / 32 * 32;
  static constexpr int LDK = KP + 8;                    // F row stride (bf16)
  static constexpr int LDW = ACM_CC + 16;               // window row stride (bf16): [pos][ch]; 288 B keeps the 4-row transposed reads on distinct banks
  static constexpr size_t F_BYTES = (size_t)64 * LDK * 2;
  static constexpr size_t W_BYTES = (size_t)KP * LDW * 2;
  static constexpr size_t KH_BYTES = (size_t)64 * D * DP * 2;             // K rows, f16
  static constexpr size_t TF_BYTES = (size_t)2 * 8 * D * LWP * 4;         // f32 tables (build scratch)
  static constexpr size_t TP_BYTES = (size_t)2 * 8 * DPH * LWP * 4;       // pair-packed tables
  static constexpr size_t K_BYTES = KH_BYTES + TF_BYTES + TP_BYTES;
  static constexpr size_t LDS = F_BYTES + (W_BYTES > K_BYTES ? W_BYTES : K_BYTES);
  static_assert(KH_BYTES % 16 == 0 && TF_BYTES % 16 == 0, "LDS sub-buffers stay 16-byte aligned");
};

__device__ __forceinline__ uint32_t pack_h2_bounded(float lo, float hi) {   // operands known to be far inside the f16 range: no clamps
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ float dot2_f16(uint32_t a, uint32_t b, float acc) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), acc, false);
}

template <int R>
__global__ __launch_bounds__(256, 2) void jbu_conv_lowres_kernel(const bf16_t* __restrict__ src, const float* __restrict__ Kf, int ldk, int h, int w,
                                                                 int C, float* __restrict__ out, bf16_t* __restrict__ out16, int kf_half) {
  using L = LowCfg<R>;
  constexpr int D = L::D, D2 = L::D2, DPH = L::DPH, DP = L::DP, LW = L::LW, LWP = L::LWP, KP = L::KP, LDK = L::LDK, LDW = L::LDW;
  extern __shared__ __attribute__((aligned(16))) char lc_sm[];
  bf16_t* sF = reinterpret_cast<bf16_t*>(lc_sm);                          // [64][LDK]
  bf16_t* sW = reinterpret_cast<bf16_t*>(lc_sm + L::F_BYTES);            // [KP][LDW]            (chunk loop)
  uint16_t* sKh = reinterpret_cast<uint16_t*>(lc_sm + L::F_BYTES);       // [64][D][DP] f16      (Keff build; aliases sW)
  float* sTf = reinterpret_cast<float*>(lc_sm + L::F_BYTES + L::KH_BYTES);              // [2][8][D][LWP] f32: x tables, y tables
  uint32_t* sTp = reinterpret_cast<uint32_t*>(lc_sm + L::F_BYTES + L::KH_BYTES + L::TF_BYTES);   // [2][8][DPH][LWP] f16 pairs (taps 2jp, 2jp+1)
  const int H = 2 * h, W = 2 * w;
  const int tiles_x = (W + AC_T - 1) / AC_T;
  const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + seq;
  const int ty0 = (blk / tiles_x) * AC_T, tx0 = (blk % tiles_x) * AC_T;
  const int ly0 = ty0 / 2 - L::OFF, lx0 = tx0 / 2 - L::OFF;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- 1. tables + the block's kernel rows ----
  if (tid < 2 * 8 * D) {
    const bool isy = tid >= 8 * D;
    const int e = isy ? tid - 8 * D : tid, pl = e / D, t = e % D;
    const int size = isy ? H : W, lo_size = isy ? h : w, org = isy ? ty0 : tx0, lorg = isy ? ly0 : lx0;
    int u = org + pl + t - R;
    u = u > size - 1 + R ? size - 1 + R : u;                               // ragged last block: stay inside the padded image
    u = reflect_idx(u, size);
    int idx[4]; float wt[4];
    cubic_taps(u, lo_size, size, idx, wt);
    float* row = sTf + (isy ? 8 * D * LWP : 0) + (pl * D + t) * LWP;
#pragma unroll
    for (int c = 0; c < LWP; ++c) row[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int c = idx[k] - lorg;
      c = c < 0 ? 0 : (c > LW - 1 ? LW - 1 : c);                           // by construction already inside the window
      row[isy ? c : c + c / 3] += wt[k];                                   // x rows: column 3q + j at slot 4q + j
    }
  }
  {
    constexpr int NK = (D2 * 64 + 255) / 256;                              // loads per thread: issued back to back, stored afterwards
    float kv[NK];
    auto kidx = [&](int u) -> int64_t {
      int i = tid + u * 256;
      i = i < D2 * 64 ? i : D2 * 64 - 1;
      const int pxl = i / D2, t = i % D2;                                  // coalesced along the taps of one pixel
      int y = ty0 + (pxl >> 3), x = tx0 + (pxl & 7);
      y = y < H ? y : H - 1; x = x < W ? x : W - 1;
      return (((int64_t)b * H + y) * W + x) * ldk + t;
    };
    if (kf_half) {                                                         // f16 rows out of the all-f16 fixup chain (one branch around the whole batch of loads)
      uint16_t kh[NK];
#pragma unroll
      for (int u = 0; u < NK; ++u) kh[u] = reinterpret_cast<const uint16_t*>(Kf)[kidx(u)];
#pragma unroll
      for (int u = 0; u < NK; ++u) kv[u] = h2f(f16_t{kh[u]});
    } else {
#pragma unroll
      for (int u = 0; u < NK; ++u) kv[u] = Kf[kidx(u)];
    }
#pragma unroll
    for (int u = 0; u < NK; ++u) {
      const int i = tid + u * 256;
      const int pxl = i / D2, t = i % D2;
      if (i < D2 * 64) sKh[(pxl * D + t / D) * DP + t % D] = f2h(kv[u]).bits;    // K values are O(1) softmax-like weights
    }
    if constexpr (DP > D)                                                   // the zero partner of the odd tap
      for (int i = tid; i < 64 * D; i += 256) sKh[i * DP + D] = 0;
  }
  __syncthreads();
  for (int i = tid; i < 2 * 8 * DPH * LWP; i += 256) {                      // tables -> f16 pairs along the tap index
    const int col = i % LWP, jp = (i / LWP) % DPH, tp = i / (LWP * DPH);   // tp = table * 8 + pixel
    const float* r0 = sTf + (tp * D + 2 * jp) * LWP + col;
    sTp[i] = pack_h2_bounded(r0[0], 2 * jp + 1 < D ? r0[LWP] : 0.f);
  }
  __syncthreads();
  // ---- 2. Keff = Wy^T . K . Wx for pixel p, window columns 3q..3q+2 ----
  {
    const int p = tid >> 2, q = tid & 3, py = p >> 3, px = p & 7;
    const uint32_t* kp = reinterpret_cast<const uint32_t*>(sKh) + p * D * DPH;
    const uint32_t* wx = sTp + px * DPH * LWP + 4 * q;
    const uint32_t* wy = sTp + (8 + py) * DPH * LWP;
    float T[DP][3];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int jp = 0; jp < DPH; ++jp) {
        const uint32_t k2 = kp[i * DPH + jp];
        t0 = dot2_f16(k2, wx[jp * LWP], t0); t1 = dot2_f16(k2, wx[jp * LWP + 1], t1); t2 = dot2_f16(k2, wx[jp * LWP + 2], t2);
      }
      T[i][0] = t0; T[i][1] = t1; T[i][2] = t2;
    }
    if constexpr (DP > D) { T[D][0] = 0.f; T[D][1] = 0.f; T[D][2] = 0.f; }
    uint32_t T2[DPH][3];                                                   // rows (2 ip, 2 ip + 1) of T as f16 pairs
#pragma unroll
    for (int ip = 0; ip < DPH; ++ip)
#pragma unroll
      for (int c = 0; c < 3; ++c) T2[ip][c] = pack_h2_bounded(T[2 * ip][c], T[2 * ip + 1][c]);
    float e[LW][3];
#pragma unroll
    for (int ly = 0; ly < LW; ++ly) { e[ly][0] = 0.f; e[ly][1] = 0.f; e[ly][2] = 0.f; }
#pragma unroll
    for (int ip = 0; ip < DPH; ++ip)
#pragma unroll
      for (int ly = 0; ly < LW; ++ly) {
        const uint32_t wv = wy[ip * LWP + ly];
        e[ly][0] = dot2_f16(wv, T2[ip][0], e[ly][0]); e[ly][1] = dot2_f16(wv, T2[ip][1], e[ly][1]); e[ly][2] = dot2_f16(wv, T2[ip][2], e[ly][2]);
      }
    // F row of the pixel, K index = q * QW + 3 ly + c: every thread owns one contiguous, 8-byte aligned run and writes it with wide
    // stores (window columns >= LW of the last thread carry zero table weights, so those slots are exact zeros)
    constexpr int QW = L::QW;
    uint32_t pk[QW / 2];
#pragma unroll
    for (int j = 0; j < QW / 2; ++j) {
      const int i0 = 2 * j, i1 = 2 * j + 1;
      const float v0 = i0 < 3 * LW ? e[i0 / 3 < LW ? i0 / 3 : 0][i0 % 3] : 0.f;
      const float v1 = i1 < 3 * LW ? e[i1 / 3 < LW ? i1 / 3 : 0][i1 % 3] : 0.f;
      pk[j] = pack_bf2(v0, v1);
    }
    bf16_t* fr = sF + p * LDK + q * QW;                                    // F does not alias sK / the tables
    if constexpr (QW % 8 == 0) {
#pragma unroll
      for (int j = 0; j < QW / 8; ++j) *reinterpret_cast<uint4*>(fr + 8 * j) = make_uint4(pk[4 * j], pk[4 * j + 1], pk[4 * j + 2], pk[4 * j + 3]);
    } else {
#pragma unroll
      for (int j = 0; j < QW / 4; ++j) *reinterpret_cast<uint2*>(fr + 4 * j) = make_uint2(pk[2 * j], pk[2 * j + 1]);
    }
    if constexpr (KP > L::NPOS) {                                          // K padding: (KP - NPOS) / 4 slots per thread
      static_assert((KP - L::NPOS) == 16, "padding is written as one 8-byte piece per thread");
      *reinterpret_cast<uint2*>(sF + p * LDK + L::NPOS + 4 * q) = make_uint2(0u, 0u);
    }
  }
  const bf16_t* sb = src + (int64_t)b * h * w * C;
  for (int c0 = 0; c0 < C; c0 += ACM_CC) {
    __syncthreads();                                                       // F complete / Keff build done with sK / previous chunk's MFMAs done with sW
    // ---- 3a. window [pos][128 ch] as it lies in HBM: 16 x 16-byte pieces per position ----
    {
      constexpr int NS = KP * (ACM_CC / 8) / 256;                          // 16-byte pieces per thread: all in flight before the first LDS store
      static_assert(KP * (ACM_CC / 8) % 256 == 0, "window pieces must divide over the threads");
      uint4 wv[NS];
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const int i = tid + u * 256, pos = i / (ACM_CC / 8), qc = i % (ACM_CC / 8);
        const int pc = pos < L::NPOS ? pos : L::NPOS - 1;                  // K padding rows: F is zero there, any finite value will do
        const int qd = pc / L::QW, rem = pc % L::QW;                       // K slot -> window (ly, lx); unused slots clamp onto the window
        const int wy_ = rem / 3 < LW ? rem / 3 : LW - 1, wx_ = 3 * qd + rem % 3 < LW ? 3 * qd + rem % 3 : LW - 1;
        int sy = ly0 + wy_, sx = lx0 + wx_;
        sy = sy < 0 ? 0 : (sy > h - 1 ? h - 1 : sy); sx = sx < 0 ? 0 : (sx > w - 1 ? w - 1 : sx);   // weight 0 outside the image too
        int cc = c0 + 8 * qc; cc = cc + 8 <= C ? cc : C - 8;               // ragged last chunk: a valid duplicate, never stored
        wv[u] = *reinterpret_cast<const uint4*>(sb + ((int64_t)sy * w + sx) * C + cc);
      }
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const int i = tid + u * 256, pos = i / (ACM_CC / 8), qc = i % (ACM_CC / 8);
        *reinterpret_cast<uint4*>(sW + pos * LDW + 8 * qc) = wv[u];
      }
    }
    __syncthreads();
    // ---- 3b. wave: channels [32 wave, +32) x 64 pixels ----
    f32x4 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((ext_vector_type(4))) short s4_t;
    typedef __attribute__((address_space(3))) s4_t* lds_s4_t;
    typedef __attribute__((ext_vector_type(8))) short s8_t;
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;         // transposed read: lane 4q+p of a 16-lane group addresses row q, cols 4p..4p+3
#pragma unroll 1
    for (int k0 = 0; k0 < KP; k0 += 32) {
      bf16x8 fa[4], fw[2];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[mi] = *reinterpret_cast<const bf16x8*>(sF + (mi * 16 + (lane & 15)) * LDK + k0 + g * 8);
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) {
        const bf16_t* p0 = sW + (k0 + 8 * g + qq) * LDW + wave * 32 + nj * 16 + 4 * pp;
        const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(p0));
        const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(p0 + 4 * LDW));
        const s8_t both = (s8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        fw[nj] = __builtin_bit_cast(bf16x8, both);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nj], fa[mi], acc[mi][nj], 0, 0, 0);
    }
    // lane: pixel mi*16 + (lane & 15), channels c0 + 32 wave + 16 nj + 4 (lane >> 4) .. +4
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int pxl = mi * 16 + (lane & 15);
      const int y = ty0 + (pxl >> 3), x = tx0 + (pxl & 7);
      if (y < H && x < W) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          const int c = c0 + wave * 32 + nj * 16 + (lane >> 4) * 4;
          if (c < C) {
            const int64_t o = (((int64_t)b * H + y) * W + x) * C + c;
            if (out) *reinterpret_cast<float4*>(out + o) = make_float4(acc[mi][nj][0], acc[mi][nj][1], acc[mi][nj][2], acc[mi][nj][3]);
            if (out16) {
              uint2 qv; qv.x = pack_bf2(acc[mi][nj][0], acc[mi][nj][1]); qv.y = pack_bf2(acc[mi][nj][2], acc[mi][nj][3]);
              *reinterpret_cast<uint2*>(out16 + o) = qv;
            }
          }
        }
      }
    }
  }
}

// ---- the same low-res adaptive convolution for the EXACT tower mode (SG_PREC_F16X2, round 3) ------------------------------------------------
// f32-grade results on the f16 matrix pipe, as the tower's linears have them: Keff is built in f32 (the scalar arithmetic the kernel above
// started from) and split into two f16 planes F_hi + F_lo, the source rows are two-plane f16 (h2_t: [8 hi | 8 lo] per 8 channels, staged as they
// lie in HBM: 512 bytes per window position and 128-channel chunk), and every product is Win_hi.F_hi + Win_lo.F_hi + Win_hi.F_lo into one f32
// accumulator.  Stage outputs stay two-plane (the next stage's source), the last stage writes f32.  130 KB of LDS at r = 5: one workgroup per CU.
template <int R> struct LowX2Cfg {
  using L = LowCfg<R>;
  static constexpr int LDWB = ACM_CC * 4 + 32;                          // window row stride in BYTES: 128 two-plane channels + 32 (rows 8 banks apart)
  static constexpr size_t F_BYTES = (size_t)2 * 64 * L::LDK * 2;        // F_hi, F_lo
  static constexpr size_t W_BYTES = (size_t)L::KP * LDWB;
  static constexpr size_t K_BYTES = (size_t)64 * L::D2 * 4 + (size_t)2 * 8 * L::D * L::LWP * 4;
  static constexpr size_t LDS = F_BYTES + (W_BYTES > K_BYTES ? W_BYTES : K_BYTES);
};
template <int R>
__global__ __launch_bounds__(256) void jbu_conv_lowres_x2_kernel(const h2_t* __restrict__ src, const float* __restrict__ Kf, int ldk, int h, int w,
                                                                 int C, float* __restrict__ out, h2_t* __restrict__ out2) {
  using L = LowCfg<R>;
  using X = LowX2Cfg<R>;
  constexpr int D = L::D, D2 = L::D2, LW = L::LW, LWP = L::LWP, KP = L::KP, LDK = L::LDK, LDWB = X::LDWB;
  extern __shared__ __attribute__((aligned(16))) char lx_sm[];
  uint16_t* sFh = reinterpret_cast<uint16_t*>(lx_sm);                     // [64][LDK] f16
  uint16_t* sFl = sFh + 64 * LDK;
  char* sW = lx_sm + X::F_BYTES;                                          // [KP][LDWB]           (chunk loop)
  float* sK = reinterpret_cast<float*>(lx_sm + X::F_BYTES);              // [64][D2]             (Keff build; aliases sW)
  float* sWx = sK + 64 * D2;                                              // [8][D][LWP]
  float* sWy = sWx + 8 * D * LWP;
  const int H = 2 * h, W = 2 * w;
  const int tiles_x = (W + AC_T - 1) / AC_T;
  const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + seq;
  const int ty0 = (blk / tiles_x) * AC_T, tx0 = (blk % tiles_x) * AC_T;
  const int ly0 = ty0 / 2 - L::OFF, lx0 = tx0 / 2 - L::OFF;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- 1. tables + the block's kernel rows (f32) ----
  if (tid < 2 * 8 * D) {
    const bool isy = tid >= 8 * D;
    const int e = isy ? tid - 8 * D : tid, pl = e / D, t = e % D;
    const int size = isy ? H : W, lo_size = isy ? h : w, org = isy ? ty0 : tx0, lorg = isy ? ly0 : lx0;
    int u = org + pl + t - R;
    u = u > size - 1 + R ? size - 1 + R : u;
    u = reflect_idx(u, size);
    int idx[4]; float wt[4];
    cubic_taps(u, lo_size, size, idx, wt);
    float* row = (isy ? sWy : sWx) + (pl * D + t) * LWP;
#pragma unroll
    for (int c = 0; c < LWP; ++c) row[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int c = idx[k] - lorg;
      c = c < 0 ? 0 : (c > LW - 1 ? LW - 1 : c);
      row[isy ? c : c + c / 3] += wt[k];                                   // x rows: column 3q + j at slot 4q + j
    }
  }
  {
    constexpr int NK = (D2 * 64 + 255) / 256;
    float kv[NK];
#pragma unroll
    for (int u = 0; u < NK; ++u) {
      int i = tid + u * 256;
      i = i < D2 * 64 ? i : D2 * 64 - 1;
      const int pxl = i / D2, t = i % D2;
      int y = ty0 + (pxl >> 3), x = tx0 + (pxl & 7);
      y = y < H ? y : H - 1; x = x < W ? x : W - 1;
      kv[u] = Kf[(((int64_t)b * H + y) * W + x) * ldk + t];
    }
#pragma unroll
    for (int u = 0; u < NK; ++u) {
      const int i = tid + u * 256;
      if (i < D2 * 64) sK[i] = kv[u];
    }
  }
  __syncthreads();
  // ---- 2. Keff = Wy^T . K . Wx in f32, split into two f16 planes ----
  {
    const int p = tid >> 2, q = tid & 3, py = p >> 3, px = p & 7;
    const float* kp = sK + p * D2;
    const float* wx = sWx + px * D * LWP + 4 * q;
    const float* wy = sWy + py * D * LWP;
    float T[D][3];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float kv = kp[i * D + j];
        t0 += kv * wx[j * LWP]; t1 += kv * wx[j * LWP + 1]; t2 += kv * wx[j * LWP + 2];
      }
      T[i][0] = t0; T[i][1] = t1; T[i][2] = t2;
    }
    float e[LW][3];
#pragma unroll
    for (int ly = 0; ly < LW; ++ly) { e[ly][0] = 0.f; e[ly][1] = 0.f; e[ly][2] = 0.f; }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int ly = 0; ly < LW; ++ly) {
        const float wv = wy[i * LWP + ly];
        e[ly][0] += wv * T[i][0]; e[ly][1] += wv * T[i][1]; e[ly][2] += wv * T[i][2];
      }
    constexpr int QW = L::QW;
    uint32_t ph[QW / 2], pl[QW / 2];
#pragma unroll
    for (int j = 0; j < QW / 2; ++j) {
      const int i0 = 2 * j, i1 = 2 * j + 1;
      const float v0 = i0 < 3 * LW ? e[i0 / 3 < LW ? i0 / 3 : 0][i0 % 3] : 0.f;
      const float v1 = i1 < 3 * LW ? e[i1 / 3 < LW ? i1 / 3 : 0][i1 % 3] : 0.f;
      split_h2(v0, v1, ph[j], pl[j]);
    }
    uint16_t* fh = sFh + p * LDK + q * QW;
    uint16_t* fl = sFl + p * LDK + q * QW;
#pragma unroll
    for (int j = 0; j < QW / 4; ++j) {
      *reinterpret_cast<uint2*>(fh + 4 * j) = make_uint2(ph[2 * j], ph[2 * j + 1]);
      *reinterpret_cast<uint2*>(fl + 4 * j) = make_uint2(pl[2 * j], pl[2 * j + 1]);
    }
    if constexpr (KP > L::NPOS) {
      static_assert((KP - L::NPOS) == 16, "padding is written as one 8-byte piece per thread");
      *reinterpret_cast<uint2*>(sFh + p * LDK + L::NPOS + 4 * q) = make_uint2(0u, 0u);
      *reinterpret_cast<uint2*>(sFl + p * LDK + L::NPOS + 4 * q) = make_uint2(0u, 0u);
    }
  }
  const h2_t* sb = src + (int64_t)b * h * w * C;
  for (int c0 = 0; c0 < C; c0 += ACM_CC) {
    __syncthreads();                                                       // F complete / Keff build done with sK / previous chunk's MFMAs done with sW
    // ---- 3a. window [K slot][128 two-plane channels] as it lies in HBM: 32 x 16-byte pieces per position, two rounds of loads ----
    {
      constexpr int NPC = ACM_CC * 4 / 16;                                 // 16-byte pieces per position and chunk
      constexpr int NS = KP * NPC / 256 / 2;
      static_assert(KP * NPC % 512 == 0, "window pieces must divide over the threads and the two rounds");
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        uint4 wv[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          const int i = tid + (half * NS + u) * 256, pos = i / NPC, pc16 = i % NPC;
          const int pc = pos < L::NPOS ? pos : L::NPOS - 1;
          const int qd = pc / L::QW, rem = pc % L::QW;
          const int wy_ = rem / 3 < LW ? rem / 3 : LW - 1, wx_ = 3 * qd + rem % 3 < LW ? 3 * qd + rem % 3 : LW - 1;
          int sy = ly0 + wy_, sx = lx0 + wx_;
          sy = sy < 0 ? 0 : (sy > h - 1 ? h - 1 : sy); sx = sx < 0 ? 0 : (sx > w - 1 ? w - 1 : sx);
          int cc = c0 + 4 * pc16; cc = cc + 4 <= C ? cc : C - 4;           // piece = 4 h2 elements' worth of bytes: element index c0 + 4 * piece (ragged last chunk: a valid duplicate)
          wv[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sb + ((int64_t)sy * w + sx) * C) + (int64_t)cc * 4);
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          const int i = tid + (half * NS + u) * 256, pos = i / NPC, pc16 = i % NPC;
          *reinterpret_cast<uint4*>(sW + pos * LDWB + 16 * pc16) = wv[u];
        }
      }
    }
    __syncthreads();
    // ---- 3b. wave: channels [32 wave, +32) x 64 pixels, three f16 MFMAs per product ----
    f32x4 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((ext_vector_type(4))) short s4_t;
    typedef __attribute__((address_space(3))) s4_t* lds_s4_t;
    typedef __attribute__((ext_vector_type(8))) short s8_t;
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll 1
    for (int k0 = 0; k0 < KP; k0 += 32) {
      bf16x8 fah[4], fal[4], fwh[2], fwl[2];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        fah[mi] = *reinterpret_cast<const bf16x8*>(sFh + (mi * 16 + (lane & 15)) * LDK + k0 + g * 8);
        fal[mi] = *reinterpret_cast<const bf16x8*>(sFl + (mi * 16 + (lane & 15)) * LDK + k0 + g * 8);
      }
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) {
        // channels 32 wave + 16 nj + 4 pp .. + 3 of rows k0 + 8 g + qq (and + 4): storage group (8 channels) gi, first / second half of its plane
        const int gi = wave * 4 + nj * 2 + (pp >> 1);
        const char* p0 = sW + (k0 + 8 * g + qq) * LDWB + gi * 32 + (pp & 1) * 8;
#pragma unroll
        for (int plane = 0; plane < 2; ++plane) {
          const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(p0 + plane * 16));
          const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(p0 + plane * 16 + 4 * LDWB));
          const s8_t both = (s8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          if (plane == 0) fwh[nj] = __builtin_bit_cast(bf16x8, both); else fwl[nj] = __builtin_bit_cast(bf16x8, both);
        }
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          acc[mi][nj] = mfma_16x16x32<true>(fwh[nj], fah[mi], acc[mi][nj]);
          acc[mi][nj] = mfma_16x16x32<true>(fwl[nj], fah[mi], acc[mi][nj]);
          acc[mi][nj] = mfma_16x16x32<true>(fwh[nj], fal[mi], acc[mi][nj]);
        }
    }
    // lane: pixel mi*16 + (lane & 15), channels c0 + 32 wave + 16 nj + 4 (lane >> 4) .. +4
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int pxl = mi * 16 + (lane & 15);
      const int y = ty0 + (pxl >> 3), x = tx0 + (pxl & 7);
      if (y < H && x < W) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          const int c = c0 + wave * 32 + nj * 16 + (lane >> 4) * 4;
          if (c < C) {
            const int64_t o = (((int64_t)b * H + y) * W + x) * C + c;
            if (out) *reinterpret_cast<float4*>(out + o) = make_float4(acc[mi][nj][0], acc[mi][nj][1], acc[mi][nj][2], acc[mi][nj][3]);
            if (out2) {                                                    // two-plane row: element c sits in group c >> 3, half (c >> 2) & 1
              uint2 hv, lv;
              split_h2(acc[mi][nj][0], acc[mi][nj][1], hv.x, lv.x); split_h2(acc[mi][nj][2], acc[mi][nj][3], hv.y, lv.y);
              char* g8 = reinterpret_cast<char*>(out2 + (o & ~(int64_t)7)) + ((c >> 2) & 1) * 8;
              *reinterpret_cast<uint2*>(g8) = hv; *reinterpret_cast<uint2*>(g8 + 16) = lv;
            }
          }
        }
      }
    }
  }
}

// Stand-alone adaptive convolution in FeatUp's NCHW calling convention
// (featup.adaptive_conv_cuda.AdaptiveConv.apply as used at upsamplers.py:274; semantics restated from
// adaptive_conv_py_simple, upsamplers.py:14-25):  out[b,c,y,x] = sum_{i,j<d} in[b,c,y+i,x+j] * filt[b,y,x,i,j]
__global__ __launch_bounds__(256) void adaptive_conv_nchw_kernel(const float* __restrict__ in, const float* __restrict__ filt,
                                                                 int C, int h, int w, int d, float* __restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z / C, c = blockIdx.z % C;
  if (x >= w || y >= h) return;
  const int wp = w + d - 1, hp = h + d - 1;
  const float* ip = in + ((int64_t)(b * C + c) * hp + y) * wp + x;
  const float* fp = filt + (((int64_t)b * h + y) * w + x) * d * d;
  float acc = 0.f;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) acc += ip[i * wp + j] * fp[i * d + j];
  out[((int64_t)(b * C + c) * h + y) * w + x] = acc;
}

// tokens' = tokens - cls_hat * (cos(tokens, cls_hat) * factor)   (segmentor.py:322-336), cls_hat = cls / ||cls||
__global__ __launch_bounds__(256) void global_debias_kernel(const float* __restrict__ tokens, const float* __restrict__ cls, int n, int E,
                                                            float factor, float* __restrict__ out) {
  __shared__ float s_inv;
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* cr = cls + (int64_t)b * E;
  if (wave == 0) {
    float ss = 0.f;
    for (int i = lane; i < E; i += 64) ss += cr[i] * cr[i];
    ss = wave_sum(ss);
    if (lane == 0) s_inv = 1.0f / sqrtf(ss);
  }
  __syncthreads();
  const int t = blockIdx.x * 4 + wave;
  if (t >= n) return;
  const float ci = s_inv;
  const float* f = tokens + ((int64_t)b * n + t) * E;
  float ff = 0.f, fc = 0.f, cc = 0.f;
  for (int i = lane; i < E; i += 64) { const float x = f[i], c = cr[i] * ci; ff += x * x; fc += x * c; cc += c * c; }
  ff = wave_sum(ff); fc = wave_sum(fc); cc = wave_sum(cc);
  const float w = (fc / (sqrtf(ff) * sqrtf(cc))) * factor;
  float* o = out + ((int64_t)b * n + t) * E;
  for (int i = lane; i < E; i += 64) o[i] = f[i] - cr[i] * ci * w;
}

struct JbuStage {
  int r;
  float *range_temp, *sigma, *rp0_w, *rp0_b, *rp3_w, *rp3_b, *fx0_w, *fx0_b, *fx3_w, *fx3_b;
  // throughput mode: the two fixup linears on the bf16 MFMA GEMM, operands zero-padded to [NP, KP1] / [NP, NP] (NP, KP1 multiples of 64)
  void *fx0_w16, *fx3_w16; float *fx0_bp, *fx3_bp;
  void *fx0_w16h, *fx3_w16h;                              // the same operands in f16: the all-2-byte fixup chain of the low-res path (round 3)
  void *fx0_wh2, *fx3_wh2;                                // ... and as two-plane f16 (SG_PREC_F16X2: f32-grade linears on the f16 matrix pipe)
};
static inline int jbu_np(int r) { const int d = 2 * r + 1; return (int)align_up((size_t)d * d, 64); }
static inline int jbu_kp1(int r) { const int d = 2 * r + 1; return (int)align_up((size_t)d * d + 3, 64); }

}  // namespace sg

using namespace sg;

struct sg_jbu {
  int device, kind, C, n_stage_sets;
  void* arena; size_t arena_bytes;
  JbuStage st[4];
  float *fin_w, *fin_b;
  void* fin_w16;
  void* fin_wh2;                                           // two-plane f16 copy of the final 1x1 weight (C % 32 == 0)
  std::vector<uint8_t> have;
};

namespace sg {
static const char* kStageTensor[10] = {"range_temp", "sigma_spatial", "range_proj.0.weight", "range_proj.0.bias", "range_proj.3.weight",
                                       "range_proj.3.bias", "fixup_proj.0.weight", "fixup_proj.0.bias", "fixup_proj.3.weight",
                                       "fixup_proj.3.bias"};
static int64_t stage_numel(int slot, int r) {
  const int64_t d2 = (int64_t)(2 * r + 1) * (2 * r + 1);
  switch (slot) {
    case 0: case 1: return 1;
    case 2: return KEY_DIM * 3; case 3: return KEY_DIM; case 4: return KEY_DIM * KEY_DIM; case 5: return KEY_DIM;
    case 6: return d2 * (d2 + 3); case 7: return d2; case 8: return d2 * d2; default: return d2;
  }
}
__global__ void scale_kernel(float* p, float a, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] *= a;
}
}  // namespace sg

extern "C" int sg_jbu_create(sg_jbu** out, int device, int kind, int feat_dim) {
  SG_REQUIRE(out && (kind == 0 || kind == 1) && feat_dim > 0 && feat_dim % 4 == 0, "sg_jbu_create: bad arguments (kind 0 = jbu_one, 1 = jbu_stack; feat_dim %% 4 == 0)");
  DeviceGuard dg(device);
  sg_jbu* j = new sg_jbu();
  j->device = device; j->kind = kind; j->C = feat_dim; j->n_stage_sets = kind == 0 ? 1 : 4;
  const int r = kind == 0 ? 5 : 3;                     // JBUOne radius 5, JBUStack radius 3 (upsamplers.py:281-284,308)
  size_t bytes = 0;
  auto lay = [&](char* base) {
    size_t off = 0;
    auto take = [&](size_t n) { off = align_up(off, 256); void* p = base ? base + off : nullptr; off += n; return p; };
    for (int s = 0; s < j->n_stage_sets; ++s) {
      JbuStage& S = j->st[s]; S.r = r;
      float** slots[10] = {&S.range_temp, &S.sigma, &S.rp0_w, &S.rp0_b, &S.rp3_w, &S.rp3_b, &S.fx0_w, &S.fx0_b, &S.fx3_w, &S.fx3_b};
      for (int t = 0; t < 10; ++t) *slots[t] = (float*)take((size_t)stage_numel(t, r) * 4);
      S.fx0_w16 = take((size_t)jbu_np(r) * jbu_kp1(r) * 2); S.fx3_w16 = take((size_t)jbu_np(r) * jbu_np(r) * 2);
      S.fx0_bp = (float*)take((size_t)jbu_np(r) * 4); S.fx3_bp = (float*)take((size_t)jbu_np(r) * 4);
      S.fx0_w16h = take((size_t)jbu_np(r) * jbu_kp1(r) * 2); S.fx3_w16h = take((size_t)jbu_np(r) * jbu_np(r) * 2);
      S.fx0_wh2 = take((size_t)jbu_np(r) * jbu_kp1(r) * 4); S.fx3_wh2 = take((size_t)jbu_np(r) * jbu_np(r) * 4);
    }
    j->fin_w = (float*)take((size_t)feat_dim * feat_dim * 4);
    j->fin_b = (float*)take((size_t)feat_dim * 4);
    j->fin_w16 = take((size_t)feat_dim * feat_dim * 2);
    j->fin_wh2 = take((size_t)feat_dim * feat_dim * 4);
    bytes = align_up(off, 256);
  };
  lay(nullptr);
  j->arena_bytes = bytes;
  hipError_t e = hipMalloc(&j->arena, bytes);
  if (e != hipSuccess) { delete j; return fail(SG_ERR_HIP, "sg_jbu_create: hipMalloc(%zu) -> %s", bytes, hipGetErrorString(e)); }
  lay((char*)j->arena);
  SG_HIP(hipMemset(j->arena, 0, bytes));                  // the zero padding of the bf16 fixup operands
  j->have.assign(j->n_stage_sets * 10 + 2, 0);
  *out = j;
  return SG_OK;
}

extern "C" void sg_jbu_destroy(sg_jbu* j) {
  if (!j) return;
  if (j->arena) (void)hipFree(j->arena);
  delete j;
}

extern "C" int sg_jbu_set_tensor(sg_jbu* j, const char* name, const float* src, int64_t numel, sg_stream st) {
  SG_REQUIRE(j && name && src, "sg_jbu_set_tensor: null argument");
  DeviceGuard dg(j->device);
  hipStream_t s = as_stream(st);
  auto put = [&](float* dst, int64_t n) -> int {
    SG_REQUIRE(numel == n, "sg_jbu_set_tensor(%s): expected %lld elements, got %lld", name, (long long)n, (long long)numel);
    SG_HIP(hipMemcpyAsync(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    return SG_OK;
  };
  if (!strcmp(name, "fixup_proj.1.weight")) {
    SG_TRY(put(j->fin_w, (int64_t)j->C * j->C));
    if (j->C % 64 == 0) SG_TRY(pack_rows(src, j->C, j->C, j->C, j->fin_w16, j->C, 1, s));
    if (j->C % 32 == 0) SG_TRY(pack_rows(src, j->C, j->C, j->C, j->fin_wh2, j->C, HK_F16X2, s));
    j->have[j->n_stage_sets * 10] = 1; return SG_OK;
  }
  if (!strcmp(name, "fixup_proj.1.bias")) {
    SG_TRY(put(j->fin_b, j->C));
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)cdiv(j->C, 256)), dim3(256), 0, s, j->fin_b, 0.1f, (int64_t)j->C);   // epilogue is alpha*acc + bias
    SG_LAUNCH_CHECK();
    j->have[j->n_stage_sets * 10 + 1] = 1; return SG_OK;
  }
  int set = -1; const char* rest = nullptr;
  if (j->kind == 0 && !strncmp(name, "up.", 3)) { set = 0; rest = name + 3; }
  else if (j->kind == 1 && !strncmp(name, "up", 2) && name[2] >= '1' && name[2] <= '4' && name[3] == '.') { set = name[2] - '1'; rest = name + 4; }
  if (set >= 0) {
    for (int t = 0; t < 10; ++t)
      if (!strcmp(rest, kStageTensor[t])) {
        JbuStage& S = j->st[set];
        float* slots[10] = {S.range_temp, S.sigma, S.rp0_w, S.rp0_b, S.rp3_w, S.rp3_b, S.fx0_w, S.fx0_b, S.fx3_w, S.fx3_b};
        SG_TRY(put(slots[t], stage_numel(t, S.r)));
        if (t == 9) {                                  // K += 0.1 * (H.W3^T + b3): the GEMM epilogue computes 0.1*acc + bias
          hipLaunchKernelGGL(scale_kernel, dim3((unsigned)cdiv(numel, 256)), dim3(256), 0, s, slots[t], 0.1f, numel);
          SG_LAUNCH_CHECK();
        }
        const int d2s = (2 * S.r + 1) * (2 * S.r + 1);
        if (t == 6) { SG_TRY(pack_rows(src, d2s, d2s + 3, d2s + 3, S.fx0_w16, jbu_kp1(S.r), 1, s)); SG_TRY(pack_rows(src, d2s, d2s + 3, d2s + 3, S.fx0_w16h, jbu_kp1(S.r), HK_F16, s)); SG_TRY(pack_rows(src, d2s, d2s + 3, d2s + 3, S.fx0_wh2, jbu_kp1(S.r), HK_F16X2, s)); }
        if (t == 8) { SG_TRY(pack_rows(src, d2s, d2s, d2s, S.fx3_w16, jbu_np(S.r), 1, s)); SG_TRY(pack_rows(src, d2s, d2s, d2s, S.fx3_w16h, jbu_np(S.r), HK_F16, s)); SG_TRY(pack_rows(src, d2s, d2s, d2s, S.fx3_wh2, jbu_np(S.r), HK_F16X2, s)); }
        if (t == 7) SG_HIP(hipMemcpyAsync(S.fx0_bp, slots[t], (size_t)d2s * 4, hipMemcpyDeviceToDevice, s));
        if (t == 9) SG_HIP(hipMemcpyAsync(S.fx3_bp, slots[t], (size_t)d2s * 4, hipMemcpyDeviceToDevice, s));
        j->have[set * 10 + t] = 1;
        return SG_OK;
      }
  }
  return fail(SG_ERR_INVALID, "sg_jbu_set_tensor: unknown tensor name '%s' for this upsampler kind", name);
}

namespace sg {
struct JbuPlan { float *gs, *proj, *X, *H1, *Kf, *hr, *bufA, *bufB; void* x16; bf16_t *X16, *H116; float *rowdot, *geff, *g0, *clsl; };
constexpr int JBU_QMAX = 32;                         // queries the fused logits tail keeps in registers
static size_t jbu_plan(const sg_jbu* j, int B, int gh, int gw, void* ws, bool dry, JbuPlan& p) {
  const int r = j->st[0].r, d2 = (2 * r + 1) * (2 * r + 1);
  const int64_t pixels = (int64_t)B * 16 * gh * 16 * gw;              // final resolution
  size_t off = 0;
  auto take = [&](size_t n) { off = align_up(off, 256); void* q = dry ? nullptr : (char*)ws + off; off += n; return q; };
  p.gs = (float*)take((size_t)pixels * 3 * 4);
  p.proj = (float*)take((size_t)pixels * KEY_DIM * 4);
  const int NP = jbu_np(r), KP1 = jbu_kp1(r);
  p.X = (float*)take((size_t)pixels * (d2 + 3) * 4 + 1024);           // + slack: the padded fixup GEMM reads its residual NP columns wide
  p.H1 = (float*)take((size_t)pixels * d2 * 4);
  p.Kf = (float*)take((size_t)pixels * NP * 4);                       // row stride d2 (parity mode) or NP (throughput mode)
  p.X16 = (bf16_t*)take((size_t)pixels * KP1 * 4);                      // 2-byte operand rows (throughput mode) or two-plane f16 ones (SG_PREC_F16X2: 4 bytes per element)
  p.H116 = (bf16_t*)take((size_t)pixels * NP * 4);
  p.hr = (float*)take((size_t)pixels * j->C * 4);
  p.bufA = (float*)take((size_t)pixels / 4 * j->C * 4);                // stage-3 output (8x): ping
  p.bufB = (float*)take((size_t)pixels * j->C * 4);                    // stage-2 / stage-4 output: pong
  p.x16 = take((size_t)pixels * j->C * 2);
  p.rowdot = (float*)take((size_t)pixels * (j->C / 64 + 1) * 4);       // fused tail: per-pixel partial |out|^2 - |x|^2, one slot per 64 columns
  p.geff = (float*)take((size_t)j->C * JBU_QMAX * 4);
  p.g0 = (float*)take((size_t)JBU_QMAX * 4);
  p.clsl = (float*)take((size_t)B * JBU_QMAX * 4);
  return align_up(off, 256);
}
}  // namespace sg

extern "C" size_t sg_jbu_workspace_bytes(const sg_jbu* j, int B, int gh, int gw) {
  if (!j || B <= 0 || gh <= 0 || gw <= 0) return 0;
  JbuPlan p;
  return jbu_plan(j, B, gh, gw, nullptr, true, p);
}

// The four 2x stages (JBULearnedRange.forward x 4): source [B, gh*gw, C] -> *x_out [B, 16gh*16gw, C] f32 inside the workspace
// (and its bf16 copy in p.x16 in throughput mode when C % 64 == 0)
static int jbu_stages(sg_jbu* j, const float* source, const float* guidance, int B, int gh, int gw, int GH, int GW, int precision,
                      const JbuPlan& p, const float** x_out, hipStream_t s, bool want_f32_x = true, bool* x2_in_hr = nullptr) {
  const int C = j->C;
  const float* src = source;
  int h = gh, w = gw;
  // low-res adaptive convolution (throughput mode): its bf16 stage outputs live in the region the hi-res tensor would have used
  const bool lowres_ok = precision == SG_PREC_BF16 && C % 64 == 0 && gh >= 2 && gw >= 2;
  bf16_t* tok16 = (bf16_t*)p.hr;
  bf16_t* o16[3];
  {
    size_t off = align_up((size_t)B * gh * gw * C * 2, 256);
    for (int t = 0; t < 3; ++t) { o16[t] = (bf16_t*)((char*)p.hr + off); off += align_up((size_t)B * (gh << (t + 1)) * (gw << (t + 1)) * C * 2, 256); }
  }
  // exact tower mode: the same low-res formulation on two-plane f16 operands (jbu_conv_lowres_x2_kernel); its stage outputs (4 bytes per element)
  // live where the hi-res tensor would have
  const bool x2low_ok = precision == SG_PREC_F16X2 && C % 64 == 0 && gh >= 2 && gw >= 2;
  h2_t* tok2 = (h2_t*)p.hr;
  h2_t* o2[3];
  {
    size_t off = align_up((size_t)B * gh * gw * C * 4, 256);
    for (int t = 0; t < 3; ++t) { o2[t] = (h2_t*)((char*)p.hr + off); off += align_up((size_t)B * (gh << (t + 1)) * (gw << (t + 1)) * C * 4, 256); }
    // the 8x stage's output goes to the f32 path's 8x buffer (bufA: exactly pixels / 4 x C x 4 bytes, unused by this chain) instead: the hi-res
    // region is then dead while the LAST stage runs, which writes the two-plane copy of x there for the final 1x1 GEMM -- no pack pass
    o2[2] = (h2_t*)p.bufA;
  }
  if (x2_in_hr) *x2_in_hr = false;
  for (int stg = 0; stg < 4; ++stg) {
    const JbuStage& S = j->st[j->kind == 0 ? 0 : stg];
    const int r = S.r, d = 2 * r + 1, d2 = d * d, oh = 2 * h, ow = 2 * w;
    SG_REQUIRE(r < oh && r < ow, "sg_jbu_upsample: reflect padding %d needs a guidance grid larger than %dx%d", r, oh, ow);
    SG_REQUIRE(d2 <= 128, "sg_jbu_upsample: window %d too large", d);
    const int64_t pixels = (int64_t)B * oh * ow;
    // ping-pong: source -> bufA (2x) -> bufB (4x) -> bufA (8x) -> bufB (16x)
    float* dst = (stg % 2 == 0) ? p.bufA : p.bufB;
    hipLaunchKernelGGL(jbu_pool_kernel, dim3((unsigned)cdiv(pixels * 3, 256)), dim3(256), 0, s, guidance, B, GH, GW, oh, ow, p.gs);
    SG_LAUNCH_CHECK();
    hipLaunchKernelGGL(jbu_range_proj_kernel, dim3((unsigned)cdiv(pixels, 256)), dim3(256), 0, s, p.gs, pixels, S.rp0_w, S.rp0_b, S.rp3_w,
                       S.rp3_b, p.proj);
    SG_LAUNCH_CHECK();
    const bool fast = precision == SG_PREC_BF16 && C % 8 == 0;       // throughput mode: bf16 MFMA for the fixup linears and the convolution
    const bool x2 = precision == SG_PREC_F16X2;                       // exact tower mode: f32 kernels, the three linears as two-plane f16 GEMMs (f32-grade, 3 f16 MFMAs per product)
    const int NP = jbu_np(r), KP1 = jbu_kp1(r), ldk = (fast || x2) ? NP : d2;
    // the low-res path (below) runs its fixup chain in f16 end to end: operand rows, GELU output, residual and the kernel rows the convolution
    // converts to f16 anyway -- 3040 -> 1792 bytes of HBM traffic per pixel and stage, and f16's 11 bits in place of bf16's 8 on the way
    const bool h16 = fast && (r == 5 || r == 3) && lowres_ok;
    {
      const int WT = AC_T + 2 * r;
      const size_t lds = (size_t)WT * WT * JK_LD * sizeof(float);
      SG_REQUIRE(lds <= 64 * 1024, "sg_jbu_upsample: window %d too large", d);
      using JK = void (*)(const float*, const float*, int, int, int, const float*, const float*, float*, bf16_t*, int);
      const JK jk = fast ? (r == 5 ? jbu_kernel_tiled_kernel<5, true> : r == 3 ? jbu_kernel_tiled_kernel<3, true> : jbu_kernel_tiled_kernel<0, true>)
                         : jbu_kernel_tiled_kernel<0, false>;
      if (x2 && (r == 5 || r == 3)) {                       // exact mode: two-plane keys, exact arithmetic, two-plane operand rows written directly
        const dim3 grid((unsigned)(cdiv(oh, AC_T) * cdiv(ow, AC_T)), (unsigned)B);
        constexpr size_t lds5 = JkmCfg<5, true>::LDS, lds3 = JkmCfg<3, true>::LDS;
        if (r == 5) {
          SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(jbu_kernel_mfma_kernel<5, true>), lds5));
          hipLaunchKernelGGL((jbu_kernel_mfma_kernel<5, true>), grid, dim3(256), lds5, s, p.proj, p.gs, oh, ow, S.range_temp, S.sigma, p.X, p.X16, KP1, 0);
        } else {
          SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(jbu_kernel_mfma_kernel<3, true>), lds3));
          hipLaunchKernelGGL((jbu_kernel_mfma_kernel<3, true>), grid, dim3(256), lds3, s, p.proj, p.gs, oh, ow, S.range_temp, S.sigma, p.X, p.X16, KP1, 0);
        }
      } else if (fast && (r == 5 || r == 3)) {               // key dot products on the matrix pipe
        const dim3 grid((unsigned)(cdiv(oh, AC_T) * cdiv(ow, AC_T)), (unsigned)B);
        if (r == 5) {
          SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(jbu_kernel_mfma_kernel<5>), JkmCfg<5>::LDS));
          hipLaunchKernelGGL(jbu_kernel_mfma_kernel<5>, grid, dim3(256), JkmCfg<5>::LDS, s, p.proj, p.gs, oh, ow, S.range_temp, S.sigma, p.X, p.X16, KP1, h16 ? 1 : 0);
        } else {
          SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(jbu_kernel_mfma_kernel<3>), JkmCfg<3>::LDS));
          hipLaunchKernelGGL(jbu_kernel_mfma_kernel<3>, grid, dim3(256), JkmCfg<3>::LDS, s, p.proj, p.gs, oh, ow, S.range_temp, S.sigma, p.X, p.X16, KP1, h16 ? 1 : 0);
        }
      } else {
      SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(jk), 64 * 1024));
      hipLaunchKernelGGL(jk, dim3((unsigned)(cdiv(oh, AC_T) * cdiv(ow, AC_T)), (unsigned)B), dim3(256), lds, s, p.proj, p.gs, oh,
                         ow, r, S.range_temp, S.sigma, p.X, fast ? p.X16 : nullptr, KP1);
      }
      SG_LAUNCH_CHECK();
    }
    SG_REQUIRE(pixels < (1ll << 31), "sg_jbu_upsample: too many pixels");
    if (fast) {  // H1 = GELU(X . W0^T + b0) (bf16);  Kf = X[:, :d2] + 0.1 * (H1 . W3^T + b3), columns >= d2 are padding
      GemmBf16Args g{};
      g.A = p.X16; g.lda = KP1; g.W = (const bf16_t*)(h16 ? S.fx0_w16h : S.fx0_w16); g.ldw = KP1; g.bias = S.fx0_bp; g.C = p.H116; g.ldc = NP; g.c_is_bf16 = 1;
      g.M = (int)pixels; g.N = NP; g.K = KP1; g.batch = 1; g.act = ACT_GELU; g.alpha = 1.f; g.f16 = h16 ? 1 : 0;
      SG_TRY(gemm_bf16(g, s));
      GemmBf16Args q{};
      q.A = p.H116; q.lda = NP; q.W = (const bf16_t*)(h16 ? S.fx3_w16h : S.fx3_w16); q.ldw = NP; q.bias = S.fx3_bp;
      q.C = p.Kf; q.ldc = NP; q.M = (int)pixels; q.N = NP; q.K = NP; q.batch = 1; q.act = ACT_NONE; q.alpha = 0.1f;
      if (h16) { q.residual = reinterpret_cast<const float*>(p.X16); q.ldr = KP1; q.res_half = 1; q.c_is_bf16 = 1; q.f16 = 1; }   // Kf rows in f16, residual = the operand rows
      else { q.residual = p.X; q.ldr = d2 + 3; q.c_is_bf16 = 0; }
      SG_TRY(gemm_bf16(q, s));
    } else if (x2) {  // the same two linears on the two-plane GEMM: X rows packed to [KP1] two-plane, GELU output two-plane, Kf f32 rows of NP
      if (!(r == 5 || r == 3)) SG_TRY(pack_rows(p.X, pixels, d2 + 3, d2 + 3, p.X16, KP1, HK_F16X2, s));   // (r = 3 / 5: the range kernel wrote the two-plane rows itself)
      GemmBf16Args g{};
      g.A = p.X16; g.lda = KP1; g.W = (const bf16_t*)S.fx0_wh2; g.ldw = KP1; g.bias = S.fx0_bp; g.C = p.H116; g.ldc = NP; g.c_is_bf16 = 1;
      g.M = (int)pixels; g.N = NP; g.K = KP1; g.batch = 1; g.act = ACT_GELU; g.alpha = 1.f; g.h2 = 1;
      SG_TRY(gemm_bf16(g, s));
      GemmBf16Args q{};
      q.A = p.H116; q.lda = NP; q.W = (const bf16_t*)S.fx3_wh2; q.ldw = NP; q.bias = S.fx3_bp; q.residual = p.X; q.ldr = d2 + 3;
      q.C = p.Kf; q.ldc = NP; q.c_is_bf16 = 0; q.M = (int)pixels; q.N = NP; q.K = NP; q.batch = 1; q.act = ACT_NONE; q.alpha = 0.1f; q.h2 = 1;
      SG_TRY(gemm_bf16(q, s));
    } else {  // fixup: H1 = GELU(X . W0^T + b0);  Kf = X[:, :d2] + 0.1 * (H1 . W3^T + b3)
      GemmF32Args g{};
      g.A = p.X; g.lda = d2 + 3; g.B = S.fx0_w; g.sbk = 1; g.sbn = d2 + 3; g.bias = S.fx0_b; g.C = p.H1; g.ldc = d2;
      g.M = (int)pixels; g.N = d2; g.K = d2 + 3; g.batch = 1; g.inner = 1; g.act = ACT_GELU; g.alpha = 1.f;
      SG_TRY(gemm_f32(g, s));
      GemmF32Args q{};
      q.A = p.H1; q.lda = d2; q.B = S.fx3_w; q.sbk = 1; q.sbn = d2; q.bias = S.fx3_b; q.residual = p.X; q.ldr = d2 + 3; q.C = p.Kf; q.ldc = d2;
      q.M = (int)pixels; q.N = d2; q.K = d2; q.batch = 1; q.inner = 1; q.act = ACT_NONE; q.alpha = 0.1f;
      SG_TRY(gemm_f32(q, s));
    }
    if (x2 && x2low_ok && (r == 5 || r == 3)) {
      const h2_t* s2 = stg == 0 ? tok2 : o2[stg - 1];
      if (stg == 0) SG_TRY(pack_rows(source, (int64_t)B * gh * gw, C, C, tok2, C, HK_F16X2, s));
      h2_t* d2 = stg == 3 ? (x2_in_hr ? (h2_t*)p.hr : nullptr) : o2[stg];
      float* d32 = stg == 3 ? dst : nullptr;
      if (stg == 3 && x2_in_hr) *x2_in_hr = true;
      dim3 grid((unsigned)(cdiv(oh, AC_T) * cdiv(ow, AC_T)), (unsigned)B);
      if (r == 5) {
        SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(jbu_conv_lowres_x2_kernel<5>), LowX2Cfg<5>::LDS));
        hipLaunchKernelGGL(jbu_conv_lowres_x2_kernel<5>, grid, dim3(256), LowX2Cfg<5>::LDS, s, s2, p.Kf, ldk, h, w, C, d32, d2);
      } else {
        SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(jbu_conv_lowres_x2_kernel<3>), LowX2Cfg<3>::LDS));
        hipLaunchKernelGGL(jbu_conv_lowres_x2_kernel<3>, grid, dim3(256), LowX2Cfg<3>::LDS, s, s2, p.Kf, ldk, h, w, C, d32, d2);
      }
      SG_LAUNCH_CHECK();
      src = dst; h = oh; w = ow;
      continue;
    }
    const bool mfma_conv = fast;
    const bool lowres = fast && (r == 5 || r == 3) && lowres_ok;   // bicubic folded into the per-pixel kernel: no hi-res tensor at all
    if (lowres) {
      // bf16 chain: tokens -> o16[0] (2x) -> o16[1] (4x) -> o16[2] (8x) -> x16 (16x); f32 only out of the last stage
      bf16_t* s16 = stg == 0 ? tok16 : o16[stg - 1];
      if (stg == 0) SG_TRY(pack_rows(source, (int64_t)B * gh * gw, C, C, tok16, C, 1, s));
      bf16_t* d16 = stg == 3 ? (bf16_t*)p.x16 : o16[stg];
      float* d32 = (stg == 3 && want_f32_x) ? dst : nullptr;   // the fused tail works from the bf16 copy alone: 4 B per element not written
      dim3 grid((unsigned)(cdiv(oh, AC_T) * cdiv(ow, AC_T)), (unsigned)B);
      if (r == 5) {
        SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(jbu_conv_lowres_kernel<5>), LowCfg<5>::LDS));
        hipLaunchKernelGGL(jbu_conv_lowres_kernel<5>, grid, dim3(256), LowCfg<5>::LDS, s, s16, p.Kf, ldk, h, w, C, d32, d16, h16 ? 1 : 0);
      } else {
        SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(jbu_conv_lowres_kernel<3>), LowCfg<3>::LDS));
        hipLaunchKernelGGL(jbu_conv_lowres_kernel<3>, grid, dim3(256), LowCfg<3>::LDS, s, s16, p.Kf, ldk, h, w, C, d32, d16, h16 ? 1 : 0);
      }
      SG_LAUNCH_CHECK();
      src = dst; h = oh; w = ow;
      continue;
    }
    if (mfma_conv) hipLaunchKernelGGL(jbu_bicubic_kernel<bf16_t>, dim3((unsigned)cdiv(pixels * (C / 4), 256)), dim3(256), 0, s, src, B, h, w, C, oh, ow, (bf16_t*)p.hr);
    else hipLaunchKernelGGL(jbu_bicubic_kernel<float>, dim3((unsigned)cdiv(pixels * (C / 4), 256)), dim3(256), 0, s, src, B, h, w, C, oh, ow, p.hr);
    SG_LAUNCH_CHECK();
    if (mfma_conv) {                                         // throughput mode: the matrix-core formulation
      const int WT = AC_T + 2 * r, KP = (int)align_up((size_t)WT * WT, 32);
      const size_t lds = (size_t)(64 + ACM_CC) * (KP + 8) * sizeof(bf16_t);
      SG_REQUIRE(lds <= 160 * 1024, "sg_jbu_upsample: window %d needs %zu bytes of LDS", d, lds);
      SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(jbu_adaptive_conv_mfma_kernel), 160 * 1024));
      dim3 grid((unsigned)(cdiv(oh, AC_T) * cdiv(ow, AC_T)), (unsigned)B);
      hipLaunchKernelGGL(jbu_adaptive_conv_mfma_kernel, grid, dim3(256), lds, s, (const bf16_t*)p.hr, p.Kf, ldk, oh, ow, C, r, KP, dst,
                         (stg == 3 && C % 64 == 0) ? (bf16_t*)p.x16 : nullptr);
      SG_LAUNCH_CHECK();
    } else {
      const int WT = AC_T + 2 * r;
      const size_t lds = ((size_t)WT * WT * AC_LD + (size_t)d2 * 64) * sizeof(float);
      if (lds > 48 * 1024) SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(jbu_adaptive_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      dim3 grid((unsigned)(cdiv(oh, AC_T) * cdiv(ow, AC_T)), (unsigned)cdiv(C, AC_CC), (unsigned)B);
      hipLaunchKernelGGL(jbu_adaptive_conv_kernel, grid, dim3(256), lds, s, p.hr, p.Kf, ldk, oh, ow, C, r, dst);
      SG_LAUNCH_CHECK();
    }
    src = dst; h = oh; w = ow;
  }
  *x_out = src;
  return SG_OK;
}

// source [B, gh*gw, C] (patch tokens), guidance [B,3,GH,GW] (the normalised, padded tile) -> out [B, (16gh*16gw), C]
extern "C" int sg_jbu_upsample(sg_jbu* j, const float* source, const float* guidance, int B, int gh, int gw, int GH, int GW, int precision,
                               float* out, void* ws, size_t ws_bytes, sg_stream st) {
  SG_REQUIRE(j && source && guidance && out && ws, "sg_jbu_upsample: null argument");
  for (size_t i = 0; i < j->have.size(); ++i) if (!j->have[i]) return fail(SG_ERR_STATE, "sg_jbu_upsample: upsampler weights incomplete");
  DeviceGuard dg(j->device);
  hipStream_t s = as_stream(st);
  JbuPlan p;
  const size_t need = jbu_plan(j, B, gh, gw, ws, false, p);
  if (need > ws_bytes) return fail(SG_ERR_STATE, "sg_jbu_upsample: workspace %zu < required %zu", ws_bytes, need);
  const int C = j->C;
  const float* src = nullptr;
  bool x2_ready = false;
  SG_TRY(jbu_stages(j, source, guidance, B, gh, gw, GH, GW, precision, p, &src, s, true, &x2_ready));
  // out = x + 0.1 * (x . Wf^T + bf)     (bias pre-scaled by 0.1 at load)
  const int64_t pixels = (int64_t)B * 16 * gh * 16 * gw;
  if (precision == SG_PREC_BF16 && C % 64 == 0) {
    if (C % 8 != 0) SG_TRY(pack_rows(src, pixels, C, C, p.x16, C, 1, s));    // (C % 64 == 0 implies the matrix-core conv wrote x16 already)
    GemmBf16Args g{};
    g.A = (const bf16_t*)p.x16; g.lda = C; g.W = (const bf16_t*)j->fin_w16; g.ldw = C; g.bias = j->fin_b; g.residual = src; g.ldr = C;
    g.C = out; g.ldc = C; g.c_is_bf16 = 0; g.M = (int)pixels; g.N = C; g.K = C; g.batch = 1; g.act = 0; g.alpha = 0.1f;
    return gemm_bf16(g, s);
  }
  if (precision == SG_PREC_F16X2 && C % 32 == 0 && pixels >= 1024) {   // exact tower mode: x packed to two-plane f16 (the hi-res scratch is free by now), f32-grade GEMM
    if (!x2_ready) SG_TRY(pack_rows(src, pixels, C, C, p.hr, C, HK_F16X2, s));   // (the two-plane low-res convolution wrote this copy itself)
    GemmBf16Args g{};
    g.A = (const bf16_t*)p.hr; g.lda = C; g.W = (const bf16_t*)j->fin_wh2; g.ldw = C; g.bias = j->fin_b; g.residual = src; g.ldr = C;
    g.C = out; g.ldc = C; g.c_is_bf16 = 0; g.M = (int)pixels; g.N = C; g.K = C; g.batch = 1; g.act = 0; g.alpha = 0.1f; g.h2 = 1;
    return gemm_bf16(g, s);
  }
  GemmF32Args g{};
  g.A = src; g.lda = C; g.B = j->fin_w; g.sbk = 1; g.sbn = C; g.bias = j->fin_b; g.residual = src; g.ldr = C; g.C = out; g.ldc = C;
  g.M = (int)pixels; g.N = C; g.K = C; g.batch = 1; g.inner = 1; g.act = 0; g.alpha = 0.1f;
  return gemm_f32(g, s);
}

// ---- fused tail (throughput mode): per-pixel class logits WITHOUT writing the C x S^2 feature map ---------------------------------
// reference: out = x + 0.1 * fixup_proj(x) (upsamplers.py:301,325); feats /= |feats|; logits = feats . T^T (+ lambda * cls_logits)
// (segmentor.py:374-379).  With z = 0.1 * (x Wf^T + bf):
//     out . T[q]  = x . (T[q] + 0.1 Wf^T T[q]) + 0.1 bf . T[q]  =  x . Geff[:, q] + g0[q]          (f32, no GEMM: Q <= 32)
//     |out|^2     = |x|^2 + sum_c z (2 x + z)                                                       (the GEMM's row-dot epilogue)
// so the only C x C GEMM keeps its result in registers and HBM sees x once more (f32) plus Q floats per pixel.
namespace sg {
__global__ __launch_bounds__(256) void jbu_geff_kernel(const float* __restrict__ text, const float* __restrict__ Wf, const float* __restrict__ bf01,
                                                       int C, int Q, float* __restrict__ geff, float* __restrict__ g0) {
  // block = 64 channels x 4 K-quarters (one per wave): Wf[k][c] reads are coalesced over c, 8 independent partial sums per thread
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, q = blockIdx.y;
  const float* tq = text + (int64_t)q * C;
  const int kq = (C + 3) / 4, k0 = wave * kq, k1 = k0 + kq < C ? k0 + kq : C;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += Wf[(int64_t)(k + u) * C + c] * tq[k + u];
    }
    for (; k < k1; ++k) a[0] += Wf[(int64_t)k * C + c] * tq[k];
  }
  part[wave][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (wave == 0 && c < C) geff[c * JBU_QMAX + q] = tq[c] + 0.1f * (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]);   // (Wf^T T^T)[c, q]
  if (blockIdx.x == 0 && wave == 1) {
    float s0 = 0.f;
    for (int k = lane; k < C; k += 64) s0 += bf01[k] * tq[k];                // bf01 = 0.1 * bias (scaled at load)
    s0 = wave_sum(s0);
    if (lane == 0) g0[q] = s0;
  }
}
// cls_logits[b, q] = (cls[b] / |cls[b]|) . T[q]     (segmentor.py:309-311)
__global__ __launch_bounds__(64) void jbu_cls_logits_kernel(const float* __restrict__ cls, const float* __restrict__ text, int C, int Q,
                                                            float* __restrict__ out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* cr = cls + (int64_t)b * C;
  float ss = 0.f;
  for (int i = lane; i < C; i += 64) ss += cr[i] * cr[i];
  const float inv = 1.0f / sqrtf(wave_sum(ss));
  for (int q = 0; q < Q; ++q) {
    float d = 0.f;
    for (int i = lane; i < C; i += 64) d += cr[i] * text[(int64_t)q * C + i];
    d = wave_sum(d);
    if (lane == 0) out[b * JBU_QMAX + q] = d * inv;
  }
}
// One wave = 128 consecutive pixels, a lane owns pixels `lane` and `lane + 64`.  x rows are fetched coalesced (8 lanes x 16 B per pixel
// row piece) into a per-wave LDS tile [128 px][32 ch] and read back pixel-per-lane (row stride 36 floats: conflict-free b128), so every
// lane walks ITS pixels' channels while Geff[c][:] comes from LDS as a broadcast read shared by both pixels -- Q running dots + |x|^2 per
// pixel, no cross-lane reduction, and the logits of a query go out as 256 contiguous bytes per wave-instruction.
constexpr int PL_LD = 36, PL_PPL = 2;
template <int QP, typename XT>
__global__ __launch_bounds__(256, 2) void jbu_pixel_logits_kernel(const XT* __restrict__ x, const float* __restrict__ rowdot, int slots,
                                                                  const float* __restrict__ geff, const float* __restrict__ g0,
                                                                  const float* __restrict__ clsl, float lambda, int64_t pixels, int64_t P, int C,
                                                                  int Q, float* __restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) float pl_sm[];
  float* sG = pl_sm;                                              // [C][QP]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* tile = pl_sm + (size_t)C * QP + wave * (64 * PL_PPL) * PL_LD;   // [128][PL_LD], private to the wave
  for (int i = threadIdx.x; i < C * QP; i += 256) sG[i] = (i % QP) < Q ? geff[(i / QP) * JBU_QMAX + (i % QP)] : 0.f;
  __syncthreads();
  const int64_t pix0 = ((int64_t)blockIdx.x * 4 + wave) * (64 * PL_PPL);
  if (pix0 >= pixels) return;
  float acc[PL_PPL][QP], nx[PL_PPL];
#pragma unroll
  for (int e = 0; e < PL_PPL; ++e) {
    nx[e] = 0.f;
#pragma unroll
    for (int q = 0; q < QP; ++q) acc[e][q] = 0.f;
  }
  const int lp = lane >> 3, lc = lane & 7;                        // load role: pixel (within a group of 8) and 16-byte piece of its 128-byte row piece
  for (int c0 = 0; c0 < C; c0 += 32) {
    float4 v[8 * PL_PPL];
#pragma unroll
    for (int u = 0; u < 8 * PL_PPL; ++u) {
      int64_t pr = pix0 + u * 8 + lp;
      pr = pr < pixels ? pr : pixels - 1;
      if constexpr (sizeof(XT) == 4) v[u] = *reinterpret_cast<const float4*>(x + pr * C + c0 + 4 * lc);
      else {                                                      // bf16 rows (the throughput tail keeps x in bf16 only)
        const uint2 raw = *reinterpret_cast<const uint2*>(x + pr * C + c0 + 4 * lc);
        v[u] = make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                           __uint_as_float(raw.y & 0xffff0000u));
      }
    }
#pragma unroll
    for (int u = 0; u < 8 * PL_PPL; ++u) *reinterpret_cast<float4*>(tile + (u * 8 + lp) * PL_LD + 4 * lc) = v[u];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // LDS executes a wave's instructions in order: ordering only, no barrier
    __builtin_amdgcn_wave_barrier();
#pragma unroll 2
    for (int k = 0; k < 8; ++k) {
      float4 xv[PL_PPL];
#pragma unroll
      for (int e = 0; e < PL_PPL; ++e) {
        xv[e] = *reinterpret_cast<const float4*>(tile + (lane + 64 * e) * PL_LD + 4 * k);
        nx[e] += xv[e].x * xv[e].x + xv[e].y * xv[e].y + xv[e].z * xv[e].z + xv[e].w * xv[e].w;
      }
      const float* g = sG + (size_t)(c0 + 4 * k) * QP;            // same address on every lane: broadcast
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const float g0v = g[q], g1v = g[QP + q], g2v = g[2 * QP + q], g3v = g[3 * QP + q];
#pragma unroll
        for (int e = 0; e < PL_PPL; ++e) acc[e][q] += xv[e].x * g0v + xv[e].y * g1v + xv[e].z * g2v + xv[e].w * g3v;
      }
    }
    __builtin_amdgcn_wave_barrier();                              // the tile is rewritten in the next round
  }
#pragma unroll
  for (int e = 0; e < PL_PPL; ++e) {
    const int64_t pix = pix0 + lane + 64 * e;
    if (pix >= pixels) continue;
    float n2 = nx[e];
    for (int sidx = 0; sidx < slots; ++sidx) n2 += rowdot[pix * slots + sidx];        // fixed order: deterministic
    const float inv = 1.0f / sqrtf(n2);
    const int64_t b = pix / P, pp = pix % P;
#pragma unroll
    for (int q = 0; q < QP; ++q)
      if (q < Q) {
        float vv = (acc[e][q] + g0[q]) * inv;
        if (clsl) vv += lambda * clsl[b * JBU_QMAX + q];
        logits[(b * Q + q) * P + pp] = vv;
      }
  }
}

// The same product on the matrix pipe (round 3; Q <= 16, C % 32 == 0): logits = x . Geff is a [pixels, C] x [C, 16] GEMM whose A operand is
// the bf16 rows as they lie in HBM.  Geff is held as TWO bf16 planes (hi + lo = 16 significant bits, well below the 8 bits of x) in LDS,
// [16 q][C] each, so a B fragment is one ds_read_b128; a wave owns 64 consecutive pixels (4 row tiles of v_mfma_f32_16x16x32_bf16) per
// round and PLM_ROUNDS rounds, the A fragments come straight from global memory (16 B per lane: row = lane % 16, 8 consecutive channels), one
// k-step ahead.  |x|^2 rides along on the vector pipe from the same fragments.  The lane-per-pixel VALU form above ran at a third of the
// f32 vector peak (3.1 ms per 8 tiles of 592 x 592 at C = 768: 69 GFLOP of f32 FMAs); this one is bound by the 4.3 GB read of x.
constexpr int PLM_ROUNDS = 4;
__global__ __launch_bounds__(256, 2) void jbu_pixel_logits_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ rowdot, int slots,
                                                                       const float* __restrict__ geff, const float* __restrict__ g0,
                                                                       const float* __restrict__ clsl, float lambda, int64_t pixels, int64_t P, int C,
                                                                       int Q, float* __restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) char plm_sm[];
  bf16_t* sGh = reinterpret_cast<bf16_t*>(plm_sm);                        // [16][C + 8]  (+8: rows 16 bytes apart in the banks)
  const int ldg = C + 8;
  bf16_t* sGl = sGh + 16 * ldg;
  float* sN = reinterpret_cast<float*>(sGl + 16 * ldg);                   // [4 waves][64] 1 / |out| of the wave's pixels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16 * C; i += 256) {
    const int q = i / C, c = i % C;
    const float v = q < Q ? geff[c * JBU_QMAX + q] : 0.f;
    const bf16_t hi = f2bf(v);
    sGh[q * ldg + c] = hi;
    sGl[q * ldg + c] = f2bf(v - bf2f(hi));
  }
  __syncthreads();
  const int r = lane & 15, g = lane >> 4;
  float* myN = sN + wave * 64;
  for (int round = 0; round < PLM_ROUNDS; ++round) {
    const int64_t pix0 = (((int64_t)blockIdx.x * PLM_ROUNDS + round) * 4 + wave) * 64;
    if (pix0 >= pixels) return;                                           // wave-uniform; no barrier below
    const bf16_t* xr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int64_t pr = pix0 + t * 16 + r;
      pr = pr < pixels ? pr : pixels - 1;
      xr[t] = x + pr * C + 8 * g;
    }
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float nx[4] = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a_cur[4], a_nxt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a_cur[t] = *reinterpret_cast<const bf16x8*>(xr[t]);
    const int nk = C / 32;
    for (int ks = 0; ks < nk; ++ks) {
      const int kn = ks + 1 < nk ? ks + 1 : ks;
#pragma unroll
      for (int t = 0; t < 4; ++t) a_nxt[t] = *reinterpret_cast<const bf16x8*>(xr[t] + 32 * kn);
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sGh + r * ldg + 32 * ks + 8 * g);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(sGl + r * ldg + 32 * ks + 8 * g);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t] = mfma_16x16x32<false>(a_cur[t], bh, acc[t]);               // D[pixel i][query j]: lane = j + 16 (i / 4), 4 consecutive pixels
        acc[t] = mfma_16x16x32<false>(a_cur[t], bl, acc[t]);
        const uint4 raw = __builtin_bit_cast(uint4, a_cur[t]);
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
          nx[t] = __builtin_fmaf(lo, lo, nx[t]); nx[t] = __builtin_fmaf(hi, hi, nx[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) a_cur[t] = a_nxt[t];
    }
    // |x|^2 of pixel (t, r): the four 8-channel slices of every k-step sit in lanes r, r + 16, r + 32, r + 48
#pragma unroll
    for (int t = 0; t < 4; ++t) { nx[t] += __shfl_xor(nx[t], 16, 64); nx[t] += __shfl_xor(nx[t], 32, 64); }
    if (g == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int64_t pix = pix0 + t * 16 + r;
        pix = pix < pixels ? pix : pixels - 1;
        float n2 = nx[t];
        for (int sidx = 0; sidx < slots; ++sidx) n2 += rowdot[pix * slots + sidx];      // fixed order: deterministic
        myN[t * 16 + r] = 1.0f / sqrtf(n2);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // lane holds pixels pix0 + 16 t + 4 g .. + 3 of query r
    if (r < Q) {
      const float g0q = g0[r];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int64_t pix = pix0 + t * 16 + 4 * g;
        if (pix >= pixels) continue;
        const float4 inv = *reinterpret_cast<const float4*>(myN + t * 16 + 4 * g);
        const int64_t b = pix / P, pp = pix % P;
        float4 o = make_float4((acc[t][0] + g0q) * inv.x, (acc[t][1] + g0q) * inv.y, (acc[t][2] + g0q) * inv.z, (acc[t][3] + g0q) * inv.w);
        if (clsl) { const float cv = lambda * clsl[b * JBU_QMAX + r]; o.x += cv; o.y += cv; o.z += cv; o.w += cv; }
        float* dst = logits + (b * Q + r) * P + pp;
        if (pp + 3 < P && pix + 3 < pixels && (P & 3) == 0) *reinterpret_cast<float4*>(dst) = o;   // P % 4 == 0: the four pixels share the image and the store is aligned
        else {
          const float ov[4] = {o.x, o.y, o.z, o.w};
          for (int e = 0; e < 4; ++e) {
            const int64_t pe = pix + e;
            if (pe < pixels) logits[((pe / P) * Q + r) * P + pe % P] = ov[e] + ((clsl && pe / P != b) ? lambda * (clsl[(pe / P) * JBU_QMAX + r] - clsl[b * JBU_QMAX + r]) : 0.f);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();                                      // myN is rewritten in the next round
  }
}
}  // namespace sg

// sg_jbu_logits replaces, for a batch of tiles,  feats = upsampler(tokens, img) -> feats /= |feats| -> feats @ T^T (+ lambda * cls_logits)
// (segmentor.py:368-379) without materialising the [S^2, C] feature map (throughput mode).
extern "C" int sg_jbu_logits(sg_jbu* j, const float* source, const float* guidance, int B, int gh, int gw, int GH, int GW, int precision,
                             const float* text, int Q, const float* cls, float cls_token_lambda, float* logits, void* ws, size_t ws_bytes,
                             sg_stream st) {
  SG_REQUIRE(j && source && guidance && text && logits && ws, "sg_jbu_logits: null argument");
  SG_REQUIRE(Q >= 1 && Q <= JBU_QMAX, "sg_jbu_logits: 1 <= Q <= %d", JBU_QMAX);
  SG_REQUIRE(precision == SG_PREC_BF16 && j->C % 64 == 0 && j->C >= 512, "sg_jbu_logits: the fused tail is the bf16 throughput path (C %% 64 == 0, C >= 512); use sg_jbu_upsample + sg_cosine_logits otherwise");
  for (size_t i = 0; i < j->have.size(); ++i) if (!j->have[i]) return fail(SG_ERR_STATE, "sg_jbu_logits: upsampler weights incomplete");
  DeviceGuard dg(j->device);
  hipStream_t s = as_stream(st);
  JbuPlan p;
  const size_t need = jbu_plan(j, B, gh, gw, ws, false, p);
  if (need > ws_bytes) return fail(SG_ERR_STATE, "sg_jbu_logits: workspace %zu < required %zu", ws_bytes, need);
  const int C = j->C;
  const int64_t P = (int64_t)16 * gh * 16 * gw, pixels = (int64_t)B * P;
  SG_REQUIRE(pixels >= 1024 && pixels < (1ll << 31), "sg_jbu_logits: pixel count out of range");
  const float* x = nullptr;
  SG_TRY(jbu_stages(j, source, guidance, B, gh, gw, GH, GW, precision, p, &x, s, /*want_f32_x=*/false));
  const bf16_t* x16 = (const bf16_t*)p.x16;                 // the 16x features exist in bf16 only: the conv's 4 B/element f32 store and its two re-reads are gone
  hipLaunchKernelGGL(jbu_geff_kernel, dim3((unsigned)cdiv(C, 64), (unsigned)Q), dim3(256), 0, s, text, j->fin_w, j->fin_b, C, Q, p.geff, p.g0);
  SG_LAUNCH_CHECK();
  const bool use_cls = cls != nullptr && cls_token_lambda != 0.f;
  if (use_cls) { hipLaunchKernelGGL(jbu_cls_logits_kernel, dim3(B), dim3(64), 0, s, cls, text, C, Q, p.clsl); SG_LAUNCH_CHECK(); }
  GemmBf16Args g{};
  g.A = x16; g.lda = C; g.W = (const bf16_t*)j->fin_w16; g.ldw = C; g.bias = j->fin_b; g.residual = (const float*)x16; g.ldr = C;
  g.C = p.rowdot; g.ldc = C; g.c_is_bf16 = 0; g.M = (int)pixels; g.N = C; g.K = C; g.batch = 1; g.act = 0; g.alpha = 0.1f;
  g.rowdot = p.rowdot; g.rowdot_ld = C / 64; g.rowdot_res_bf16 = 1;
  SG_TRY(gemm_bf16(g, s));
  const int slots = C / 64;
  const unsigned grid = (unsigned)cdiv(pixels, 4 * 64 * PL_PPL);
#define SG_JBU_PIX(QP)                                                                                                        \
  do {                                                                                                                         \
    const size_t lds = ((size_t)C * QP + 4 * 64 * PL_PPL * PL_LD) * sizeof(float);                                             \
    if (lds > 48 * 1024) SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(jbu_pixel_logits_kernel<QP, bf16_t>), lds)); \
    hipLaunchKernelGGL((jbu_pixel_logits_kernel<QP, bf16_t>), dim3(grid), dim3(256), lds, s, x16, p.rowdot, slots, p.geff, p.g0,  \
                       use_cls ? p.clsl : nullptr, cls_token_lambda, pixels, P, C, Q, logits);                                 \
  } while (0)
  if (Q <= 16 && C % 32 == 0) {                              // matrix-pipe form
    const size_t lds = (size_t)2 * 16 * (C + 8) * sizeof(bf16_t) + 4 * 64 * sizeof(float);
    SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(jbu_pixel_logits_mfma_kernel), lds));
    hipLaunchKernelGGL(jbu_pixel_logits_mfma_kernel, dim3((unsigned)cdiv(pixels, 4 * 64 * PLM_ROUNDS)), dim3(256), lds, s, x16, p.rowdot, slots, p.geff, p.g0,
                       use_cls ? p.clsl : nullptr, cls_token_lambda, pixels, P, C, Q, logits);
  } else if (Q <= 8) SG_JBU_PIX(8); else if (Q <= 16) SG_JBU_PIX(16); else SG_JBU_PIX(32);
#undef SG_JBU_PIX
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_extract_tiles(const sg_tile_batch* t, int up_h, int up_w, float* out, sg_stream s) {
  SG_REQUIRE(t && out && t->scene && t->windows && t->n_tiles > 0, "sg_extract_tiles: bad arguments");
  SG_REQUIRE(up_h >= t->tile_h + t->pad_t && up_w >= t->tile_w + t->pad_l, "sg_extract_tiles: output smaller than the padded tile");
  const int64_t total = (int64_t)t->n_tiles * 3 * up_h * up_w;
  hipLaunchKernelGGL(extract_tiles_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(s), *t, up_h, up_w, out);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_global_debias(const float* tokens, const float* cls, int B, int n, int E, float factor, float* out, sg_stream s) {
  SG_REQUIRE(tokens && cls && out && B > 0 && n > 0 && E > 0 && B < 65536, "sg_global_debias: bad arguments");
  hipLaunchKernelGGL(global_debias_kernel, dim3((unsigned)cdiv(n, 4), (unsigned)B), dim3(256), 0, as_stream(s), tokens, cls, n, E, factor, out);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" int sg_adaptive_conv(const float* input, const float* filters, int B, int C, int h, int w, int d, float* out, sg_stream s) {
  SG_REQUIRE(input && filters && out, "sg_adaptive_conv: null pointer");
  SG_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0 && d > 0 && (int64_t)B * C < 65536 && cdiv(h, 4) < 65536, "sg_adaptive_conv: bad shape");
  hipLaunchKernelGGL(adaptive_conv_nchw_kernel, dim3((unsigned)cdiv(w, 64), (unsigned)cdiv(h, 4), (unsigned)(B * C)), dim3(256), 0,
                     as_stream(s), input, filters, C, h, w, d, out);
  SG_LAUNCH_CHECK();
  return SG_OK;
}
