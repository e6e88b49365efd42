// Training-free token refinements that act on the last-block output (before ln_post):
//   outlier suppression  (reference outlier_suppression.py:15-61 detection, :115-214 mean interpolation)
//   weak-token replacement (reference self_attention_enhancement.py:71-150, 247-324)
// The reference walks the outliers with Python loops and k*8 `.item()` host syncs per tile; here the
// whole step is three small launches per batch of tiles, no host round trip:
//   1. select_topk      : top-k of A[cls,i]/(A[i,i]+1e-8) (or the k smallest A[i,i]) per image, in LDS
//   2. refine_compute   : for every selected token, cosine to its 8 clamped neighbours, the softmax
//                         weights, the replacement row and the decontaminated neighbour rows, all read
//                         from the ORIGINAL map, written to scratch
//   3. refine_scatter   : the reference's write order resolved in parallel -- a neighbour cell takes the
//                         value of the LAST (outlier, neighbour) pair that targets it, cells equal to the
//                         outlier itself are skipped, outlier cells are written last (they always win).
#include "rowops.h"

namespace sg {

__constant__ int c_dy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
__constant__ int c_dx[8] = {-1, 0, 1, -1, 1, -1, 0, 1};

// ---- 1. top-k selection --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_topk_kernel(const float* __restrict__ attn_cls, const float* __restrict__ attn_diag,
                                                          int N, int k, int mode, int32_t* __restrict__ idx) {
  extern __shared__ float vals[];                       // n scores, larger = selected first
  __shared__ float red_v[4];
  __shared__ int red_i[4];
  __shared__ int winner;
  const int b = blockIdx.x, n = N - 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < n; i += 256) {
    const float d = attn_diag[(int64_t)b * N + 1 + i];
    vals[i] = mode == 0 ? attn_cls[(int64_t)b * N + 1 + i] / (d + 1e-8f) : -d;
  }
  __syncthreads();
  for (int r = 0; r < k; ++r) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = tid; i < n; i += 256) {
      const float v = vals[i];
      if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      float fv = red_v[0]; int fi = red_i[0];
      for (int w = 1; w < 4; ++w)
        if (red_v[w] > fv || (red_v[w] == fv && red_i[w] < fi)) { fv = red_v[w]; fi = red_i[w]; }
      if (fi == 0x7fffffff) fi = 0;                     // all -inf / NaN: degenerate input
      winner = fi;
      idx[(int64_t)b * k + r] = fi;
      vals[fi] = -INFINITY;
    }
    __syncthreads();
    (void)winner;
  }
}

int select_topk(const float* attn_cls, const float* attn_diag, int B, int N, int k, int mode, int32_t* idx, hipStream_t s) {
  const int n = N - 1;
  SG_REQUIRE(k >= 1 && k <= n, "select_topk: k=%d out of range for %d patches", k, n);
  SG_REQUIRE((size_t)n * 4 <= 160 * 1024 - 256, "select_topk: %d patches exceed LDS", n);
  const size_t lds = (size_t)n * sizeof(float);
  if (lds > 48 * 1024) SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(select_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(select_topk_kernel, dim3(B), dim3(256), lds, s, attn_cls, attn_diag, N, k, mode, idx);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// ---- 2. per-outlier neighbourhood arithmetic ------------------------------------------------------------
// scratch layout per image: [k][9][D] f32 -- row 0 = replacement, rows 1..8 = decontaminated neighbours
size_t refine_scratch_bytes(int B, int D, int k) { return (size_t)B * k * 9 * D * sizeof(float); }

__global__ __launch_bounds__(256) void refine_compute_kernel(const float* __restrict__ tokens, int64_t sb, int64_t st,
                                                             const int32_t* __restrict__ idx, int gh, int gw, int D, int k,
                                                             int decontaminate, float temp, float* __restrict__ scratch) {
  __shared__ float s_cos[8];
  __shared__ float s_w[8];
  __shared__ int s_tok[8];
  const int i = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cell = idx[(int64_t)b * k + i];
  const int cy = cell / gw, cx = cell % gw;
  const float* base = tokens + (int64_t)b * sb;
  const float* centre = base + (int64_t)(1 + cell) * st;
  if (tid < 8) {
    int ny = cy + c_dy[tid], nx = cx + c_dx[tid];
    ny = ny < 0 ? 0 : (ny > gh - 1 ? gh - 1 : ny);
    nx = nx < 0 ? 0 : (nx > gw - 1 ? gw - 1 : nx);
    s_tok[tid] = 1 + ny * gw + nx;
  }
  __syncthreads();
  for (int j = wave; j < 8; j += 4) {                    // F.normalize(eps=1e-12) then dot
    const float* nb = base + (int64_t)s_tok[j] * st;
    float dot = 0.f, nn = 0.f, cc = 0.f;
    for (int d = lane; d < D; d += 64) { const float x = nb[d], c = centre[d]; dot += x * c; nn += x * x; cc += c * c; }
    dot = wave_sum(dot); nn = wave_sum(nn); cc = wave_sum(cc);
    if (lane == 0) s_cos[j] = dot / (fmaxf(sqrtf(nn), 1e-12f) * fmaxf(sqrtf(cc), 1e-12f));
  }
  __syncthreads();
  if (tid == 0) {                                        // softmax(clamp(1 - cos, min 0)) over the 8 neighbours
    float w[8], mx = -INFINITY, sum = 0.f;
    for (int j = 0; j < 8; ++j) { w[j] = fmaxf(1.0f - s_cos[j], 0.f); mx = fmaxf(mx, w[j]); }
    for (int j = 0; j < 8; ++j) { w[j] = expf(w[j] - mx); sum += w[j]; }
    for (int j = 0; j < 8; ++j) s_w[j] = w[j] / sum;
  }
  __syncthreads();
  float* out = scratch + ((int64_t)b * k + i) * 9 * D;
  for (int d = tid; d < D; d += 256) {
    const float c = centre[d];
    float rep = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = base[(int64_t)s_tok[j] * st + d];
      rep += x * s_w[j];
      if (decontaminate) {
        const float sigma = fminf(fmaxf(s_cos[j] * temp, 0.f), 1.f);
        out[(int64_t)(1 + j) * D + d] = x - c * sigma;
      }
    }
    out[d] = rep;
  }
}

// ---- 3. ordered scatter ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void refine_scatter_kernel(float* __restrict__ tokens, int64_t sb, int64_t st,
                                                             const int32_t* __restrict__ idx, int gh, int gw, int D, int k,
                                                             int decontaminate, const float* __restrict__ scratch) {
  __shared__ int s_skip;
  const int p = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int32_t* id = idx + (int64_t)b * k;
  int target, src_row;
  if (tid == 0) s_skip = 0;
  __syncthreads();
  // the "who writes this cell last" test runs one candidate per thread (it was a serial loop of up to 8k iterations on thread 0)
  if (p < k) {                                           // replacement of outlier p: always written
    target = id[p]; src_row = p * 9;
    // duplicates cannot occur in a top-k; if they did, the LAST index wins (advanced-index assignment order)
    for (int q = p + 1 + tid; q < k; q += 256) if (id[q] == target) s_skip = 1;
  } else {
    if (!decontaminate) return;
    const int pp = p - k, i = pp >> 3, j = pp & 7;
    const int cell = id[i], cy = cell / gw, cx = cell % gw;
    int ny = cy + c_dy[j], nx = cx + c_dx[j];
    ny = ny < 0 ? 0 : (ny > gh - 1 ? gh - 1 : ny);
    nx = nx < 0 ? 0 : (nx > gw - 1 ? gw - 1 : nx);
    target = ny * gw + nx; src_row = i * 9 + 1 + j;
    if (tid == 0 && target == cell) s_skip = 1;            // clamped onto the outlier itself
    for (int q = tid; q < k; q += 256) if (id[q] == target) s_skip = 1;          // an outlier cell: replacement wins
    for (int q = pp + 1 + tid; q < k * 8; q += 256) {      // a later pair targets the same cell
      const int qi = q >> 3, qj = q & 7;
      const int qc = id[qi], qy = qc / gw, qx = qc % gw;
      int y = qy + c_dy[qj], x = qx + c_dx[qj];
      y = y < 0 ? 0 : (y > gh - 1 ? gh - 1 : y);
      x = x < 0 ? 0 : (x > gw - 1 ? gw - 1 : x);
      if (y * gw + x == target && (y * gw + x) != qc) s_skip = 1;
    }
  }
  __syncthreads();
  if (s_skip) return;
  const float* src = scratch + ((int64_t)b * k * 9 + src_row) * D;
  float* dst = tokens + (int64_t)b * sb + (int64_t)(1 + target) * st;
  for (int d = tid; d < D; d += 256) dst[d] = src[d];
}

int neighbour_refine(float* tokens, int64_t sb, int64_t st, const int32_t* idx, int B, int gh, int gw, int D, int k,
                     int decontaminate, float contamination_temp, void* scratch, hipStream_t s) {
  SG_REQUIRE(k >= 1 && k <= gh * gw && B < 65536, "neighbour_refine: bad k=%d", k);
  hipLaunchKernelGGL(refine_compute_kernel, dim3(k, B), dim3(256), 0, s, tokens, sb, st, idx, gh, gw, D, k, decontaminate,
                     contamination_temp, (float*)scratch);
  SG_LAUNCH_CHECK();
  hipLaunchKernelGGL(refine_scatter_kernel, dim3(decontaminate ? k * 9 : k, B), dim3(256), 0, s, tokens, sb, st, idx, gh, gw, D, k, decontaminate,
                     (const float*)scratch);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// ---- self-attention enhancement, mode='attention' (reference self_attention_enhancement.py:152-245) ----------------------------
// A = head-averaged attention of block L-2 [B,N,N].  boost the diagonal of the patch rows by clamp(thr - A[i,i], 0) * strength,
// L1-renormalise every row (sum + 1e-8), drop the CLS column (the reference feeds a zero CLS feature), then tokens' = A' . tokens.
__global__ __launch_bounds__(256) void head_mean_kernel(const float* __restrict__ probs, int H, int64_t NN, float* __restrict__ A) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= NN) return;
  float acc = 0.f;
  for (int h = 0; h < H; ++h) acc += probs[((int64_t)b * H + h) * NN + i];
  A[(int64_t)b * NN + i] = acc / (float)H;
}
__global__ __launch_bounds__(256) void attn_boost_kernel(float* __restrict__ A, int N, float strength, float thr) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (i >= N) return;
  float* r = A + ((int64_t)b * N + i) * N;
  const float boost = i >= 1 ? fmaxf(thr - r[i], 0.f) * strength : 0.f;
  float sum = 0.f;
  for (int j = lane; j < N; j += 64) sum += r[j] + (j == i ? boost : 0.f);
  const float inv = 1.0f / (wave_sum(sum) + 1e-8f);
  for (int j = lane; j < N; j += 64) r[j] = j == 0 ? 0.f : (r[j] + (j == i ? boost : 0.f)) * inv;
}
// ---- attention-map layer fusion (reference open_clip/transformer.py:647-690) ------------------------------------------------------
// row 0 and the diagonal of the fused map [B,N,N] (what detect_outliers_by_attention consumes)
__global__ void fusion_row_diag_kernel(const float* __restrict__ A, int N, float* __restrict__ a_cls, float* __restrict__ a_diag) {
  const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const float* Ab = A + (int64_t)b * N * N;
  a_cls[(int64_t)b * N + j] = Ab[j];
  a_diag[(int64_t)b * N + j] = Ab[(int64_t)j * (N + 1)];
}
// zero the columns 1 + idx[b, :] of A[b], then L1-normalise every row: A / (sum + 1e-8)   (:668-675); one wave per row, in place
__global__ __launch_bounds__(256) void fusion_mask_normalize_kernel(float* __restrict__ A, const int32_t* __restrict__ idx, int N, int k) {
  extern __shared__ int32_t s_idx[];
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < k; i += 256) s_idx[i] = idx[(int64_t)b * k + i] + 1;
  __syncthreads();
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float* r = A + ((int64_t)b * N + row) * N;
  float sum = 0.f;
  for (int j = lane; j < N; j += 64) {
    bool masked = false;
    for (int i = 0; i < k; ++i) masked = masked || (s_idx[i] == j);
    const float v = masked ? 0.f : r[j];
    r[j] = v;
    sum += v;
  }
  const float inv = 1.0f / (wave_sum(sum) + 1e-8f);
  for (int j = lane; j < N; j += 64) r[j] *= inv;
}
int fusion_row_diag(const float* A, int B, int N, float* a_cls, float* a_diag, hipStream_t s) {
  hipLaunchKernelGGL(fusion_row_diag_kernel, dim3((unsigned)cdiv(N, 256), (unsigned)B), dim3(256), 0, s, A, N, a_cls, a_diag);
  SG_LAUNCH_CHECK();
  return SG_OK;
}
int fusion_mask_normalize(float* A, const int32_t* idx, int B, int N, int k, hipStream_t s) {
  hipLaunchKernelGGL(fusion_mask_normalize_kernel, dim3((unsigned)cdiv(N, 4), (unsigned)B), dim3(256), (size_t)(k > 0 ? k : 1) * 4, s, A, idx, N, k);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

int head_mean(const float* probs, int B, int H, int N, float* A, hipStream_t s) {
  const int64_t NN = (int64_t)N * N;
  hipLaunchKernelGGL(head_mean_kernel, dim3((unsigned)cdiv(NN, 256), (unsigned)B), dim3(256), 0, s, probs, H, NN, A);
  SG_LAUNCH_CHECK();
  return SG_OK;
}
// tokens [B,N,D] f32 (sb, st strides); A [B,N,N] (destroyed); tmp [B,N,D] f32
int attn_mode_enhance(float* tokens, int64_t sb, int64_t st, float* A, int B, int N, int D, float strength, float threshold, float* tmp,
                      hipStream_t s) {
  hipLaunchKernelGGL(attn_boost_kernel, dim3((unsigned)cdiv(N, 4), (unsigned)B), dim3(256), 0, s, A, N, strength, threshold);
  SG_LAUNCH_CHECK();
  GemmF32Args g{};
  g.A = A; g.lda = N; g.sAo = (int64_t)N * N; g.B = tokens; g.sbk = st; g.sbn = 1; g.sBo = sb;
  g.C = tmp; g.ldc = D; g.sCo = (int64_t)N * D; g.M = N; g.N = D; g.K = N; g.batch = B; g.inner = 1; g.act = 0; g.alpha = 1.f;
  SG_TRY(gemm_f32(g, s));
  // patch rows only (the CLS feature is passed through unchanged, transformer.py:703,717)
  SG_HIP(hipMemcpy2DAsync(tokens + st, (size_t)sb * 4, tmp + D, (size_t)N * D * 4, (size_t)(N - 1) * D * 4, B, hipMemcpyDeviceToDevice, s));
  return SG_OK;
}

// ---- cross-tile boundary fusion (reference cross_tile_fusion.py:24-320; unwired there) -----------------------------
// Tiles arrive in raster order, so only the 'top' and 'left' directions ever find a cached neighbour.  The reference's
// strip extraction aliases (row strips are views, column strips are copies), which fixes the data flow restated in
// oracle/refine.py::CrossTileFusionOracle:
//   left  : fuse(ORIGINAL left columns of the tile, ORIGINAL right columns of the left neighbour)   -> columns [0,bw)
//   top   : fuse(ORIGINAL top rows of the tile, FINAL bottom rows of the upper neighbour)            -> rows [0,bw)
//           (final = original with that neighbour's own left result in columns [0,bw)); the left result wins the corner.
// Neither depends on another tile's top result (gh >= 2 bw), so every strip of a scene is fused in parallel:
// one workgroup per (tile, direction) writes its strip to scratch, a second launch scatters.
constexpr int CTF_MAX_STRIP = 128;
// Under tile sharding the neighbour may live on another rank, so neighbour strips always come from PACKED
// strip buffers indexed by the GLOBAL tile id ([T, S, C]); the tokens / results are local ([n_local, ...], first tile = tile0).
//   pack 0: ORIGINAL right columns  (gh x bw)          pack 1: FINAL bottom rows (bw x gw; columns [0,bw) from the tile's left result)
__global__ void ctf_pack_kernel(const float* __restrict__ tokens, const float* __restrict__ left_result, int tile0, int wg, int gh, int gw,
                                int C, int bw, int which, float* __restrict__ out) {
  const int i_loc = blockIdx.y;
  const int S = which == 0 ? gh * bw : bw * gw;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)S * C) return;
  const int c = (int)(i % C), e = (int)(i / C);
  const int n = gh * gw;
  const float* t = tokens + (int64_t)i_loc * n * C;
  float v;
  if (which == 0) v = t[(int64_t)((e / bw) * gw + (gw - bw) + (e % bw)) * C + c];
  else {
    const int col = e % gw, row = gh - bw + e / gw;
    if (col < bw && left_result && ((tile0 + i_loc) % wg) > 0) v = left_result[((int64_t)i_loc * gh * bw + row * bw + col) * C + c];
    else v = t[(int64_t)(row * gw + col) * C + c];
  }
  out[(int64_t)i_loc * S * C + i] = v;
}

__global__ __launch_bounds__(256) void ctf_fuse_kernel(const float* __restrict__ tokens, const float* __restrict__ nbr_strips, int tile0, int wg,
                                                       int gh, int gw, int C, int bw, int mode, float strength, int pass,
                                                       float* __restrict__ result) {
  extern __shared__ float sm[];
  const int i_loc = blockIdx.x, tile = tile0 + i_loc, hi = tile / wg, wi = tile % wg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = gh * gw;
  const bool is_left = pass == 0;
  if (is_left ? (wi == 0) : (hi == 0)) return;
  const int S = is_left ? gh * bw : bw * gw;                  // both strips of a pair have the same length
  float* sim = sm;                                            // [S][L2]   L2 = S (weighted) or 2S (attention)
  const int L2 = mode == 0 ? S : 2 * S;
  float* rn_cur = sm + S * L2;                                // [S] norms / row scalars
  float* rn_nbr = rn_cur + S;
  const float* cur_t = tokens + (int64_t)i_loc * n * C;
  const int nb_tile = is_left ? tile - 1 : tile - wg;
  const float* nbr_s = nbr_strips + (int64_t)nb_tile * S * C;
  auto cur_idx = [&](int e) { return is_left ? (e / bw) * gw + (e % bw) : e; };                        // left cols | top rows
  auto nbr_row = [&](int e) -> const float* { return nbr_s + (int64_t)e * C; };
  const float eps = 1e-6f;
  for (int e = wave; e < S; e += 4) {                         // norms (weighted mode)
    const float* x = cur_t + (int64_t)cur_idx(e) * C; const float* y = nbr_row(e);
    float a = 0.f, b2 = 0.f;
    for (int c = lane; c < C; c += 64) { a += x[c] * x[c]; b2 += y[c] * y[c]; }
    a = wave_sum(a); b2 = wave_sum(b2);
    if (lane == 0) { rn_cur[e] = sqrtf(a) + eps; rn_nbr[e] = sqrtf(b2) + eps; }
  }
  __syncthreads();
  for (int p = wave; p < S * L2; p += 4) {                    // similarity / score matrix, one wave per entry
    const int i = p / L2, j = p % L2;
    const float* x = cur_t + (int64_t)cur_idx(i) * C;
    const float* y = (mode == 1 && j < S) ? cur_t + (int64_t)cur_idx(j) * C : nbr_row(mode == 1 ? j - S : j);
    float d = 0.f;
    for (int c = lane; c < C; c += 64) d += x[c] * y[c];
    d = wave_sum(d);
    if (lane == 0) sim[p] = mode == 0 ? d / (rn_cur[i] * rn_nbr[j]) : d / sqrtf((float)C);
  }
  __syncthreads();
  for (int i = tid; i < S; i += 256) {                        // per-row weights
    float* r = sim + i * L2;
    if (mode == 0) {
      float mean = 0.f;
      for (int j = 0; j < L2; ++j) mean += r[j];
      mean /= (float)L2;
      float var = 0.f;
      for (int j = 0; j < L2; ++j) var += (r[j] - mean) * (r[j] - mean);
      const float thr = mean + sqrtf(var / (float)(L2 - 1));  // torch.std: unbiased
      float msum = 0.f, wsum = 0.f;
      for (int j = 0; j < L2; ++j) { const float mg = fmaxf(r[j] - thr, 0.f); msum += mg; r[j] = mg * mg; wsum += mg * mg; }
      wsum += eps;
      for (int j = 0; j < L2; ++j) r[j] /= wsum;
      rn_cur[i] = strength * fminf(fmaxf(msum / (float)L2, 0.f), 1.f);       // blend factor of the row
    } else {
      float mx = -INFINITY, sum = 0.f;
      for (int j = 0; j < L2; ++j) mx = fmaxf(mx, r[j]);
      for (int j = 0; j < L2; ++j) { r[j] = expf(r[j] - mx); sum += r[j]; }
      for (int j = 0; j < L2; ++j) r[j] /= sum;
      rn_cur[i] = strength;
    }
  }
  __syncthreads();
  float* dst = result + (int64_t)i_loc * S * C;
  for (int p = tid; p < S * C; p += 256) {                    // blended strip
    const int i = p / C, c = p % C;
    const float* w = sim + i * L2;
    float agg = 0.f;
    if (mode == 1) for (int j = 0; j < S; ++j) agg += w[j] * cur_t[(int64_t)cur_idx(j) * C + c];
    for (int j = 0; j < S; ++j) agg += w[(mode == 1 ? S : 0) + j] * nbr_row(j)[c];
    const float sfac = rn_cur[i];
    dst[p] = cur_t[(int64_t)cur_idx(i) * C + c] * (1.f - sfac) + agg * sfac;
  }
}

__global__ void ctf_apply_kernel(float* __restrict__ tokens, const float* __restrict__ left_result, const float* __restrict__ top_result,
                                 int tile0, int wg, int gh, int gw, int C, int bw) {
  const int i_loc = blockIdx.y, tile = tile0 + i_loc, hi = tile / wg, wi = tile % wg;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = gh * gw;
  if (i >= (int64_t)n * C) return;
  const int c = (int)(i % C), p = (int)(i / C), row = p / gw, col = p % gw;
  if (wi > 0 && col < bw) tokens[(int64_t)i_loc * n * C + i] = left_result[((int64_t)i_loc * gh * bw + row * bw + col) * C + c];
  else if (hi > 0 && row < bw) tokens[(int64_t)i_loc * n * C + i] = top_result[((int64_t)i_loc * bw * gw + row * gw + col) * C + c];
}

static int ctf_check(int gh, int gw, int C, int bw, const char* who) {
  SG_REQUIRE(gh > 0 && gw > 0 && C > 0 && bw > 0, "%s: bad shape", who);
  SG_REQUIRE(gh >= 2 * bw && gw >= 2 * bw, "%s: patch grid %dx%d too small for boundary width %d", who, gh, gw, bw);
  SG_REQUIRE(gh * bw <= CTF_MAX_STRIP && gw * bw <= CTF_MAX_STRIP, "%s: strips longer than %d tokens", who, CTF_MAX_STRIP);
  return SG_OK;
}
static int ctf_pack(const float* tokens, const float* left_result, int n_local, int tile0, int wg, int gh, int gw, int C, int bw, int which,
                    float* out, hipStream_t s) {
  const int S = which == 0 ? gh * bw : bw * gw;
  hipLaunchKernelGGL(ctf_pack_kernel, dim3((unsigned)cdiv((int64_t)S * C, 256), (unsigned)n_local), dim3(256), 0, s, tokens, left_result, tile0, wg,
                     gh, gw, C, bw, which, out);
  SG_LAUNCH_CHECK();
  return SG_OK;
}
static int ctf_fuse(const float* tokens, const float* nbr_strips, int n_local, int tile0, int wg, int gh, int gw, int C, int bw, int mode,
                    float strength, int pass, float* result, hipStream_t s) {
  const int S = pass == 0 ? gh * bw : bw * gw;
  const size_t lds = ((size_t)S * (mode == 0 ? S : 2 * S) + 2 * S) * sizeof(float);
  if (lds > 48 * 1024) SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ctf_fuse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(ctf_fuse_kernel, dim3(n_local), dim3(256), lds, s, tokens, nbr_strips, tile0, wg, gh, gw, C, bw, mode, strength, pass, result);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

}  // namespace sg

using namespace sg;

extern "C" size_t sg_cross_tile_scratch_bytes(int T, int gh, int gw, int C, int bw) {
  const size_t sl = (size_t)T * gh * bw * C, st = (size_t)T * bw * gw * C;
  return (sl + st + (sl > st ? sl : st)) * sizeof(float) + 512;
}

extern "C" int sg_cross_tile_fusion(float* tokens, int hg, int wg, int gh, int gw, int C, int bw, int mode, float strength, void* scratch,
                                    sg_stream st) {
  SG_REQUIRE(tokens && scratch, "sg_cross_tile_fusion: null pointer");
  SG_REQUIRE(hg > 0 && wg > 0, "sg_cross_tile_fusion: bad tile grid");
  SG_TRY(ctf_check(gh, gw, C, bw, "sg_cross_tile_fusion"));
  SG_REQUIRE(mode == 0 || mode == 1, "sg_cross_tile_fusion: mode must be 0 (weighted) or 1 (attention)");
  hipStream_t s = as_stream(st);
  const int T = hg * wg;
  float* left = reinterpret_cast<float*>(scratch);
  float* top = left + (size_t)T * gh * bw * C;
  float* strips = top + (size_t)T * bw * gw * C;
  SG_TRY(ctf_pack(tokens, nullptr, T, 0, wg, gh, gw, C, bw, 0, strips, s));
  SG_TRY(ctf_fuse(tokens, strips, T, 0, wg, gh, gw, C, bw, mode, strength, 0, left, s));
  SG_TRY(ctf_pack(tokens, left, T, 0, wg, gh, gw, C, bw, 1, strips, s));
  SG_TRY(ctf_fuse(tokens, strips, T, 0, wg, gh, gw, C, bw, mode, strength, 1, top, s));
  hipLaunchKernelGGL(ctf_apply_kernel, dim3((unsigned)cdiv((int64_t)gh * gw * C, 256), (unsigned)T), dim3(256), 0, s, tokens, left, top, 0, wg, gh, gw, C, bw);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// The three steps of the fusion for a rank that holds tiles [tile0, tile0 + n_local) of the raster list; the caller exchanges the
// packed strips between the steps (two small all-gathers):
//   pack(0) -> gather -> fuse(0) -> pack(1, left_result) -> gather -> fuse(1) -> apply
extern "C" int sg_cross_tile_pack(const float* tokens, const float* left_result, int n_local, int tile0, int wg, int gh, int gw, int C, int bw,
                                  int which, float* out, sg_stream st) {
  SG_REQUIRE(tokens && out && n_local > 0 && tile0 >= 0 && wg > 0 && (which == 0 || which == 1), "sg_cross_tile_pack: bad arguments");
  SG_TRY(ctf_check(gh, gw, C, bw, "sg_cross_tile_pack"));
  return ctf_pack(tokens, left_result, n_local, tile0, wg, gh, gw, C, bw, which, out, as_stream(st));
}
extern "C" int sg_cross_tile_fuse(const float* tokens, const float* nbr_strips, int n_local, int tile0, int wg, int gh, int gw, int C, int bw,
                                  int mode, float strength, int pass, float* result, sg_stream st) {
  SG_REQUIRE(tokens && nbr_strips && result && n_local > 0 && tile0 >= 0 && wg > 0 && (pass == 0 || pass == 1) && (mode == 0 || mode == 1),
             "sg_cross_tile_fuse: bad arguments");
  SG_TRY(ctf_check(gh, gw, C, bw, "sg_cross_tile_fuse"));
  return ctf_fuse(tokens, nbr_strips, n_local, tile0, wg, gh, gw, C, bw, mode, strength, pass, result, as_stream(st));
}
extern "C" int sg_cross_tile_apply(float* tokens, const float* left_result, const float* top_result, int n_local, int tile0, int wg, int gh,
                                   int gw, int C, int bw, sg_stream st) {
  SG_REQUIRE(tokens && left_result && top_result && n_local > 0 && tile0 >= 0 && wg > 0, "sg_cross_tile_apply: bad arguments");
  SG_TRY(ctf_check(gh, gw, C, bw, "sg_cross_tile_apply"));
  hipLaunchKernelGGL(ctf_apply_kernel, dim3((unsigned)cdiv((int64_t)gh * gw * C, 256), (unsigned)n_local), dim3(256), 0, as_stream(st), tokens,
                     left_result, top_result, tile0, wg, gh, gw, C, bw);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

namespace sg {

}  // namespace sg
