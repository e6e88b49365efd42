// C ABI of libsegearth_hip.so: context / weight packing / the ViT forward orchestration.
// Everything here is host code issuing asynchronous launches on the caller's stream; the only
// allocations happen in sg_create (one arena sized from the architecture descriptor).
#include <string>
#include <vector>
#include <mutex>
#include <utility>
#include <math.h>
#include "rowops.h"

namespace sg {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
  return code;
}

// ---- live kernel timing ------------------------------------------------------------------------------------
struct ProfState {
  bool on = false;
  int cap = 0;
  std::vector<hipEvent_t> start[PROF_NCAT], stop[PROF_NCAT];
  int used[PROF_NCAT] = {};
  double work[PROF_NCAT] = {};
  int64_t dropped[PROF_NCAT] = {};
};
// measurement state belongs to the CALLING THREAD (bench.py enables, launches and reads on one thread): two host threads driving
// two contexts never share it
static thread_local ProfState g_prof;

static std::mutex g_dev_mu;                                        // guards the two per-device caches below
struct LdsOptIn { int dev; const void* kernel; size_t bytes; };
static std::vector<LdsOptIn> g_lds_done;                           // (device, kernel) -> the dynamic-LDS size already opted in to
static std::vector<std::pair<int, int>> g_cu_count;               // (device, compute units)
int ensure_dynamic_lds(const void* kernel, size_t bytes) {
  int dev = 0;
  SG_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_dev_mu);
  LdsOptIn* hit = nullptr;
  for (auto& e : g_lds_done) if (e.dev == dev && e.kernel == kernel) hit = &e;
  if (hit && hit->bytes >= bytes) return SG_OK;                   // a LARGER request than the cached one raises the limit again (jbu_pixel_logits_kernel: size depends on C)
  SG_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  if (hit) hit->bytes = bytes; else g_lds_done.push_back(LdsOptIn{dev, kernel, bytes});
  return SG_OK;
}
int device_cu_count() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lk(g_dev_mu);
  for (const auto& e : g_cu_count) if (e.first == dev) return e.second;
  int n = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
  g_cu_count.emplace_back(dev, n);
  return n;
}
bool prof_on() { return g_prof.on; }
void prof_begin(int cat, double work, hipStream_t s) {
  if (!g_prof.on) return;
  if (g_prof.used[cat] >= g_prof.cap) { g_prof.dropped[cat]++; return; }
  g_prof.work[cat] += work;
  (void)hipEventRecord(g_prof.start[cat][g_prof.used[cat]], s);
}
void prof_end(int cat, hipStream_t s) {
  if (!g_prof.on || g_prof.used[cat] >= g_prof.cap) return;
  (void)hipEventRecord(g_prof.stop[cat][g_prof.used[cat]], s);
  g_prof.used[cat]++;
}

// ---- a bump allocator over caller-provided (or arena) memory ------------------------------------------------
struct Bump {
  char* base; size_t off, cap; bool dry;
  Bump(void* p, size_t c, bool d) : base((char*)p), off(0), cap(c), dry(d) {}
  void* take(size_t bytes) {
    off = align_up(off, 256);
    void* p = dry ? nullptr : (void*)(base + off);
    off += bytes;
    return p;
  }
  template <typename T> T* get(size_t count) { return reinterpret_cast<T*>(take(count * sizeof(T))); }
};

struct LayerW {
  void *w_qkv, *w_out, *w_fc, *w_proj;                 // packed [N_out, K] in the compute dtype
  uint8_t *w_qkv8, *w_fc8, *w_proj8;                   // SG_PREC_FP8: e4m3 copies + per-output-channel scales
  float *s_qkv, *s_fc, *s_proj;
  float *b_qkv, *b_out, *b_fc, *b_proj, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  // LayerNorm folded into the GEMMs (2-byte modes without fp8 linears): W' = gamma o W, c = row sums of W', b' = b + W.beta  (rowops.h)
  void *w_qkv_f = nullptr, *w_fc_f = nullptr;
  float *c_qkv = nullptr, *c_fc = nullptr, *bf_qkv = nullptr, *bf_fc = nullptr;
  float *stage_qkv = nullptr, *stage_fc = nullptr;     // f32 copies of the two weights, kept from sg_vit_set_tensor to sg_vit_finalize only
  bool folded = false;
};

}  // namespace sg

using namespace sg;

struct sg_context {
  sg_vit_desc d;
  int device;
  int Kpatch, Kpad;
  int hk;                                              // HalfKind of the GEMM / attention operands: 0 = f32 (parity mode), 1 = bf16, 2 = f16
  bool fp8;                                            // SG_PREC_FP8: bf16 machinery + fp8 linears in the ordinary blocks
  size_t esz;                                          // bytes per element of the compute dtype
  void* arena; size_t arena_bytes;
  std::vector<LayerW> layers;
  void* w_patch;                                       // [D, Kpad]
  void* w_projT;                                       // [E, D]
  float *cls_emb, *pos, *lnpre_g, *lnpre_b, *lnpost_g, *lnpost_b;
  std::vector<uint8_t> have;                           // which tensors have arrived
  int n_expected;
  bool finalized;
};

namespace sg {

static int expected_tensors(const sg_vit_desc& d) { return 8 + 12 * d.layers; }

// ---- variants of the last-block attention: which (Q,K) terms, summed or not ---------------------------------------
struct Variant { int n_terms, sum_scores, qsel[3], ksel[3]; float scale_mul; int resoftmax; int gauss; };   // sel: 0=q 1=k 2=v; gauss: 1 plain omega, 2 omega*|q||k|*scale
static bool variant_of(int model_type, Variant& v) {
  switch (model_type) {
    case SG_VANILLA:      v = {1, 0, {0, 0, 0}, {1, 0, 0}, 1.f, 0, 0}; return true;
    case SG_CLEARCLIP:    v = {1, 0, {0, 0, 0}, {0, 0, 0}, 1.f, 0, 0}; return true;
    case SG_SCLIP:        v = {2, 0, {0, 1, 0}, {0, 1, 0}, 1.f, 0, 0}; return true;
    case SG_SEGEARTH:     v = {3, 0, {0, 1, 2}, {0, 1, 2}, 1.f, 0, 0}; return true;
    case SG_SFP:          v = {2, 1, {0, 1, 0}, {0, 1, 0}, 0.5f, 0, 0}; return true;
    case SG_EXPERIMENTAL: v = {2, 1, {1, 0, 0}, {1, 0, 0}, 1.f, 1, 0}; return true;
    // Gaussian-window variants (transformer.py:909-932): the similarity map is NOT applied on these paths in the reference
    case SG_NACLIP:       v = {1, 0, {1, 0, 0}, {1, 0, 0}, 1.f, 0, 1}; return true;      // k k^T * scale + omega
    case SG_NONLY:        v = {1, 0, {0, 0, 0}, {1, 0, 0}, 0.f, 0, 2}; return true;      // omega * scale * |q_i| |k_j| only
    case SG_GAV:          v = {1, 0, {0, 0, 0}, {1, 0, 0}, 1.f, 0, 2}; return true;      // q k^T * scale + omega * scale * |q_i| |k_j|
    default: return false;
  }
}

// Generic multi-term attention in either precision.  Element (b, t, h, d) of a Q/K operand lives at
// p + b*sb + t*st + h*dh + d (elements of the compute dtype); V has its own strides.
// f32: materialised scores/probs in `scores`/`probs` ([B*H,N,N] each).  bf16: fused kernel.
struct AttnBuffers { float* scores; float* probs; float* lse; float* lse1; float* omega; float* qnorm; float* knorm; };
struct AttnSpec {
  const void* q[3]; const void* k[3]; int64_t sb, st;
  const void* v; int64_t v_sb, v_st;
  int n_terms, sum_scores, resoftmax, causal;
  float scale; const float* scale_per_image;
  const float* bias; float bias_w; int64_t bias_bstride; const float* bias_rn; const float* bias_cn;
  float out_scale;
  void* ctx; int64_t ctx_sb, ctx_st;
  bool want_lse;
};

static int attn_generic(int bf16, const AttnSpec& sp, int B, int N, int H, int dh, const AttnBuffers& buf, hipStream_t s) {
  if (bf16) {
    AttnArgs a{};
    for (int t = 0; t < sp.n_terms; ++t) { a.q[t] = (const bf16_t*)sp.q[t]; a.k[t] = (const bf16_t*)sp.k[t]; }
    a.v = (const bf16_t*)sp.v; a.sb = sp.sb; a.st = sp.st; a.v_sb = sp.v_sb; a.v_st = sp.v_st;
    a.n_terms = sp.n_terms; a.sum_scores = sp.sum_scores; a.causal = sp.causal;
    a.B = B; a.N = N; a.H = H; a.dh = dh; a.scale = sp.scale; a.scale_per_image = sp.scale_per_image;
    a.out_scale = sp.out_scale; a.ctx_sb = sp.ctx_sb; a.ctx_st = sp.ctx_st; a.f16 = bf16 == HK_F16; a.h2 = bf16 == HK_F16X2;
    if (a.h2) { a.sb *= 2; a.st *= 2; a.v_sb *= 2; a.v_st *= 2; a.ctx_sb *= 2; a.ctx_st *= 2; }   // two-plane f16: the kernel addresses in f16 units
    if (sp.resoftmax) {
      AttnArgs p = a; p.ctx = nullptr; p.bias = nullptr; p.lse_out = buf.lse1; p.resoftmax = 0;
      SG_TRY(attention_bf16(p, s));
      a.resoftmax = 1; a.lse_in = buf.lse1;
    }
    a.bias = sp.bias; a.bias_w = sp.bias_w; a.bias_bstride = sp.bias_bstride; a.bias_rn = sp.bias_rn; a.bias_cn = sp.bias_cn; a.ctx = (bf16_t*)sp.ctx; a.lse_out = sp.want_lse ? buf.lse : nullptr;
    return attention_bf16(a, s);
  }
  const int64_t NN = (int64_t)N * N;
  auto scores_of = [&](int t, bool accumulate) {
    GemmF32Args g{};
    g.A = (const float*)sp.q[t]; g.lda = sp.st; g.sAo = sp.sb; g.sAi = dh;
    g.B = (const float*)sp.k[t]; g.sbk = 1; g.sbn = sp.st; g.sBo = sp.sb; g.sBi = dh;
    g.C = buf.scores; g.ldc = N; g.sCo = (int64_t)H * NN; g.sCi = NN;
    g.residual = accumulate ? buf.scores : nullptr; g.ldr = N;
    g.M = N; g.N = N; g.K = dh; g.batch = B * H; g.inner = H; g.act = 0; g.alpha = 1.f;
    return gemm_f32(g, s);
  };
  const int64_t rows = (int64_t)B * H * N;
  if (sp.sum_scores) {
    for (int t = 0; t < sp.n_terms; ++t) SG_TRY(scores_of(t, t > 0));
    SG_TRY(softmax_rows(buf.scores, N, rows, N, H, sp.scale_per_image, sp.scale, sp.bias, sp.bias_w, sp.bias_bstride, sp.bias_rn, sp.bias_cn, sp.resoftmax ? 1 : 0, 0,
                        buf.probs, sp.want_lse ? buf.lse : nullptr, s));
  } else {
    for (int t = 0; t < sp.n_terms; ++t) {
      SG_TRY(scores_of(t, false));
      SG_TRY(softmax_rows(buf.scores, N, rows, N, H, sp.scale_per_image, sp.scale, sp.bias, sp.bias_w, sp.bias_bstride, sp.bias_rn, sp.bias_cn, 0, t > 0, buf.probs,
                          (sp.want_lse && t == 0) ? buf.lse : nullptr, s, sp.causal));
    }
  }
  GemmF32Args g{};
  g.A = buf.probs; g.lda = N; g.sAo = (int64_t)H * NN; g.sAi = NN;
  g.B = (const float*)sp.v; g.sbk = sp.v_st; g.sbn = 1; g.sBo = sp.v_sb; g.sBi = dh;
  g.C = (float*)sp.ctx; g.ldc = sp.ctx_st; g.sCo = sp.ctx_sb; g.sCi = dh;
  g.M = N; g.N = dh; g.K = N; g.batch = B * H; g.inner = H; g.act = 0; g.alpha = sp.out_scale;
  return gemm_f32(g, s);
}

// Attention over packed qkv [B,N,3D] (compute dtype) -> ctx [B,N,D] (compute dtype).
static int run_attention(int bf16, const void* qkv, int B, int N, int D, int H, int model_type, const float* sim, float sim_w,
                         const float* scale_per_image, void* ctx, bool want_lse, const AttnBuffers& buf, hipStream_t s, bool causal = false) {
  const int dh = D / H;
  Variant v;
  if (!variant_of(model_type, v)) return fail(SG_ERR_INVALID, "attention variant %d is not built (NACLIP / NOnly / GAV: SURVEY.md §8f rank 3)", model_type);
  const size_t e = hk_esz(bf16);
  AttnSpec sp{};
  for (int t = 0; t < v.n_terms; ++t) { sp.q[t] = (const char*)qkv + (size_t)v.qsel[t] * D * e; sp.k[t] = (const char*)qkv + (size_t)v.ksel[t] * D * e; }
  sp.v = (const char*)qkv + (size_t)2 * D * e;
  sp.st = sp.v_st = 3 * (int64_t)D; sp.sb = sp.v_sb = (int64_t)N * 3 * D;
  sp.n_terms = v.n_terms; sp.sum_scores = v.sum_scores; sp.resoftmax = v.resoftmax; sp.causal = causal ? 1 : 0;
  sp.scale = v.scale_mul / sqrtf((float)dh); sp.scale_per_image = scale_per_image;
  sp.bias = sim; sp.bias_w = sim_w; sp.bias_bstride = (int64_t)(N - 1) * (N - 1); sp.out_scale = 1.f;
  if (v.gauss) {
    const int gside = (int)lroundf(sqrtf((float)(N - 1)));                 // the reference assumes a square grid here (transformer.py:912)
    SG_REQUIRE(gside * gside == N - 1, "Gaussian-window attention needs a square patch grid (N-1 = %d)", N - 1);
    SG_REQUIRE(buf.omega && buf.qnorm && buf.knorm, "Gaussian-window attention: scratch missing");
    SG_TRY(gaussian_bias(gside, gside, 1.0f, buf.omega, s));
    sp.bias = buf.omega; sp.bias_bstride = 0; sp.bias_w = 1.f;
    if (v.gauss == 2) {
      SG_TRY(head_norms(sp.q[0], bf16, sp.sb, sp.st, B, N, H, dh, buf.qnorm, s));
      SG_TRY(head_norms(sp.k[0], bf16, sp.sb, sp.st, B, N, H, dh, buf.knorm, s));
      sp.bias_rn = buf.qnorm; sp.bias_cn = buf.knorm; sp.bias_w = 1.0f / sqrtf((float)dh);
    }
  }
  sp.ctx = ctx; sp.ctx_sb = (int64_t)N * D; sp.ctx_st = D; sp.want_lse = want_lse;
  return attn_generic(bf16, sp, B, N, H, dh, buf, s);
}

// y = act(A . W^T + bias) (+ residual) in the context's compute dtype; out_f32 forces an f32 C.
static int linear(int bf16, const void* A, int64_t lda, const void* W, const float* bias, const float* residual, void* C,
                  int64_t ldc, bool c_f32, int M, int N, int K, int act, hipStream_t s) {
  if (bf16) {
    GemmBf16Args g{};
    g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = K; g.bias = bias; g.residual = residual; g.ldr = ldc;
    g.C = C; g.ldc = ldc; g.c_is_bf16 = c_f32 ? 0 : 1; g.M = M; g.N = N; g.K = K; g.batch = 1; g.act = act; g.alpha = 1.f;
    g.f16 = bf16 == HK_F16; g.h2 = bf16 == HK_F16X2;
    return gemm_bf16(g, s);
  }
  GemmF32Args g{};
  g.A = (const float*)A; g.lda = lda; g.B = (const float*)W; g.sbk = 1; g.sbn = K; g.bias = bias; g.residual = residual; g.ldr = ldc;
  g.C = (float*)C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.batch = 1; g.inner = 1; g.act = act; g.alpha = 1.f;
  return gemm_f32(g, s);
}
// The two GEMM forms a folded LayerNorm is made of (2-byte modes, persistent kernel; GemmBf16Args::copy16 / ln_stats):
//   producer: C (f32) = A.W^T + bias (+ residual), plus its 2-byte copy `copy16` [M, N] and the slice statistics of the finished rows
//   consumer: C (2-byte) = act(rstd (A.W'^T - mean c) + b') with (mean, rstd) per row
static int linear_ln_producer(int hk, const void* A, int64_t lda, const void* W, const float* bias, const float* residual, float* C, int64_t ldc,
                              void* copy16, float* slice_stats, int M, int N, int K, hipStream_t s) {
  GemmBf16Args g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = K; g.bias = bias; g.residual = residual; g.ldr = ldc;
  g.C = C; g.ldc = ldc; g.c_is_bf16 = 0; g.M = M; g.N = N; g.K = K; g.batch = 1; g.act = ACT_NONE; g.alpha = 1.f; g.f16 = hk == HK_F16; g.h2 = hk == HK_F16X2;
  g.copy16 = copy16; g.ld16 = N; g.row_stats = slice_stats;
  return gemm_bf16(g, s);
}
static int linear_ln_consumer(int hk, const void* A, int64_t lda, const void* Wf, const float* bias_f, const float* c_vec, const float* mean_rstd,
                              void* C, int64_t ldc, int M, int N, int K, int act, hipStream_t s) {
  GemmBf16Args g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)Wf; g.ldw = K; g.bias = bias_f;
  g.C = C; g.ldc = ldc; g.c_is_bf16 = 1; g.M = M; g.N = N; g.K = K; g.batch = 1; g.act = act; g.alpha = 1.f; g.f16 = hk == HK_F16; g.h2 = hk == HK_F16X2;
  g.ln_stats = mean_rstd; g.ln_c = c_vec;
  return gemm_bf16(g, s);
}

// C = act((A8 . W8^T) * sa[m] * sw[n] + bias) (+ residual): fp8 e4m3 operands, f32 accumulate
static int linear_fp8(const uint8_t* A8, const float* sa, int64_t lda, const uint8_t* W8, const float* sw, const float* bias,
                      const float* residual, void* C, int64_t ldc, bool c_f32, int M, int N, int K, int act, hipStream_t s) {
  GemmBf16Args g{};
  g.A = (const bf16_t*)A8; g.lda = lda; g.W = (const bf16_t*)W8; g.ldw = K; g.bias = bias; g.residual = residual; g.ldr = ldc;
  g.C = C; g.ldc = ldc; g.c_is_bf16 = c_f32 ? 0 : 1; g.M = M; g.N = N; g.K = K; g.batch = 1; g.act = act; g.alpha = 1.f;
  g.fp8 = 1; g.row_scale = sa; g.col_scale = sw;
  return gemm_bf16(g, s);
}
// The MX forms (GemmBf16Args::a_mx / c_mx): A8 with E8M0 block scales a_mx [K/128][M][4] instead of row scales, and / or the output written as
// e4m3 [M,N] + block scales c_mx_scale [N/128][M][4] (then C is unused).
static int linear_fp8_mx(const uint8_t* A8, const float* sa, const uint8_t* a_mx, int64_t lda, const uint8_t* W8, const float* sw, const float* bias,
                         const float* residual, void* C, int64_t ldc, bool c_f32, uint8_t* c_mx, uint8_t* c_mx_scale, int M, int N, int K, int act,
                         hipStream_t s) {
  GemmBf16Args g{};
  g.A = (const bf16_t*)A8; g.lda = lda; g.W = (const bf16_t*)W8; g.ldw = K; g.bias = bias; g.residual = residual; g.ldr = ldc;
  g.C = c_mx ? (void*)c_mx : C; g.ldc = ldc; g.c_is_bf16 = (c_f32 && !c_mx) ? 0 : 1; g.M = M; g.N = N; g.K = K; g.batch = 1; g.act = act; g.alpha = 1.f;
  g.fp8 = 1; g.row_scale = sa; g.col_scale = sw; g.a_mx = a_mx; g.c_mx = c_mx; g.c_mx_scale = c_mx_scale;
  return gemm_bf16(g, s);
}

__global__ void zero_diag_kernel(float* sim, int n, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) sim[(i / n) * (int64_t)n * n + (i % n) * (int64_t)(n + 1)] = 0.f;
}
// `kind` = HalfKind of src (two-plane f16: a contiguous buffer whose rows are multiples of 8 elements, so flat index = element index)
__global__ void unpack_bf16_kernel(const bf16_t* src, float* dst, int64_t n, int kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = kind == HK_F16X2 ? ld_elem<h2_t>(reinterpret_cast<const h2_t*>(src), i) : kind == HK_F16 ? h2f(f16_t{src[i]}) : bf2f(src[i]);
}

// similarity map from L2-normalised patch rows xhat [B,n,D] (compute dtype) -> sim [B,n,n] f32
static int similarity_from_xhat(int bf16, const void* xhat, int B, int n, int D, float temperature, int add_self, float* sim, hipStream_t s) {
  if (bf16) {
    GemmBf16Args g{};
    g.A = (const bf16_t*)xhat; g.lda = D; g.strideA = (int64_t)n * D; g.W = (const bf16_t*)xhat; g.ldw = D; g.strideW = (int64_t)n * D;
    g.C = sim; g.ldc = n; g.strideC = (int64_t)n * n; g.c_is_bf16 = 0; g.M = n; g.N = n; g.K = D; g.batch = B; g.act = 0;
    g.alpha = 1.0f / temperature; g.f16 = bf16 == HK_F16; g.h2 = bf16 == HK_F16X2;
    SG_TRY(gemm_bf16(g, s));
  } else {
    GemmF32Args g{};
    g.A = (const float*)xhat; g.lda = D; g.sAo = (int64_t)n * D; g.B = (const float*)xhat; g.sbk = 1; g.sbn = D; g.sBo = (int64_t)n * D;
    g.C = sim; g.ldc = n; g.sCo = (int64_t)n * n; g.M = n; g.N = n; g.K = D; g.batch = B; g.inner = 1; g.act = 0; g.alpha = 1.0f / temperature;
    SG_TRY(gemm_f32(g, s));
  }
  if (!add_self) {
    const int64_t total = (int64_t)B * n;
    hipLaunchKernelGGL(zero_diag_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, sim, n, total);
    SG_LAUNCH_CHECK();
  }
  return SG_OK;
}

// ---- workspace plan of one forward (shared by the size query and the forward itself) ---------------------------------
struct Plan {
  void *patchA; float* patchOut; float* pos_r; float* x; void* xn; void* qkv; void* ctx; void* hbuf; void* xhat; float* sim;
  float *lse, *lse1, *attn_cls, *attn_diag, *out_last, *y; int32_t *idx_out, *idx_sa; void* refine_scratch;
  float *scores, *probs;
  float *omega, *qnorm, *knorm;
  float *ln_slice, *ln_rows;                              // folded LayerNorm: slice statistics [D/64][R][2] (slice-major) written by the producing GEMM, (mean, rstd) [R][2]
  uint8_t* hmx;                                           // SG_PREC_FP8: MX block scales of h8 ([M/128][R][4] E8M0), written by the fc GEMM's epilogue
  uint8_t *x8, *h8; float *sx8, *sh8;                     // SG_PREC_FP8: quantised LN output / GELU output + per-row scales
  float *attn_avg, *sa_tmp, *sa_qk32, *sa_scores, *sa_probs;   // self-attention enhancement, mode='attention'; layer fusion
  float *lf_acc;                                              // layer fusion: the EMA of the head-averaged attention maps [B,N,N]
  // GEM
  float* x_gem; void* gnorm[3]; void* gatt[3]; float* inv_temp; float* gem_out; void* ctx2;
};

static size_t plan(const sg_context* c, int B, int gh, int gw, const sg_forward_opts* o, void* ws, bool dry, Plan& p) {
  const sg_vit_desc& d = c->d;
  const int n = gh * gw, N = n + 1;
  const int64_t R = (int64_t)B * N;
  const size_t e = c->esz;
  Bump b(ws, 0, dry);
  p.patchA = b.take((size_t)B * n * c->Kpad * e);
  p.patchOut = b.get<float>((size_t)B * n * d.width);
  p.pos_r = b.get<float>((size_t)N * d.width);
  p.x = b.get<float>(R * d.width);
  p.xn = b.take(R * d.width * e);
  p.qkv = b.take(R * 3 * d.width * e);
  p.ctx = b.take(R * d.width * e);
  p.hbuf = b.take(R * d.mlp_width * e);
  p.x8 = p.h8 = p.hmx = nullptr; p.sx8 = p.sh8 = nullptr;
  p.ln_slice = p.ln_rows = nullptr;
  if (c->hk != HK_F32 && !c->fp8 && d.width % 64 == 0) { p.ln_slice = b.get<float>(R * (d.width / 64) * 2); p.ln_rows = b.get<float>(R * 2); }
  if (c->fp8) {
    p.x8 = (uint8_t*)b.take(R * d.width); p.sx8 = b.get<float>(R);
    p.h8 = (uint8_t*)b.take(R * d.mlp_width); p.sh8 = b.get<float>(R);
    p.hmx = (uint8_t*)b.take(R * (d.mlp_width / 32 + 4));
  }
  p.xhat = nullptr; p.sim = nullptr;
  if (o->similarity_enabled) { p.xhat = b.take((size_t)B * n * d.width * e); p.sim = b.get<float>((size_t)B * n * n); }
  p.lse = b.get<float>((size_t)B * d.heads * N);
  p.lse1 = b.get<float>((size_t)B * d.heads * N);
  p.attn_cls = b.get<float>((size_t)B * N);
  p.attn_diag = b.get<float>((size_t)B * N);
  p.out_last = b.get<float>(R * d.width);
  p.y = b.get<float>(R * d.embed_dim);
  const int k_out = o->outlier_enabled ? (o->outlier_top_k < n ? o->outlier_top_k : n) : 0;
  const int k_sa = (o->outlier_enabled && o->selfattn_enabled) ? (o->selfattn_top_k < n ? o->selfattn_top_k : n) : 0;
  p.idx_out = b.get<int32_t>((size_t)B * (k_out > 0 ? k_out : 1));
  p.idx_sa = b.get<int32_t>((size_t)B * (k_sa > 0 ? k_sa : 1));
  const int kmax = k_out > k_sa ? k_out : k_sa;
  p.refine_scratch = b.take(refine_scratch_bytes(B, d.width, kmax > 0 ? kmax : 1));
  p.scores = p.probs = nullptr;
  if (!c->hk) { p.scores = b.get<float>((size_t)B * d.heads * N * N); p.probs = b.get<float>((size_t)B * d.heads * N * N); }
  p.attn_avg = p.sa_tmp = p.sa_qk32 = p.sa_scores = p.sa_probs = p.lf_acc = nullptr;
  if (o->layer_fusion_enabled) p.lf_acc = b.get<float>((size_t)B * N * N);
  if (o->layer_fusion_enabled || (o->outlier_enabled && o->selfattn_enabled && o->selfattn_mode == 1)) {
    p.attn_avg = b.get<float>((size_t)B * N * N); p.sa_tmp = b.get<float>(R * d.width);
    if (c->hk) {                                        // one image at a time: f32 copies of q|k, scores and probabilities of all heads
      p.sa_qk32 = b.get<float>((size_t)N * 2 * d.width); p.sa_scores = b.get<float>((size_t)d.heads * N * N); p.sa_probs = b.get<float>((size_t)d.heads * N * N);
    }
  }
  p.omega = p.qnorm = p.knorm = nullptr;
  if (o->model_type == SG_NACLIP || o->model_type == SG_NONLY || o->model_type == SG_GAV) {
    p.omega = b.get<float>((size_t)n * n); p.qnorm = b.get<float>((size_t)B * d.heads * N); p.knorm = b.get<float>((size_t)B * d.heads * N);
  }
  p.x_gem = nullptr;
  if (o->model_type == SG_GEM) {
    p.x_gem = b.get<float>(R * d.width);
    p.gem_out = b.get<float>(R * d.width);
    for (int t = 0; t < 3; ++t) { p.gnorm[t] = b.take(R * d.width * e); p.gatt[t] = b.take(R * d.width * e); }
    p.ctx2 = b.take(R * d.width * e);
    p.inv_temp = b.get<float>(B);
  }
  return align_up(b.off, 256);
}

static int find_layer_tensor(const char* rest, int& slot) {
  static const char* names[12] = {"ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                                  "attn.out_proj.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
                                  "mlp.c_proj.weight", "mlp.c_proj.bias"};
  for (int i = 0; i < 12; ++i) if (!strcmp(rest, names[i])) { slot = i; return 1; }
  return 0;
}

}  // namespace sg

extern "C" const char* sg_last_error(void) { return g_err; }
extern "C" int sg_version(void) { return 100; }

// Live timing for bench.py: HIP events bracket every launch of a kernel family on the stream it is launched on.
extern "C" int sg_profile_enable(int capacity) {
  SG_REQUIRE(capacity > 0 && capacity <= (1 << 20), "sg_profile_enable: bad capacity");
  if (capacity > g_prof.cap) {
    for (int c = 0; c < PROF_NCAT; ++c) {
      g_prof.start[c].resize(capacity); g_prof.stop[c].resize(capacity);
      for (int i = g_prof.cap; i < capacity; ++i) { SG_HIP(hipEventCreate(&g_prof.start[c][i])); SG_HIP(hipEventCreate(&g_prof.stop[c][i])); }
    }
    g_prof.cap = capacity;
  }
  for (int c = 0; c < PROF_NCAT; ++c) { g_prof.used[c] = 0; g_prof.work[c] = 0; g_prof.dropped[c] = 0; }
  g_prof.on = true;
  return SG_OK;
}
// Raw bf16 GEMM on caller-packed operands (A [M,K], W [N,K] bf16, K % 64 == 0): C = act(A.W^T + bias) (+ residual).
extern "C" int sg_gemm_bf16_raw(const void* A, const void* W, const float* bias, const float* residual, void* C, int M, int N, int K,
                                int act, int c_is_bf16, sg_stream st) {
  SG_REQUIRE(A && W && C, "sg_gemm_bf16_raw: null pointer");
  return linear(HK_BF16, A, K, W, bias, residual, C, N, !c_is_bf16, M, N, K, act, as_stream(st));
}
// fp8 (OCP e4m3) GEMM on caller-quantised operands: A8 [M,K] with per-row scales sa [M], W8 [N,K] with per-row scales sw [N], K % 128 == 0.
extern "C" int sg_gemm_fp8_raw(const void* A8, const float* sa, const void* W8, const float* sw, const float* bias, const float* residual,
                               void* C, int M, int N, int K, int act, int c_is_bf16, sg_stream st) {
  SG_REQUIRE(A8 && W8 && sa && sw && C, "sg_gemm_fp8_raw: null pointer");
  return linear_fp8((const uint8_t*)A8, sa, K, (const uint8_t*)W8, sw, bias, residual, C, N, !c_is_bf16, M, N, K, act, as_stream(st));
}
// The MXFP8 forms of the fp8 GEMM (M >= 1024, N >= 256, N % 128 == 0 when c_mx): exactly one of sa (per-row scales) / a_mx (E8M0 block scales,
// [K/128][M][4] bytes) describes A8; when c_mx is given the result is written as e4m3 [M,N] + block scales c_mx_scale [N/128][M][4] and C is unused.
extern "C" int sg_gemm_fp8_mx_raw(const void* A8, const float* sa, const void* a_mx, const void* W8, const float* sw, const float* bias,
                                  const float* residual, void* C, void* c_mx, void* c_mx_scale, int M, int N, int K, int act, int c_is_bf16, sg_stream st) {
  SG_REQUIRE(A8 && W8 && sw && ((sa != nullptr) != (a_mx != nullptr)) && (C || c_mx), "sg_gemm_fp8_mx_raw: bad pointers");
  SG_REQUIRE(!c_mx || c_mx_scale, "sg_gemm_fp8_mx_raw: c_mx needs c_mx_scale");
  return linear_fp8_mx((const uint8_t*)A8, sa, (const uint8_t*)a_mx, K, (const uint8_t*)W8, sw, bias, residual, C, N, !c_is_bf16, (uint8_t*)c_mx,
                       (uint8_t*)c_mx_scale, M, N, K, act, as_stream(st));
}
// rows of f32 -> e4m3 + per-row absmax scale (scale[r] = max|x[r,:]| / 448), the quantiser both fp8 operands go through
extern "C" int sg_quantize_rows_fp8(const float* x, int64_t rows, int D, void* y, float* scale, sg_stream st) {
  SG_REQUIRE(x && y && scale && rows > 0 && D > 0, "sg_quantize_rows_fp8: bad arguments");
  return quantize_rows_fp8(x, 0, D, (uint8_t*)y, D, scale, rows, D, as_stream(st));
}
// Tuning hook for the bf16 GEMM tile configuration (-1 = automatic).
extern "C" int sg_set_gemm_config(int cfg) {
  set_gemm_config(cfg);
  return SG_OK;
}
extern "C" int sg_profile_disable(void) { g_prof.on = false; return SG_OK; }
// category: 0 bf16 GEMM (non-persistent tile variants), 1 fused attention, 2 f32 GEMM, 3 the persistent bf16 GEMM (all instantiations), 4 fp8 GEMM,
// 5 / 6 the persistent GEMM's folded-LayerNorm consumer / producer instantiations alone.  Call after the stream
// has been synchronised.
extern "C" int sg_profile_read(int category, double* total_ms, double* total_flops, int64_t* launches, int64_t* dropped) {
  SG_REQUIRE(category >= 0 && category < PROF_NCAT && total_ms && total_flops && launches, "sg_profile_read: bad argument");
  double ms = 0, work = 0; int64_t n = 0, drop = 0;
  // category 3 = EVERY launch of the persistent kernel: its plain instantiation plus the two folded-LayerNorm ones (5, 6)
  const int members[3] = {category, category == PROF_GEMM_PERSIST ? PROF_GEMM_PERSIST_LN_CONSUMER : -1, category == PROF_GEMM_PERSIST ? PROF_GEMM_PERSIST_LN_PRODUCER : -1};
  for (int c : members) {
    if (c < 0) continue;
    for (int i = 0; i < g_prof.used[c]; ++i) {
      float t = 0.f;
      SG_HIP(hipEventElapsedTime(&t, g_prof.start[c][i], g_prof.stop[c][i]));
      ms += t;
    }
    work += g_prof.work[c]; n += g_prof.used[c]; drop += g_prof.dropped[c];
  }
  *total_ms = ms; *total_flops = work; *launches = n;
  if (dropped) *dropped = drop;
  return SG_OK;
}

extern "C" int sg_create(sg_context** out, int device, const sg_vit_desc* desc) {
  SG_REQUIRE(out && desc, "sg_create: null argument");
  const sg_vit_desc& d = *desc;
  SG_REQUIRE(d.width > 0 && d.layers >= 2 && d.heads > 0 && d.width % d.heads == 0 && d.patch > 0 && d.embed_dim > 0 && d.grid0 > 0 &&
             d.mlp_width > 0, "sg_create: bad descriptor");
  SG_REQUIRE(d.precision == SG_PREC_F32 || d.precision == SG_PREC_BF16 || d.precision == SG_PREC_FP8 || d.precision == SG_PREC_F16 ||
             d.precision == SG_PREC_F16X2, "sg_create: bad precision %d", d.precision);
  SG_REQUIRE(d.width % 4 == 0 && d.embed_dim % 4 == 0, "sg_create: width / embed_dim must be multiples of 4");
  if (d.precision == SG_PREC_FP8)
    SG_REQUIRE(d.width % 128 == 0 && d.mlp_width % 128 == 0, "sg_create: fp8 mode needs width and mlp_width to be multiples of 128");
  if (d.precision != SG_PREC_F32) {
    SG_REQUIRE(d.width % 64 == 0 && d.mlp_width % 64 == 0, "sg_create: bf16 mode needs width and mlp_width to be multiples of 64");
    const int dh = d.width / d.heads;
    SG_REQUIRE(dh == 32 || dh == 64 || dh == 80 || dh == 128, "sg_create: bf16 mode supports head_dim 32/64/80/128, got %d", dh);
  }
  DeviceGuard dg(device);
  sg_context* c = new sg_context();
  c->d = d; c->device = device; c->fp8 = d.precision == SG_PREC_FP8;
  c->hk = hk_of_precision(d.precision);
  c->esz = hk_esz(c->hk);
  c->Kpatch = 3 * d.patch * d.patch;
  c->Kpad = (int)align_up(c->Kpatch, 64);
  c->finalized = false;
  c->n_expected = expected_tensors(d);
  c->have.assign(c->n_expected, 0);
  // arena: all packed weights
  Bump b(nullptr, 0, true);
  auto lay = [&](Bump& bb) {
    const size_t e = c->esz; const int D = d.width, M = d.mlp_width;
    c->w_patch = bb.take((size_t)D * c->Kpad * e);
    c->w_projT = bb.take((size_t)d.embed_dim * D * e);
    c->cls_emb = bb.get<float>(D);
    c->pos = bb.get<float>((size_t)(d.grid0 * d.grid0 + 1) * D);
    c->lnpre_g = bb.get<float>(D); c->lnpre_b = bb.get<float>(D); c->lnpost_g = bb.get<float>(D); c->lnpost_b = bb.get<float>(D);
    c->layers.resize(d.layers);
    for (auto& L : c->layers) {
      L.w_qkv = bb.take((size_t)3 * D * D * e); L.w_out = bb.take((size_t)D * D * e);
      L.w_fc = bb.take((size_t)M * D * e); L.w_proj = bb.take((size_t)D * M * e);
      L.b_qkv = bb.get<float>(3 * D); L.b_out = bb.get<float>(D); L.b_fc = bb.get<float>(M); L.b_proj = bb.get<float>(D);
      L.ln1_g = bb.get<float>(D); L.ln1_b = bb.get<float>(D); L.ln2_g = bb.get<float>(D); L.ln2_b = bb.get<float>(D);
      L.w_qkv8 = L.w_fc8 = L.w_proj8 = nullptr; L.s_qkv = L.s_fc = L.s_proj = nullptr;
      if (c->hk != HK_F32 && !c->fp8) {     // folded-LayerNorm operands exist for the 2-byte modes and the two-plane f16 mode
        L.w_qkv_f = bb.take((size_t)3 * D * D * e); L.w_fc_f = bb.take((size_t)M * D * e);
        L.c_qkv = bb.get<float>(3 * D); L.bf_qkv = bb.get<float>(3 * D); L.c_fc = bb.get<float>(M); L.bf_fc = bb.get<float>(M);
      }
      if (c->fp8) {
        L.w_qkv8 = (uint8_t*)bb.take((size_t)3 * D * D); L.w_fc8 = (uint8_t*)bb.take((size_t)M * D); L.w_proj8 = (uint8_t*)bb.take((size_t)D * M);
        L.s_qkv = bb.get<float>(3 * D); L.s_fc = bb.get<float>(M); L.s_proj = bb.get<float>(D);
      }
    }
  };
  lay(b);
  c->arena_bytes = align_up(b.off, 256);
  hipError_t e = hipMalloc(&c->arena, c->arena_bytes);
  if (e != hipSuccess) { delete c; return fail(SG_ERR_HIP, "sg_create: hipMalloc(%zu) -> %s", c->arena_bytes, hipGetErrorString(e)); }
  Bump real(c->arena, c->arena_bytes, false);
  lay(real);
  *out = c;
  return SG_OK;
}

extern "C" void sg_destroy(sg_context* c) {
  if (!c) return;
  for (auto& L : c->layers) { if (L.stage_qkv) (void)hipFree(L.stage_qkv); if (L.stage_fc) (void)hipFree(L.stage_fc); }
  if (c->arena) (void)hipFree(c->arena);
  delete c;
}

extern "C" int sg_vit_set_tensor(sg_context* c, const char* name, const float* src, int64_t numel, sg_stream st) {
  SG_REQUIRE(c && name && src, "sg_vit_set_tensor: null argument");
  DeviceGuard dg(c->device);
  hipStream_t s = as_stream(st);
  const sg_vit_desc& d = c->d;
  const int D = d.width, M = d.mlp_width, E = d.embed_dim;
  const int to_bf16 = c->hk;
  auto copyf = [&](float* dst, int64_t n) -> int {
    SG_REQUIRE(numel == n, "sg_vit_set_tensor(%s): expected %lld elements, got %lld", name, (long long)n, (long long)numel);
    SG_HIP(hipMemcpyAsync(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    return SG_OK;
  };
  auto packw = [&](void* dst, int rows, int cols, int cols_pad) -> int {
    SG_REQUIRE(numel == (int64_t)rows * cols, "sg_vit_set_tensor(%s): expected %lld elements, got %lld", name, (long long)rows * cols, (long long)numel);
    return pack_rows(src, rows, cols, cols, dst, cols_pad, to_bf16, s);
  };
  int slot = -1, rc = SG_OK;
  if (!strcmp(name, "conv1.weight")) { slot = 0; rc = packw(c->w_patch, D, c->Kpatch, c->Kpad); }
  else if (!strcmp(name, "class_embedding")) { slot = 1; rc = copyf(c->cls_emb, D); }
  else if (!strcmp(name, "positional_embedding")) { slot = 2; rc = copyf(c->pos, (int64_t)(d.grid0 * d.grid0 + 1) * D); }
  else if (!strcmp(name, "ln_pre.weight")) { slot = 3; rc = copyf(c->lnpre_g, D); }
  else if (!strcmp(name, "ln_pre.bias")) { slot = 4; rc = copyf(c->lnpre_b, D); }
  else if (!strcmp(name, "ln_post.weight")) { slot = 5; rc = copyf(c->lnpost_g, D); }
  else if (!strcmp(name, "ln_post.bias")) { slot = 6; rc = copyf(c->lnpost_b, D); }
  else if (!strcmp(name, "proj")) {
    slot = 7;
    SG_REQUIRE(numel == (int64_t)D * E, "sg_vit_set_tensor(proj): expected %d x %d", D, E);
    rc = transpose_pack(src, D, E, c->w_projT, to_bf16, s);
  } else {
    int li = -1, consumed = 0;
    if (sscanf(name, "transformer.resblocks.%d.%n", &li, &consumed) == 1 && consumed > 0 && li >= 0 && li < d.layers) {
      int t;
      if (find_layer_tensor(name + consumed, t)) {
        LayerW& L = c->layers[li];
        slot = 8 + li * 12 + t;
        auto stage = [&](float*& dst, int64_t n) -> int {  // f32 copy of a weight the LayerNorm in front of it will be folded into
          if (!L.w_qkv_f || numel != n) return SG_OK;
          if (!dst) SG_HIP(hipMalloc((void**)&dst, (size_t)n * 4));
          SG_HIP(hipMemcpyAsync(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
          return SG_OK;
        };
        if (t <= 3 || (t >= 6 && t <= 9)) L.folded = false; // anything the folded operands are made from
        switch (t) {
          case 0: rc = copyf(L.ln1_g, D); break;
          case 1: rc = copyf(L.ln1_b, D); break;
          case 2: rc = packw(L.w_qkv, 3 * D, D, D);
                  if (rc == SG_OK && c->fp8) rc = quantize_rows_fp8(src, 0, D, L.w_qkv8, D, L.s_qkv, 3 * D, D, s);
                  if (rc == SG_OK) rc = stage(L.stage_qkv, (int64_t)3 * D * D);
                  break;
          case 3: rc = copyf(L.b_qkv, 3 * D); break;
          case 4: rc = packw(L.w_out, D, D, D); break;
          case 5: rc = copyf(L.b_out, D); break;
          case 6: rc = copyf(L.ln2_g, D); break;
          case 7: rc = copyf(L.ln2_b, D); break;
          case 8: rc = packw(L.w_fc, M, D, D);
                  if (rc == SG_OK && c->fp8) rc = quantize_rows_fp8(src, 0, D, L.w_fc8, D, L.s_fc, M, D, s);
                  if (rc == SG_OK) rc = stage(L.stage_fc, (int64_t)M * D);
                  break;
          case 9: rc = copyf(L.b_fc, M); break;
          case 10: rc = packw(L.w_proj, D, M, M);
                   if (rc == SG_OK && c->fp8) rc = quantize_rows_fp8(src, 0, M, L.w_proj8, M, L.s_proj, D, M, s);
                   break;
          case 11: rc = copyf(L.b_proj, D); break;
        }
      }
    }
  }
  if (slot < 0) return fail(SG_ERR_INVALID, "sg_vit_set_tensor: unknown tensor name '%s'", name);
  if (rc != SG_OK) return rc;
  c->have[slot] = 1;
  c->finalized = false;
  return SG_OK;
}

extern "C" int sg_vit_finalize(sg_context* c, sg_stream st) {
  SG_REQUIRE(c, "sg_vit_finalize: null context");
  int missing = 0;
  for (int i = 0; i < c->n_expected; ++i) if (!c->have[i]) ++missing;
  if (missing) return fail(SG_ERR_STATE, "sg_vit_finalize: %d of %d tensors were never set", missing, c->n_expected);
  // fold ln_1 into the QKV weight and ln_2 into the fc weight of every block whose f32 weights are still staged; a block whose LayerNorm
  // parameters were replaced without its weights keeps the explicit LayerNorm pass (folded == false)
  DeviceGuard dg(c->device);
  const int D = c->d.width, M = c->d.mlp_width;
  for (auto& L : c->layers) {
    if (!L.w_qkv_f || L.folded || !L.stage_qkv || !L.stage_fc) continue;
    SG_TRY(fold_ln_weight(L.stage_qkv, 3 * D, D, L.ln1_g, L.ln1_b, L.b_qkv, c->hk, L.w_qkv_f, L.c_qkv, L.bf_qkv, as_stream(st)));
    SG_TRY(fold_ln_weight(L.stage_fc, M, D, L.ln2_g, L.ln2_b, L.b_fc, c->hk, L.w_fc_f, L.c_fc, L.bf_fc, as_stream(st)));
    L.folded = true;
  }
  SG_HIP(hipStreamSynchronize(as_stream(st)));               // the staged f32 copies are freed below: their last readers (and writers) are on this stream
  for (auto& L : c->layers) {
    if (L.stage_qkv) { (void)hipFree(L.stage_qkv); L.stage_qkv = nullptr; }
    if (L.stage_fc) { (void)hipFree(L.stage_fc); L.stage_fc = nullptr; }
  }
  c->finalized = true;
  return SG_OK;
}

extern "C" size_t sg_vit_workspace_bytes(const sg_context* c, int n_tiles, int gh, int gw, const sg_forward_opts* o) {
  if (!c || !o || n_tiles <= 0 || gh <= 0 || gw <= 0) return 0;
  Plan p;
  return plan(c, n_tiles, gh, gw, o, nullptr, true, p);
}

// ln2_folded: the caller's out-projection GEMM already left x's 2-byte copy in p.xn and its slice statistics in p.ln_slice;
// emit_next: the proj GEMM does the same for the x it produces (the next block's ln_1), *x16_valid reports it.
static int mlp_block(sg_context* c, const LayerW& L, float* x, const Plan& p, int64_t R, hipStream_t s, bool ln2_folded = false,
                     bool emit_next = false, bool* x16_valid = nullptr) {
  const sg_vit_desc& d = c->d;
  const int D = d.width, M = d.mlp_width;
  const int act = d.quick_gelu ? ACT_QUICK_GELU : ACT_GELU;
  if (c->fp8 && p.x8 && L.w_fc8) {                          // fp8 linears: LN -> e4m3 + row scale; GELU output re-quantised per row
    SG_TRY(layernorm_fp8(x, D, L.ln2_g, L.ln2_b, p.x8, D, p.sx8, R, D, 1e-5f, s));
    if (R >= 1024 && M % 128 == 0 && D % 256 == 0 && get_gemm_config() != 33) {
      // MXFP8 hand-off: the fc epilogue writes GELU(h) as e4m3 with one power-of-two scale per 32 columns, which the proj GEMM's scaled MFMA
      // consumes directly -- no [R, M] 2-byte intermediate and no separate row-quantisation pass (finer-grained scales than one per row, too)
      SG_TRY(linear_fp8_mx(p.x8, p.sx8, nullptr, D, L.w_fc8, L.s_fc, L.b_fc, nullptr, nullptr, M, false, p.h8, p.hmx, (int)R, M, D, act, s));
      return linear_fp8_mx(p.h8, nullptr, p.hmx, M, L.w_proj8, L.s_proj, L.b_proj, x, x, D, true, nullptr, nullptr, (int)R, D, M, ACT_NONE, s);
    }
    SG_TRY(linear_fp8(p.x8, p.sx8, D, L.w_fc8, L.s_fc, L.b_fc, nullptr, p.hbuf, M, false, (int)R, M, D, act, s));
    SG_TRY(quantize_rows_fp8(p.hbuf, 1, M, p.h8, M, p.sh8, R, M, s));
    return linear_fp8(p.h8, p.sh8, M, L.w_proj8, L.s_proj, L.b_proj, x, x, D, true, (int)R, D, M, ACT_NONE, s);
  }
  if (x16_valid) *x16_valid = false;
  if (ln2_folded) {
    SG_TRY(ln_stats_finalize(p.ln_slice, R, D, 1e-5f, p.ln_rows, s));
    SG_TRY(linear_ln_consumer(c->hk, p.xn, D, L.w_fc_f, L.bf_fc, L.c_fc, p.ln_rows, p.hbuf, M, (int)R, M, D, act, s));
  } else {
    SG_TRY(layernorm(x, D, L.ln2_g, L.ln2_b, p.xn, D, c->hk, R, D, 1e-5f, s));
    SG_TRY(linear(c->hk, p.xn, D, L.w_fc, L.b_fc, nullptr, p.hbuf, M, false, (int)R, M, D, act, s));
  }
  if (emit_next) {
    SG_TRY(linear_ln_producer(c->hk, p.hbuf, M, L.w_proj, L.b_proj, x, x, D, p.xn, p.ln_slice, (int)R, D, M, s));
    if (x16_valid) *x16_valid = true;
  } else SG_TRY(linear(c->hk, p.hbuf, M, L.w_proj, L.b_proj, x, x, D, true, (int)R, D, M, ACT_NONE, s));
  return SG_OK;
}

// One ordinary residual block (reference open_clip/transformer.py:234-254), x updated in place.
static int averaged_attention(sg_context* c, const Plan& p, int B, int N, hipStream_t s);
// x16_valid (optional, in/out): in -- p.xn / p.ln_slice already hold the 2-byte copy and the slice statistics of THIS x (written by the
// previous block's proj GEMM), so ln_1 is folded into the QKV GEMM; out -- the same for the x this block leaves behind.
static int std_block(sg_context* c, const LayerW& L, float* x, const Plan& p, int B, int N, bool stats, hipStream_t s, bool want_avg = false,
                     bool causal = false, bool* x16_valid = nullptr) {
  const sg_vit_desc& d = c->d;
  const int D = d.width, H = d.heads;
  const int64_t R = (int64_t)B * N;
  AttnBuffers ab{p.scores, p.probs, p.lse, p.lse1, p.omega, p.qnorm, p.knorm};
  // LayerNorm folding: 2-byte modes, shapes that run on the persistent GEMM; cfg 34 (tuning) switches it off
  // (small launches -- a tile or two per call -- run the residual GEMMs on smaller tiles instead, see gemm_bf16.hip few_tiles)
  const bool fold = c->hk && !c->fp8 && L.folded && p.ln_slice && R < (1ll << 31) && gemm_bf16_ln_fold_ok((int)R, D, D) && D % 256 == 0 &&
                    gemm_bf16_prefers_persistent((int)R, D) && get_gemm_config() != 34;
  const bool ln1_folded = fold && x16_valid && *x16_valid;
  if (x16_valid) *x16_valid = false;
  if (c->fp8 && p.x8 && L.w_qkv8) {
    SG_TRY(layernorm_fp8(x, D, L.ln1_g, L.ln1_b, p.x8, D, p.sx8, R, D, 1e-5f, s));
    SG_TRY(linear_fp8(p.x8, p.sx8, D, L.w_qkv8, L.s_qkv, L.b_qkv, nullptr, p.qkv, 3 * D, false, (int)R, 3 * D, D, ACT_NONE, s));
  } else if (ln1_folded) {
    SG_TRY(ln_stats_finalize(p.ln_slice, R, D, 1e-5f, p.ln_rows, s));
    SG_TRY(linear_ln_consumer(c->hk, p.xn, D, L.w_qkv_f, L.bf_qkv, L.c_qkv, p.ln_rows, p.qkv, 3 * D, (int)R, 3 * D, D, ACT_NONE, s));
  } else {
    SG_TRY(layernorm(x, D, L.ln1_g, L.ln1_b, p.xn, D, c->hk, R, D, 1e-5f, s));
    SG_TRY(linear(c->hk, p.xn, D, L.w_qkv, L.b_qkv, nullptr, p.qkv, 3 * D, false, (int)R, 3 * D, D, ACT_NONE, s));
  }
  SG_TRY(run_attention(c->hk, p.qkv, B, N, D, H, SG_VANILLA, nullptr, 0.f, nullptr, p.ctx, stats, ab, s, causal));
  if (stats)
    SG_TRY(attention_stats(p.qkv, c->hk, (int64_t)N * 3 * D, 3 * D, p.lse, B, N, H, D / H, 1.0f / sqrtf((float)(D / H)), p.attn_cls,
                           p.attn_diag, s));
  if (want_avg) SG_TRY(averaged_attention(c, p, B, N, s));
  if (fold) SG_TRY(linear_ln_producer(c->hk, p.ctx, D, L.w_out, L.b_out, x, x, D, p.xn, p.ln_slice, (int)R, D, D, s));
  else SG_TRY(linear(c->hk, p.ctx, D, L.w_out, L.b_out, x, x, D, true, (int)R, D, D, ACT_NONE, s));
  return mlp_block(c, L, x, p, R, s, fold, fold && x16_valid != nullptr, x16_valid);
}

static int gem_forward_tail(sg_context* c, const sg_forward_opts* o, const Plan& p, int B, int N, hipStream_t s);

__global__ void unpack_qk_kernel(const void* __restrict__ qkv, int N, int D, float* __restrict__ out, int kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * 2 * D) return;
  const int t = (int)(i / (2 * D)), c = (int)(i % (2 * D));
  if (kind == HK_F16X2) { out[i] = ld_elem<h2_t>(reinterpret_cast<const h2_t*>(qkv) + (int64_t)t * 3 * D, c); return; }
  const bf16_t v = reinterpret_cast<const bf16_t*>(qkv)[(int64_t)t * 3 * D + c];
  out[i] = kind == HK_F16 ? h2f(f16_t{v}) : bf2f(v);
}
// head-averaged attention matrix [B,N,N] of the block whose packed qkv is in p.qkv (the tensor the reference gets from
// nn.MultiheadAttention(need_weights=True), transformer.py:609-610).  Only the optional mode='attention' enhancer needs it.
static int averaged_attention(sg_context* c, const Plan& p, int B, int N, hipStream_t s) {
  const sg_vit_desc& d = c->d;
  const int D = d.width, H = d.heads, dh = D / H;
  const float scale = 1.0f / sqrtf((float)dh);
  if (!c->hk) return head_mean(p.probs, B, H, N, p.attn_avg, s);        // parity mode: the probabilities are materialised already
  const int64_t NN = (int64_t)N * N;
  for (int b = 0; b < B; ++b) {
    const void* qkv = (const char*)p.qkv + (size_t)b * N * 3 * D * c->esz;
    hipLaunchKernelGGL(unpack_qk_kernel, dim3((unsigned)cdiv((int64_t)N * 2 * D, 256)), dim3(256), 0, s, qkv, N, D, p.sa_qk32, c->hk);
    SG_LAUNCH_CHECK();
    GemmF32Args g{};
    g.A = p.sa_qk32; g.lda = 2 * D; g.sAi = dh; g.B = p.sa_qk32 + D; g.sbk = 1; g.sbn = 2 * D; g.sBi = dh;
    g.C = p.sa_scores; g.ldc = N; g.sCi = NN; g.M = N; g.N = N; g.K = dh; g.batch = H; g.inner = H; g.act = 0; g.alpha = 1.f;
    SG_TRY(gemm_f32(g, s));
    SG_TRY(softmax_rows(p.sa_scores, N, (int64_t)H * N, N, H, nullptr, scale, nullptr, 0.f, 0, nullptr, nullptr, 0, 0, p.sa_probs, nullptr, s));
    SG_TRY(head_mean(p.sa_probs, 1, H, N, p.attn_avg + (int64_t)b * NN, s));
  }
  return SG_OK;
}

extern "C" int sg_vit_forward(sg_context* c, const sg_tile_batch* tiles, const sg_forward_opts* o, float* out_cls, float* out_tokens,
                              void* workspace, size_t workspace_bytes, sg_stream st) {
  SG_REQUIRE(c && tiles && o && out_tokens && workspace, "sg_vit_forward: null argument");
  if (!c->finalized) return fail(SG_ERR_STATE, "sg_vit_forward: weights not finalized (call sg_vit_finalize)");
  DeviceGuard dg(c->device);                                          // the context's device, whatever the caller's current device is
  hipStream_t s = as_stream(st);
  const sg_vit_desc& d = c->d;
  const int B = tiles->n_tiles, gh = tiles->grid_h, gw = tiles->grid_w, n = gh * gw, N = n + 1;
  const int D = d.width, H = d.heads, L = d.layers, E = d.embed_dim;
  const int64_t R = (int64_t)B * N;
  SG_REQUIRE(B > 0 && gh > 0 && gw > 0, "sg_vit_forward: empty batch");
  SG_REQUIRE(tiles->scene && tiles->windows, "sg_vit_forward: null scene / windows");
  SG_REQUIRE(o->model_type == SG_GEM || out_cls, "sg_vit_forward: out_cls required");
  SG_REQUIRE(R * (int64_t)(d.mlp_width > 3 * D ? d.mlp_width : 3 * D) < (1ll << 40), "sg_vit_forward: batch too large");
  if (o->outlier_enabled || o->selfattn_enabled) SG_REQUIRE(gh == gw, "sg_vit_forward: refiners need a square patch grid (reference transformer.py:583)");
  if (o->outlier_enabled) SG_REQUIRE(o->outlier_top_k >= 1, "sg_vit_forward: outlier_top_k must be >= 1");
  Plan p;
  const size_t need = plan(c, B, gh, gw, o, workspace, false, p);
  if (need > workspace_bytes) return fail(SG_ERR_STATE, "sg_vit_forward: workspace %zu < required %zu bytes", workspace_bytes, need);
  SG_REQUIRE((((uintptr_t)workspace) & 255) == 0, "sg_vit_forward: workspace must be 256-byte aligned");

  // ---- prologue: patch embed, class token, positional embedding, ln_pre (transformer.py:559-576) ----
  const bool gem = o->model_type == SG_GEM;
  SG_TRY(patchify(*tiles, d.patch, p.patchA, c->Kpad, c->hk, s));
  SG_TRY(linear(c->hk, p.patchA, c->Kpad, c->w_patch, nullptr, nullptr, p.patchOut, D, true, B * n, D, c->Kpad, ACT_NONE, s));
  const float* pos = c->pos;
  if (gh != d.grid0 || gw != d.grid0) { SG_TRY(posembed_resize(c->pos, d.grid0, D, gh, gw, gem ? 1 : 0, p.pos_r, s)); pos = p.pos_r; }
  SG_TRY(embed_assemble(p.patchOut, D, c->cls_emb, pos, c->lnpre_g, c->lnpre_b, p.x, B, N, D, 1e-5f, s));

  if (gem) {
    const int first = L - (o->gem_depth - 1);
    SG_REQUIRE(o->gem_depth >= 2 && first >= 0, "sg_vit_forward: gem_depth %d does not fit %d layers", o->gem_depth, L);
    // GEM + outlier suppression in ONE forward (BASELINE configs[2]; the reference cannot run it, SURVEY R5 -- DESIGN.md section 7 defines it):
    // detection on the head-averaged attention of the ORDINARY stream of block L-2, suppression on the GEM stream before ln_post.
    // The other refiners have no defined place in the GEM forward: refuse rather than ignore them.
    SG_REQUIRE(!o->selfattn_enabled && !o->similarity_enabled && !o->layer_fusion_enabled,
               "sg_vit_forward: GEM composes with outlier suppression only (self-attention / similarity enhancement and layer fusion are not defined for the GEM forward)");
    const bool gem_out = o->outlier_enabled != 0;
    bool x16 = false;                                                 // p.xn / p.ln_slice describe p.x (folded LayerNorm hand-off between blocks)
    for (int i = 0; i < first; ++i) SG_TRY(std_block(c, c->layers[i], p.x, p, B, N, gem_out && i == L - 2, s, false, false, &x16));
    SG_TRY(gem_forward_tail(c, o, p, B, N, s));
    if (gem_out) {
      const int k = o->outlier_top_k < n ? o->outlier_top_k : n;
      SG_TRY(select_topk(p.attn_cls, p.attn_diag, B, N, k, 0, p.idx_out, s));
      SG_TRY(neighbour_refine(p.x_gem, (int64_t)N * D, D, p.idx_out, B, gh, gw, D, k, 1, o->outlier_contamination_temp, p.refine_scratch, s));
    }
    SG_TRY(layernorm(p.x_gem, D, c->lnpost_g, c->lnpost_b, p.xn, D, c->hk, R, D, 1e-5f, s));
  } else {
    const int mid = (L - 1) / 2;                                      // transformer.py:593
    const bool fusion = o->layer_fusion_enabled != 0;                 // transformer.py:598: takes precedence over the elif at :609
    const bool want_stats = o->outlier_enabled != 0 && !fusion;       // transformer.py:609 (R6)
    const float lf = o->layer_fusion_lambda;
    const int64_t BNN = (int64_t)B * N * N;
    bool x16 = false;                                                 // p.xn / p.ln_slice describe p.x (folded LayerNorm hand-off between blocks)
    for (int i = 0; i < L - 1; ++i) {
      if (i == mid && o->similarity_enabled)                          // normalised mid-layer patches (similarity_enhancement.py:49)
        SG_TRY(l2norm_rows(p.x + D, 0, (int64_t)N * D, D, n, p.xhat, c->hk, (int64_t)n * D, D, (int64_t)B * n, D, 1e-12f, s));
      SG_TRY(std_block(c, c->layers[i], p.x, p, B, N, want_stats && i == L - 2, s,
                       fusion || (want_stats && i == L - 2 && o->selfattn_enabled && o->selfattn_mode == 1), false, &x16));
      if (fusion) {                                                   // A_acc = lambda * A_acc + (1 - lambda) * A_l   (:601-607)
        if (i == 0) SG_HIP(hipMemcpyAsync(p.lf_acc, p.attn_avg, (size_t)BNN * 4, hipMemcpyDeviceToDevice, s));
        else SG_TRY(axpby(p.lf_acc, p.attn_avg, 1.0f - lf, lf, BNN, s));
      }
    }
    if (o->similarity_enabled)
      // the map stays f32 (similarity_enhancement.py computes it in fp32): a 2-byte map halves the attention kernel's bias fetch but was
      // measured no faster (round 2) -- the 'Experimental' kernel is bound by its two exponentials per score, not by the fetch
      SG_TRY(similarity_from_xhat(c->hk, p.xhat, B, n, D, o->similarity_temperature, o->similarity_add_self, p.sim, s));
    // ---- last block: self-self attention on ln_1(x), no residual / MLP when ignore_residual (transformer.py:627-643) ----
    const LayerW& LL = c->layers[L - 1];
    AttnBuffers ab{p.scores, p.probs, p.lse1, p.lse1, p.omega, p.qnorm, p.knorm};
    if (x16 && c->hk && !c->fp8 && LL.folded && p.ln_slice && R < (1ll << 31) && gemm_bf16_ln_fold_ok((int)R, D, D) && D % 256 == 0 &&
        gemm_bf16_prefers_persistent((int)R, D) && get_gemm_config() != 34) {                                    // block L-2's proj GEMM left x's 2-byte copy and statistics: ln_1 folded here too
      SG_TRY(ln_stats_finalize(p.ln_slice, R, D, 1e-5f, p.ln_rows, s));
      SG_TRY(linear_ln_consumer(c->hk, p.xn, D, LL.w_qkv_f, LL.bf_qkv, LL.c_qkv, p.ln_rows, p.qkv, 3 * D, (int)R, 3 * D, D, ACT_NONE, s));
    } else {
      SG_TRY(layernorm(p.x, D, LL.ln1_g, LL.ln1_b, p.xn, D, c->hk, R, D, 1e-5f, s));
      SG_TRY(linear(c->hk, p.xn, D, LL.w_qkv, LL.b_qkv, nullptr, p.qkv, 3 * D, false, (int)R, 3 * D, D, ACT_NONE, s));
    }
    if (fusion && o->ignore_residual) {                               // :630-637: the last block's own blk(x) attention joins the EMA
      if (!c->hk) {                                                   // parity mode: materialise the ordinary attention's probabilities
        AttnBuffers av{p.scores, p.probs, p.lse, p.lse1, p.omega, p.qnorm, p.knorm};
        SG_TRY(run_attention(c->hk, p.qkv, B, N, D, H, SG_VANILLA, nullptr, 0.f, nullptr, p.ctx, false, av, s));
      }
      SG_TRY(averaged_attention(c, p, B, N, s));
      if (L == 1) SG_HIP(hipMemcpyAsync(p.lf_acc, p.attn_avg, (size_t)BNN * 4, hipMemcpyDeviceToDevice, s));
      else SG_TRY(axpby(p.lf_acc, p.attn_avg, 1.0f - lf, lf, BNN, s));
    }
    const void* ctx = p.ctx; int64_t ctx_ld = D;
    if (o->model_type == SG_MASKCLIP) { ctx = (const char*)p.qkv + (size_t)2 * D * c->esz; ctx_ld = 3 * D; }   // identity attention: ctx = v
    else SG_TRY(run_attention(c->hk, p.qkv, B, N, D, H, o->model_type, (o->similarity_enabled && o->model_type < SG_NACLIP) ? p.sim : nullptr, o->similarity_weight,
                              nullptr, p.ctx, false, ab, s));
    SG_TRY(linear(c->hk, ctx, ctx_ld, LL.w_out, LL.b_out, o->ignore_residual ? nullptr : p.x, p.out_last, D, true, (int)R, D, D, ACT_NONE, s));
    if (!o->ignore_residual) SG_TRY(mlp_block(c, LL, p.out_last, p, R, s));
    // ---- attention-map layer fusion: mask the fused map's outlier columns, renormalise, re-weight every token (transformer.py:647-690) ----
    if (fusion && o->outlier_enabled) {
      const int k = o->outlier_top_k < n ? o->outlier_top_k : n;
      SG_TRY(fusion_row_diag(p.lf_acc, B, N, p.attn_cls, p.attn_diag, s));
      SG_TRY(select_topk(p.attn_cls, p.attn_diag, B, N, k, 0, p.idx_out, s));
      SG_TRY(fusion_mask_normalize(p.lf_acc, p.idx_out, B, N, k, s));
      GemmF32Args g{};
      g.A = p.lf_acc; g.lda = N; g.sAo = (int64_t)N * N; g.B = p.out_last; g.sbk = D; g.sbn = 1; g.sBo = (int64_t)N * D;
      g.C = p.sa_tmp; g.ldc = D; g.sCo = (int64_t)N * D; g.M = N; g.N = D; g.K = N; g.batch = B; g.inner = 1; g.act = 0; g.alpha = 1.f;
      SG_TRY(gemm_f32(g, s));
      SG_HIP(hipMemcpyAsync(p.out_last, p.sa_tmp, (size_t)R * D * 4, hipMemcpyDeviceToDevice, s));
    }
    // ---- refinements on the last-block output (transformer.py:698-742); need block L-2's attention, which layer fusion does not capture ----
    if (fusion) {
    } else if (o->outlier_enabled && o->selfattn_enabled && o->selfattn_mode == 1) {
      SG_TRY(attn_mode_enhance(p.out_last, (int64_t)N * D, D, p.attn_avg, B, N, D, o->selfattn_strength, o->selfattn_threshold, p.sa_tmp, s));
    } else if (o->outlier_enabled && o->selfattn_enabled) {
      const int k = o->selfattn_top_k < n ? o->selfattn_top_k : n;
      SG_TRY(select_topk(p.attn_cls, p.attn_diag, B, N, k, 1, p.idx_sa, s));
      SG_TRY(neighbour_refine(p.out_last, (int64_t)N * D, D, p.idx_sa, B, gh, gw, D, k, 0, 0.f, p.refine_scratch, s));
    }
    if (o->outlier_enabled && !fusion) {
      const int k = o->outlier_top_k < n ? o->outlier_top_k : n;
      SG_TRY(select_topk(p.attn_cls, p.attn_diag, B, N, k, 0, p.idx_out, s));
      SG_TRY(neighbour_refine(p.out_last, (int64_t)N * D, D, p.idx_out, B, gh, gw, D, k, 1, o->outlier_contamination_temp, p.refine_scratch, s));
    }
    SG_TRY(layernorm(p.out_last, D, c->lnpost_g, c->lnpost_b, p.xn, D, c->hk, R, D, 1e-5f, s));
  }
  // ---- epilogue: `@ proj` on every token (transformer.py:765-770) ----
  SG_TRY(linear(c->hk, p.xn, D, c->w_projT, nullptr, nullptr, p.y, E, true, (int)R, E, D, ACT_NONE, s));
  if (out_cls && !gem)
    SG_HIP(hipMemcpy2DAsync(out_cls, (size_t)E * 4, p.y, (size_t)N * E * 4, (size_t)E * 4, B, hipMemcpyDeviceToDevice, s));
  SG_HIP(hipMemcpy2DAsync(out_tokens, (size_t)n * E * 4, p.y + E, (size_t)N * E * 4, (size_t)n * E * 4, B, hipMemcpyDeviceToDevice, s));
  return SG_OK;
}

// GEM dual-stream blocks (reference gem/gem_utils.py:60-153).  p.x is the ordinary stream; p.x_gem the GEM stream.
static int gem_forward_tail(sg_context* c, const sg_forward_opts* o, const Plan& p, int B, int N, hipStream_t s) {
  const sg_vit_desc& d = c->d;
  const int D = d.width, H = d.heads, L = d.layers, dh = D / H;
  const int64_t R = (int64_t)B * N;
  const float scale = 1.0f / sqrtf((float)dh);
  const int first = L - (o->gem_depth - 1);
  const int bf = c->hk;
  SG_HIP(hipMemcpyAsync(p.x_gem, p.x, (size_t)R * D * 4, hipMemcpyDeviceToDevice, s));
  AttnBuffers ab{p.scores, p.probs, p.lse, p.lse1, nullptr, nullptr, nullptr};
  for (int i = first; i < L; ++i) {
    const LayerW& LW = c->layers[i];
    // ln_1(x): f32 copy for the temperature (mean token norm, gem_utils.py:79-81), compute-dtype copy for the GEMM
    SG_TRY(layernorm(p.x, D, LW.ln1_g, LW.ln1_b, p.gem_out, D, 0, R, D, 1e-5f, s));
    SG_TRY(gem_inv_temp(p.gem_out, B, N, D, scale, p.inv_temp, s));
    const void* xn = p.gem_out;
    if (bf) { SG_TRY(pack_rows(p.gem_out, R, D, D, p.xn, D, bf, s)); xn = p.xn; }
    SG_TRY(linear(bf, xn, D, LW.w_qkv, LW.b_qkv, nullptr, p.qkv, 3 * D, false, (int)R, 3 * D, D, ACT_NONE, s));
    // ordinary stream attention -> p.ctx (+ block L-2's head-averaged A[cls,:] / diag(A) when outlier suppression rides on the GEM forward)
    const bool stats = o->outlier_enabled != 0 && i == L - 2;
    SG_TRY(run_attention(bf, p.qkv, B, N, D, H, SG_VANILLA, nullptr, 0.f, nullptr, p.ctx, stats, ab, s));
    if (stats)
      SG_TRY(attention_stats(p.qkv, bf, (int64_t)N * 3 * D, 3 * D, p.lse, B, N, H, dh, scale, p.attn_cls, p.attn_diag, s));
    // GEM streams (v, k, q): normalise per head -> self-attend with values = the normalised vectors -> normalise
    const int64_t st3 = 3 * (int64_t)D;
    for (int t = 0; t < 3; ++t) {
      const char* src = (const char*)p.qkv + (size_t)(2 - t) * D * c->esz;
      SG_TRY(l2norm_rows(src, bf, st3, dh, H, p.gnorm[t], bf, D, dh, R * H, dh, 1e-12f, s));
      AttnSpec sp{};
      sp.q[0] = sp.k[0] = sp.v = p.gnorm[t]; sp.sb = sp.v_sb = (int64_t)N * D; sp.st = sp.v_st = D;
      sp.n_terms = 1; sp.scale = scale; sp.scale_per_image = p.inv_temp; sp.out_scale = 1.f;
      sp.ctx = p.gatt[t]; sp.ctx_sb = (int64_t)N * D; sp.ctx_st = D;
      SG_TRY(attn_generic(bf, sp, B, N, H, dh, ab, s));
      SG_TRY(l2norm_rows(p.gatt[t], bf, D, dh, H, p.gatt[t], bf, D, dh, R * H, dh, 1e-12f, s));
    }
    // assignment to V: mean of the three softmax(y y^T * inv_temp) . v   (gem_utils.py:101-117)
    AttnSpec sp{};
    for (int t = 0; t < 3; ++t) sp.q[t] = sp.k[t] = p.gatt[t];
    sp.sb = (int64_t)N * D; sp.st = D;
    sp.v = (const char*)p.qkv + (size_t)2 * D * c->esz; sp.v_sb = (int64_t)N * st3; sp.v_st = st3;
    sp.n_terms = 3; sp.scale = scale; sp.scale_per_image = p.inv_temp; sp.out_scale = 1.0f / 3.0f;
    sp.ctx = p.ctx2; sp.ctx_sb = (int64_t)N * D; sp.ctx_st = D;
    SG_TRY(attn_generic(bf, sp, B, N, H, dh, ab, s));
    // shared out_proj: GEM stream (residual optional, gem_utils.py:149-152), then the ordinary stream + MLP
    SG_TRY(linear(bf, p.ctx2, D, LW.w_out, LW.b_out, o->ignore_residual ? nullptr : p.x_gem, p.x_gem, D, true, (int)R, D, D, ACT_NONE, s));
    SG_TRY(linear(bf, p.ctx, D, LW.w_out, LW.b_out, p.x, p.x, D, true, (int)R, D, D, ACT_NONE, s));
    SG_TRY(mlp_block(c, LW, p.x, p, R, s));
  }
  return SG_OK;
}


// ---- CLIP text tower (reference open_clip/model.py:288-306 encode_text; init-time producer of query_features) --------------------
// Same residual blocks as the vision tower (nn.MultiheadAttention + MLP) with the causal mask of build_causal_mask, token +
// positional embedding in front, ln_final + EOT pooling (argmax of the token ids) + text_projection behind.
struct sg_text {
  sg_context core;                 // reuses the block machinery: width / heads / mlp / layers / precision live in core.d
  int context_length, vocab_size, embed_dim;
  float *tok_emb, *pos_emb, *lnf_g, *lnf_b;
  void* w_projT;                   // [E, W]
  std::vector<uint8_t> have_text;
};

__global__ __launch_bounds__(256) void text_embed_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos, int S, int ctx, int W, int vocab, float* __restrict__ x,
                                                         int* __restrict__ bad_id) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)S * ctx * W) return;
  const int c = (int)(i % W);
  const int64_t row = i / W;
  const int t = (int)(row % ctx);
  int id = tokens[row];
  if (id < 0 || id >= vocab) { if (c == 0) atomicMax(bad_id, 1); id = id < 0 ? 0 : vocab - 1; }   // reported by sg_text_encode; clamped only so the gather stays in bounds
  x[i] = tok_emb[(int64_t)id * W + c] + pos[(int64_t)t * W + c];
}
// pooled[s,:] = x[s, argmax_t tokens[s,t], :]   (text_global_pool 'argmax': the EOT token has the highest id; first maximum wins)
__global__ __launch_bounds__(256) void text_pool_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ x, int ctx, int W,
                                                        float* __restrict__ pooled) {
  __shared__ int s_arg;
  const int sidx = blockIdx.x;
  if (threadIdx.x == 0) {
    int best = tokens[(int64_t)sidx * ctx], arg = 0;
    for (int t = 1; t < ctx; ++t) { const int v = tokens[(int64_t)sidx * ctx + t]; if (v > best) { best = v; arg = t; } }
    s_arg = arg;
  }
  __syncthreads();
  const float* src = x + ((int64_t)sidx * ctx + s_arg) * W;
  for (int c = threadIdx.x; c < W; c += 256) pooled[(int64_t)sidx * W + c] = src[c];
}

extern "C" int sg_text_create(sg_text** out, int device, int width, int layers, int heads, int context_length, int vocab_size, int embed_dim,
                              int quick_gelu, int precision) {
  SG_REQUIRE(out && width > 0 && layers > 0 && heads > 0 && width % heads == 0 && context_length > 0 && vocab_size > 0 && embed_dim > 0,
             "sg_text_create: bad arguments");
  SG_REQUIRE(precision == SG_PREC_F32 || precision == SG_PREC_BF16 || precision == SG_PREC_F16 || precision == SG_PREC_F16X2, "sg_text_create: bad precision");
  SG_REQUIRE(width % 4 == 0 && embed_dim % 4 == 0, "sg_text_create: width / embed_dim must be multiples of 4");
  if (precision != SG_PREC_F32) {
    const int dh = width / heads;
    SG_REQUIRE(width % 64 == 0 && (dh == 32 || dh == 64 || dh == 80 || dh == 128), "sg_text_create: bf16 mode needs width %% 64 == 0 and head_dim 32/64/80/128");
  }
  DeviceGuard dg(device);
  sg_text* t = new sg_text();
  sg_context& c = t->core;
  c.d = sg_vit_desc{width, layers, heads, 1, embed_dim, 1, 4 * width, quick_gelu, precision};
  c.device = device; c.hk = hk_of_precision(precision); c.esz = hk_esz(c.hk); c.fp8 = false; c.Kpatch = c.Kpad = 0; c.finalized = false;
  t->context_length = context_length; t->vocab_size = vocab_size; t->embed_dim = embed_dim;
  auto lay = [&](Bump& bb) {
    const size_t e = c.esz; const int D = width, M = 4 * width;
    t->tok_emb = bb.get<float>((size_t)vocab_size * D); t->pos_emb = bb.get<float>((size_t)context_length * D);
    t->lnf_g = bb.get<float>(D); t->lnf_b = bb.get<float>(D); t->w_projT = bb.take((size_t)embed_dim * D * e);
    c.layers.resize(layers);
    for (auto& L : c.layers) {
      L.w_qkv = bb.take((size_t)3 * D * D * e); L.w_out = bb.take((size_t)D * D * e);
      L.w_fc = bb.take((size_t)M * D * e); L.w_proj = bb.take((size_t)D * M * e);
      L.b_qkv = bb.get<float>(3 * D); L.b_out = bb.get<float>(D); L.b_fc = bb.get<float>(M); L.b_proj = bb.get<float>(D);
      L.ln1_g = bb.get<float>(D); L.ln1_b = bb.get<float>(D); L.ln2_g = bb.get<float>(D); L.ln2_b = bb.get<float>(D);
    }
  };
  Bump dry(nullptr, 0, true); lay(dry);
  c.arena_bytes = align_up(dry.off, 256);
  hipError_t e = hipMalloc(&c.arena, c.arena_bytes);
  if (e != hipSuccess) { delete t; return fail(SG_ERR_HIP, "sg_text_create: hipMalloc(%zu) -> %s", c.arena_bytes, hipGetErrorString(e)); }
  Bump real(c.arena, c.arena_bytes, false); lay(real);
  t->have_text.assign(5 + 12 * layers, 0);
  *out = t;
  return SG_OK;
}

extern "C" void sg_text_destroy(sg_text* t) {
  if (!t) return;
  if (t->core.arena) (void)hipFree(t->core.arena);
  delete t;
}

extern "C" int sg_text_set_tensor(sg_text* t, const char* name, const float* src, int64_t numel, sg_stream st) {
  SG_REQUIRE(t && name && src, "sg_text_set_tensor: null argument");
  DeviceGuard dg(t->core.device);
  hipStream_t s = as_stream(st);
  sg_context& c = t->core;
  const int D = c.d.width, M = c.d.mlp_width, E = t->embed_dim;
  const int to_bf16 = c.hk;
  auto copyf = [&](float* dst, int64_t n) -> int {
    SG_REQUIRE(numel == n, "sg_text_set_tensor(%s): expected %lld elements, got %lld", name, (long long)n, (long long)numel);
    SG_HIP(hipMemcpyAsync(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    return SG_OK;
  };
  auto packw = [&](void* dst, int rows, int cols) -> int {
    SG_REQUIRE(numel == (int64_t)rows * cols, "sg_text_set_tensor(%s): expected %lld elements, got %lld", name, (long long)rows * cols, (long long)numel);
    return pack_rows(src, rows, cols, cols, dst, cols, to_bf16, s);
  };
  int slot = -1, rc = SG_OK;
  if (!strcmp(name, "token_embedding.weight")) { slot = 0; rc = copyf(t->tok_emb, (int64_t)t->vocab_size * D); }
  else if (!strcmp(name, "positional_embedding")) { slot = 1; rc = copyf(t->pos_emb, (int64_t)t->context_length * D); }
  else if (!strcmp(name, "ln_final.weight")) { slot = 2; rc = copyf(t->lnf_g, D); }
  else if (!strcmp(name, "ln_final.bias")) { slot = 3; rc = copyf(t->lnf_b, D); }
  else if (!strcmp(name, "text_projection")) {
    slot = 4;
    SG_REQUIRE(numel == (int64_t)D * E, "sg_text_set_tensor(text_projection): expected %d x %d", D, E);
    rc = transpose_pack(src, D, E, t->w_projT, to_bf16, s);
  } else {
    int li = -1, consumed = 0, tt;
    if (sscanf(name, "transformer.resblocks.%d.%n", &li, &consumed) == 1 && consumed > 0 && li >= 0 && li < c.d.layers &&
        find_layer_tensor(name + consumed, tt)) {
      LayerW& L = c.layers[li];
      slot = 5 + li * 12 + tt;
      switch (tt) {
        case 0: rc = copyf(L.ln1_g, D); break;
        case 1: rc = copyf(L.ln1_b, D); break;
        case 2: rc = packw(L.w_qkv, 3 * D, D); break;
        case 3: rc = copyf(L.b_qkv, 3 * D); break;
        case 4: rc = packw(L.w_out, D, D); break;
        case 5: rc = copyf(L.b_out, D); break;
        case 6: rc = copyf(L.ln2_g, D); break;
        case 7: rc = copyf(L.ln2_b, D); break;
        case 8: rc = packw(L.w_fc, M, D); break;
        case 9: rc = copyf(L.b_fc, M); break;
        case 10: rc = packw(L.w_proj, D, M); break;
        case 11: rc = copyf(L.b_proj, D); break;
      }
    }
  }
  if (slot < 0) return fail(SG_ERR_INVALID, "sg_text_set_tensor: unknown tensor name '%s'", name);
  if (rc != SG_OK) return rc;
  t->have_text[slot] = 1;
  return SG_OK;
}

static size_t text_plan(const sg_text* t, int S, void* ws, bool dry, Plan& p, float*& pooled, float*& x) {
  const sg_context& c = t->core;
  const int N = t->context_length, D = c.d.width;
  const int64_t R = (int64_t)S * N;
  Bump b(ws, 0, dry);
  x = b.get<float>(R * D);
  p.xn = b.take(R * D * c.esz); p.qkv = b.take(R * 3 * D * c.esz); p.ctx = b.take(R * D * c.esz); p.hbuf = b.take(R * c.d.mlp_width * c.esz);
  p.lse = b.get<float>((size_t)S * c.d.heads * N); p.lse1 = b.get<float>((size_t)S * c.d.heads * N);
  p.attn_cls = p.attn_diag = nullptr; p.omega = p.qnorm = p.knorm = nullptr; p.attn_avg = nullptr;
  p.scores = p.probs = nullptr;
  if (!c.hk) { p.scores = b.get<float>((size_t)S * c.d.heads * N * N); p.probs = b.get<float>((size_t)S * c.d.heads * N * N); }
  pooled = b.get<float>((size_t)S * D);
  p.idx_out = b.get<int32_t>(1);                            // out-of-vocabulary flag of text_embed_kernel
  return align_up(b.off, 256);
}

extern "C" size_t sg_text_workspace_bytes(const sg_text* t, int n_seq) {
  if (!t || n_seq <= 0) return 0;
  Plan p{}; float *a, *b;
  return text_plan(t, n_seq, nullptr, true, p, a, b);
}

// tokens int32 [S, context_length] (device) -> out [S, E] f32 (un-normalised, as encode_text(normalize=False))
extern "C" int sg_text_encode(sg_text* t, const int32_t* tokens, int n_seq, float* out, void* workspace, size_t workspace_bytes, sg_stream st) {
  SG_REQUIRE(t && tokens && out && workspace && n_seq > 0, "sg_text_encode: bad arguments");
  for (size_t i = 0; i < t->have_text.size(); ++i) if (!t->have_text[i]) return fail(SG_ERR_STATE, "sg_text_encode: text weights incomplete");
  DeviceGuard dg(t->core.device);
  hipStream_t s = as_stream(st);
  sg_context& c = t->core;
  const int N = t->context_length, D = c.d.width, E = t->embed_dim, S = n_seq;
  const int64_t R = (int64_t)S * N;
  Plan p{}; float *pooled, *x;
  const size_t need = text_plan(t, S, workspace, false, p, pooled, x);
  if (need > workspace_bytes) return fail(SG_ERR_STATE, "sg_text_encode: workspace %zu < required %zu", workspace_bytes, need);
  SG_HIP(hipMemsetAsync(p.idx_out, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)cdiv(R * D, 256)), dim3(256), 0, s, tokens, t->tok_emb, t->pos_emb, S, N, D, t->vocab_size, x, p.idx_out);
  SG_LAUNCH_CHECK();
  for (int i = 0; i < c.d.layers; ++i) SG_TRY(std_block(&c, c.layers[i], x, p, S, N, false, s, false, /*causal=*/true));
  hipLaunchKernelGGL(text_pool_kernel, dim3(S), dim3(256), 0, s, tokens, x, N, D, pooled);
  SG_LAUNCH_CHECK();
  SG_TRY(layernorm(pooled, D, t->lnf_g, t->lnf_b, p.xn, D, c.hk, S, D, 1e-5f, s));
  SG_TRY(linear(c.hk, p.xn, D, t->w_projT, nullptr, nullptr, out, E, true, S, E, D, ACT_NONE, s));
  // init-time call: the one entry point that synchronises, so that an id outside the vocabulary is an error as in the reference
  // (nn.Embedding raises, open_clip/model.py:292) instead of a silently clamped row
  int32_t bad = 0;
  SG_HIP(hipMemcpyAsync(&bad, p.idx_out, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  SG_HIP(hipStreamSynchronize(s));
  if (bad) return fail(SG_ERR_INVALID, "sg_text_encode: token id outside [0, %d)", t->vocab_size);
  return SG_OK;
}

// ---- stand-alone ops ------------------------------------------------------------------------------------------------------
extern "C" int sg_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int rows, int D, float eps, sg_stream s) {
  SG_REQUIRE(x && gamma && beta && y, "sg_op_layernorm: null pointer");
  return layernorm(x, D, gamma, beta, y, D, 0, rows, D, eps, as_stream(s));
}

extern "C" int sg_op_linear(const float* A, const float* W, const float* bias, const float* residual, float* C, int M, int N, int K,
                            int act, int precision, void* scratch, size_t scratch_bytes, sg_stream st) {
  SG_REQUIRE(A && W && C, "sg_op_linear: null pointer");
  hipStream_t s = as_stream(st);
  if (precision == SG_PREC_F32) return linear(HK_F32, A, K, W, bias, residual, C, N, true, M, N, K, act, s);
  const int hk = hk_of_precision(precision);
  const size_t e = hk_esz(hk);
  const int Kp = (int)align_up(K, 64);
  const size_t need = align_up((size_t)M * Kp * e, 256) + (size_t)N * Kp * e;
  if (!scratch || scratch_bytes < need) return fail(SG_ERR_STATE, "sg_op_linear: scratch %zu < %zu", scratch_bytes, need);
  bf16_t* a16 = (bf16_t*)scratch;
  bf16_t* w16 = (bf16_t*)((char*)scratch + align_up((size_t)M * Kp * e, 256));
  SG_TRY(pack_rows(A, M, K, K, a16, Kp, hk, s));
  SG_TRY(pack_rows(W, N, K, K, w16, Kp, hk, s));
  return linear(hk, a16, Kp, w16, bias, residual, C, N, true, M, N, Kp, act, s);
}

// x_new = x + A.W1^T + b1;  y = act(LayerNorm(x_new; gamma, beta).W2^T + b2)  -- the residual GEMM -> LayerNorm -> GEMM chain of a block
// (out-proj -> ln_2 -> fc, proj -> ln_1 -> QKV), either with the LayerNorm as its own pass (fold = 0) or folded into the two GEMMs
// (fold = 1: 2-byte copy + slice statistics out of the first epilogue, (mean, rstd) and the gamma-folded weight in the second).
// 2-byte and two-plane precisions; M >= 1024, D >= 512, D % 64 == 0, N2 >= 512 with fold.  All operands f32 on the device.
extern "C" size_t sg_op_ln_chain_scratch_bytes(int M, int K1, int D, int N2) {
  const size_t K1p = align_up((size_t)K1, 64);
  return align_up((size_t)M * K1p * 4, 256) + align_up((size_t)D * K1p * 4, 256) + align_up((size_t)M * D * 4, 256) + align_up((size_t)N2 * D * 4, 256) +
         align_up((size_t)M * N2 * 4, 256) + align_up((size_t)M * (D / 64 + 1) * 8, 256) + align_up((size_t)M * 8, 256) + 2 * align_up((size_t)N2 * 4, 256) + 4096;
}
extern "C" int sg_op_ln_chain(const float* A, const float* W1, const float* b1, float* x, const float* gamma, const float* beta, const float* W2,
                              const float* b2, float* y, int M, int K1, int D, int N2, int act, int precision, int fold, void* scratch,
                              size_t scratch_bytes, sg_stream st) {
  SG_REQUIRE(A && W1 && x && gamma && beta && W2 && y && scratch, "sg_op_ln_chain: null pointer");
  SG_REQUIRE(precision == SG_PREC_BF16 || precision == SG_PREC_F16 || precision == SG_PREC_F16X2, "sg_op_ln_chain: 2-byte and two-plane precisions only");
  SG_REQUIRE(D % 64 == 0, "sg_op_ln_chain: D %% 64 != 0");
  if (scratch_bytes < sg_op_ln_chain_scratch_bytes(M, K1, D, N2)) return fail(SG_ERR_STATE, "sg_op_ln_chain: scratch too small");
  hipStream_t s = as_stream(st);
  const int hk = hk_of_precision(precision);
  const size_t e = hk_esz(hk);
  const int K1p = (int)align_up(K1, 64);
  Bump b(scratch, 0, false);
  void* a16 = b.take((size_t)M * K1p * e); void* w116 = b.take((size_t)D * K1p * e); void* xn = b.take((size_t)M * D * e);
  void* w216 = b.take((size_t)N2 * D * e); void* y16 = b.take((size_t)M * N2 * e);
  float* slice = b.get<float>((size_t)M * (D / 64) * 2); float* rows = b.get<float>((size_t)M * 2);
  float* cvec = b.get<float>(N2); float* bf = b.get<float>(N2);
  SG_TRY(pack_rows(A, M, K1, K1, a16, K1p, hk, s));
  SG_TRY(pack_rows(W1, D, K1, K1, w116, K1p, hk, s));
  if (fold) {
    SG_TRY(linear_ln_producer(hk, a16, K1p, w116, b1, x, x, D, xn, slice, M, D, K1p, s));
    SG_TRY(ln_stats_finalize(slice, M, D, 1e-5f, rows, s));
    SG_TRY(fold_ln_weight(W2, N2, D, gamma, beta, b2, hk, w216, cvec, bf, s));
    SG_TRY(linear_ln_consumer(hk, xn, D, w216, bf, cvec, rows, y16, N2, M, N2, D, act, s));
  } else {
    SG_TRY(linear(hk, a16, K1p, w116, b1, x, x, D, true, M, D, K1p, ACT_NONE, s));
    SG_TRY(layernorm(x, D, gamma, beta, xn, D, hk, M, D, 1e-5f, s));
    SG_TRY(pack_rows(W2, N2, D, D, w216, D, hk, s));
    SG_TRY(linear(hk, xn, D, w216, b2, nullptr, y16, N2, false, M, N2, D, act, s));
  }
  const int64_t total = (int64_t)M * N2;
  hipLaunchKernelGGL(unpack_bf16_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, (const bf16_t*)y16, y, total, hk);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

extern "C" size_t sg_op_attention_scratch_bytes(int B, int N, int D, int H, int precision) {
  const size_t R = (size_t)B * N;
  size_t b = 4 * 256 + 2 * align_up((size_t)B * H * N * 4, 256) + align_up((size_t)(N - 1) * (N - 1) * 4, 256) + 2 * align_up((size_t)B * H * N * 4, 256);
  if (precision != SG_PREC_F32) b += align_up(R * 3 * D * hk_esz(hk_of_precision(precision)), 256) + align_up(R * D * hk_esz(hk_of_precision(precision)), 256);
  else b += 2 * align_up((size_t)B * H * N * N * 4, 256);
  return b + 4096;
}

extern "C" int sg_op_attention(const float* qkv, int B, int N, int D, int H, int variant, const float* sim, float sim_weight, float* ctx,
                               float* attn_cls, float* attn_diag, int precision, void* scratch, size_t scratch_bytes, sg_stream st) {
  SG_REQUIRE(qkv && ctx && scratch, "sg_op_attention: null pointer");
  SG_REQUIRE(D % H == 0, "sg_op_attention: D %% H != 0");
  hipStream_t s = as_stream(st);
  const int64_t R = (int64_t)B * N;
  const int bf = hk_of_precision(precision);
  Bump b(scratch, scratch_bytes, false);
  AttnBuffers ab{};
  ab.lse = b.get<float>((size_t)B * H * N); ab.lse1 = b.get<float>((size_t)B * H * N);
  ab.omega = b.get<float>((size_t)(N - 1) * (N - 1)); ab.qnorm = b.get<float>((size_t)B * H * N); ab.knorm = b.get<float>((size_t)B * H * N);
  void* qkv_c = (void*)qkv; void* ctx_c = ctx;
  if (bf) { qkv_c = b.take((size_t)R * 3 * D * hk_esz(bf)); ctx_c = b.take((size_t)R * D * hk_esz(bf)); }
  else { ab.scores = b.get<float>((size_t)B * H * N * N); ab.probs = b.get<float>((size_t)B * H * N * N); }
  if (b.off > scratch_bytes) return fail(SG_ERR_STATE, "sg_op_attention: scratch %zu < %zu", scratch_bytes, b.off);
  if (bf) SG_TRY(pack_rows(qkv, R, 3 * D, 3 * D, qkv_c, 3 * D, bf, s));
  const bool stats = attn_cls && attn_diag;
  if (variant == SG_MASKCLIP) return fail(SG_ERR_INVALID, "sg_op_attention: MaskCLIP is the identity (ctx = v)");
  SG_TRY(run_attention(bf, qkv_c, B, N, D, H, variant, sim, sim_weight, nullptr, ctx_c, stats, ab, s));
  if (stats) SG_TRY(attention_stats(qkv_c, bf, (int64_t)N * 3 * D, 3 * D, ab.lse, B, N, H, D / H, 1.0f / sqrtf((float)(D / H)), attn_cls, attn_diag, s));
  if (bf) {
    hipLaunchKernelGGL(unpack_bf16_kernel, dim3((unsigned)cdiv(R * D, 256)), dim3(256), 0, s, (const bf16_t*)ctx_c, ctx, R * D, bf);
    SG_LAUNCH_CHECK();
  }
  return SG_OK;
}

extern "C" int sg_similarity_map(const float* patches, int64_t batch_stride, int ld, int B, int n, int D, float temperature, int add_self,
                                 int precision, float* sim, void* scratch, size_t scratch_bytes, sg_stream st) {
  SG_REQUIRE(patches && sim && scratch, "sg_similarity_map: null pointer");
  hipStream_t s = as_stream(st);
  const int bf = hk_of_precision(precision);
  const size_t need = (size_t)B * n * D * hk_esz(bf);
  if (scratch_bytes < need) return fail(SG_ERR_STATE, "sg_similarity_map: scratch %zu < %zu", scratch_bytes, need);
  if (bf) SG_REQUIRE(D % 64 == 0, "sg_similarity_map: bf16 mode needs D %% 64 == 0");
  SG_TRY(l2norm_rows(patches, 0, batch_stride, ld, n, scratch, bf, (int64_t)n * D, D, (int64_t)B * n, D, 1e-12f, s));
  return similarity_from_xhat(bf, scratch, B, n, D, temperature, add_self, sim, s);
}

extern "C" size_t sg_outlier_scratch_bytes(int B, int D, int k) { return refine_scratch_bytes(B, D, k); }

extern "C" int sg_outlier_suppress(float* feats, const float* attn_cls, const float* attn_diag, int B, int gh, int gw, int D, int top_k,
                                   float contamination_temp, int32_t* out_idx, void* scratch, sg_stream st) {
  SG_REQUIRE(feats && attn_cls && attn_diag && out_idx && scratch, "sg_outlier_suppress: null pointer");
  hipStream_t s = as_stream(st);
  const int n = gh * gw, N = n + 1, k = top_k < n ? top_k : n;
  SG_TRY(select_topk(attn_cls, attn_diag, B, N, k, 0, out_idx, s));
  // feats is patch-only [B,n,D]: token t of the kernels = 1 + cell, so shift the base by one row
  return neighbour_refine(feats - D, (int64_t)n * D, D, out_idx, B, gh, gw, D, k, 1, contamination_temp, scratch, s);
}

extern "C" int sg_weak_token_replace(float* feats, const float* attn_diag, int B, int gh, int gw, int D, int top_k, int32_t* out_idx,
                                     void* scratch, sg_stream st) {
  SG_REQUIRE(feats && attn_diag && out_idx && scratch, "sg_weak_token_replace: null pointer");
  hipStream_t s = as_stream(st);
  const int n = gh * gw, N = n + 1, k = top_k < n ? top_k : n;
  SG_TRY(select_topk(attn_diag, attn_diag, B, N, k, 1, out_idx, s));
  return neighbour_refine(feats - D, (int64_t)n * D, D, out_idx, B, gh, gw, D, k, 0, 0.f, scratch, s);
}
