// Internal prototypes of the non-GEMM ops (definitions: rowops.hip, attention.hip, patchify.hip,
// refine.hip, head.hip, jbu.hip).
#pragma once
#include "common.h"

namespace sg {

int layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy, int y_is_bf16,
              int64_t rows, int D, float eps, hipStream_t s);
// fp8 (OCP e4m3) operands of the fp8 GEMMs: per-row absmax scale (scale[r] = max|row| / 448)
int layernorm_fp8(const float* x, int64_t ldx, const float* gamma, const float* beta, uint8_t* y, int64_t ldy, float* scale, int64_t rows,
                  int D, float eps, hipStream_t s);
int quantize_rows_fp8(const void* x, int x_is_bf16, int64_t ldx, uint8_t* y, int64_t ldy, float* scale, int64_t rows, int D, hipStream_t s);
int embed_assemble(const float* patches, int64_t ldp, const float* cls_emb, const float* pos, const float* gamma,
                   const float* beta, float* x, int B, int N, int D, float eps, hipStream_t s);
int posembed_resize(const float* pos, int g0, int D, int gh, int gw, int antialias, float* out, hipStream_t s);
int pack_rows(const float* src, int64_t rows, int cols, int64_t ld_src, void* dst, int cols_pad, int to_bf16, hipStream_t s);
// LayerNorm folded into its neighbouring GEMMs (GemmBf16Args::copy16 / ln_stats):
//   ln_stats_finalize: the producer's slice statistics [rows][D/64][2] (sum, centred sum of squares) -> (mean, rstd) per row [rows][2]
//   fold_ln_weight:    W' = gamma o W packed in the compute dtype, c[n] = sum_k W'[n][k] (of the ROUNDED W', so that x.W'^T - mean c cancels
//                      exactly), b'[n] = b[n] + sum_k beta[k] W[n][k]
int ln_stats_finalize(const float* slice_stats, int64_t rows, int D, float eps, float* mean_rstd, hipStream_t s);
int fold_ln_weight(const float* W, int N, int K, const float* gamma, const float* beta, const float* bias, int hk, void* Wp, float* c,
                   float* bias_f, hipStream_t s);
int transpose_pack(const float* src, int rows, int cols, void* dst, int to_bf16, hipStream_t s);
int l2norm_rows(const void* x, int x_bf16, int64_t so, int64_t si, int inner, void* y, int y_bf16, int64_t yo, int64_t yi,
                int64_t rows, int D, float eps, hipStream_t s);
int softmax_rows(const float* scores, int64_t ld, int64_t rows, int N, int H, const float* scale_per_image, float scale,
                 const float* bias, float bias_w, int64_t bias_bstride, const float* bias_rn, const float* bias_cn, int mode,
                 int accumulate, float* out, float* lse, hipStream_t s, int causal = 0);
int gaussian_bias(int gh, int gw, float std, float* omega, hipStream_t s);
int head_norms(const void* x, int is_bf16, int64_t sb, int64_t st, int B, int N, int H, int dh, float* out, hipStream_t s);
int axpby(float* y, const float* x, float a, float b, int64_t n, hipStream_t s);
int gem_inv_temp(const float* x, int B, int N, int D, float scale, float* out, hipStream_t s);

// ---- attention.hip: fused (flash-style) multi-term attention, bf16 MFMA -----------------------------
// ctx[b, i, h*dv : (h+1)*dv] = out_scale * sum_over_streams softmax_j( score_s(i, j) ) . V[b, j, h, :]
//   score_s = scale * sum_{terms of the stream} Q_t[i] . K_t[j]   (+ bias_w * bias[b, i-1, j-1])
// With sum_scores = 1 all terms form ONE stream (SFP, Experimental); otherwise each term is its own stream
// (vanilla: 1; SCLIP: 2; SegEarth / GEM: 3).
struct AttnArgs {
  const bf16_t* q[3]; const bf16_t* k[3];   // per term; element (b, t, h, d) at p + b*sb + t*st + h*dh + d
  const bf16_t* v;
  int64_t sb, st;                           // batch / token strides (elements) shared by every q and k term
  int64_t v_sb, v_st;                       // batch / token strides of v
  int n_terms, sum_scores;
  int B, N, H, dh;
  float scale; const float* scale_per_image;  // per-image scale overrides `scale` when non-null (GEM inv_temp)
  const float* bias; float bias_w;          // [B, N-1, N-1] symmetric, or null; batch stride bias_bstride (0 = shared by all images)
  int64_t bias_bstride;
  const float* bias_rn; const float* bias_cn; // optional [B,H,N] row / column factors of the bias (NOnly / GAV: |q_i|, |k_j|)
  int causal;                               // keys after the query are masked (CLIP text tower)
  int resoftmax;                            // 'Experimental': softmax(softmax(score) + bias_w*bias); needs lse_in
  const float* lse_in;                      // [B,H,N] log-sum-exp of the first softmax (resoftmax)
  float* lse_out;                           // [B,H,N] or null; when ctx == null only the LSE pass runs
  bf16_t* ctx; int64_t ctx_sb, ctx_st;      // output [B,N,H*dv] bf16
  float out_scale;
  int f16;                                  // operands and ctx are IEEE f16 instead of bf16 (SG_PREC_F16)
  int h2;                                   // SG_PREC_F16X2: operands and ctx are two-plane f16; every stride above is then in f16 UNITS (2 x the element stride)
};
int attention_bf16(const AttnArgs& a, hipStream_t s);

// head-averaged attention statistics of an ordinary block (outlier detection needs only these):
//   attn_cls[b, j] = mean_h softmax(q k^T)[0, j],  attn_diag[b, j] = mean_h softmax(q k^T)[j, j]
// from q, k (packed qkv) and the per-row log-sum-exp.  T = bf16_t or float.
int attention_stats(const void* qkv, int is_bf16, int64_t sb, int64_t st, const float* lse, int B, int N, int H, int dh,
                    float scale, float* attn_cls, float* attn_diag, hipStream_t s);

// ---- patchify.hip ------------------------------------------------------------------------------------
int patchify(const sg_tile_batch& t, int P, void* out, int Kpad, int out_bf16, hipStream_t s);

// ---- refine.hip --------------------------------------------------------------------------------------
int select_topk(const float* attn_cls, const float* attn_diag, int B, int N, int k, int mode, int32_t* idx, hipStream_t s);
size_t refine_scratch_bytes(int B, int D, int k);
int neighbour_refine(float* tokens, int64_t sb, int64_t st, const int32_t* idx, int B, int gh, int gw, int D, int k,
                     int decontaminate, float contamination_temp, void* scratch, hipStream_t s);
int head_mean(const float* probs, int B, int H, int N, float* A, hipStream_t s);
int fusion_row_diag(const float* A, int B, int N, float* a_cls, float* a_diag, hipStream_t s);
int fusion_mask_normalize(float* A, const int32_t* idx, int B, int N, int k, hipStream_t s);
int attn_mode_enhance(float* tokens, int64_t sb, int64_t st, float* A, int B, int N, int D, float strength, float threshold, float* tmp,
                      hipStream_t s);

}  // namespace sg
