/*
 * segearth_hip.h -- C ABI of libsegearth_hip.so: the MI355X (gfx950) implementation of the
 * sliding-window CLIP dense-feature path of CLIP-Decontamination / SegEarth-OV.
 *
 * This is the drop-in boundary.  Every entry point replaces one seam of the
 * reference's Python hot path; the reference file:line it stands in for is cited on each.
 * Conventions
 *   - plain C types only: device pointers, sizes, a hipStream_t passed as void*;
 *   - the CALLER owns every input, output and workspace buffer (device memory); the library
 *     allocates only in sg_create / sg_vit_set_tensor / sg_jbu_set_tensor (packed weights) and
 *     frees in sg_destroy; no hidden per-call hipMalloc, no host synchronisation in any call
 *     that takes a stream;
 *   - every call returns 0 on success or a negative sg_status; the message of the last
 *     failure on the calling thread is returned by sg_last_error(); nothing aborts or throws;
 *   - calls are asynchronous on the given stream and re-entrant across contexts: the library keeps no mutable per-process
 *     state (per-device launch bookkeeping is keyed by device and thread-safe; the measurement and tuning hooks below act on
 *     the calling thread only); an entry point that takes a context makes the context's device current for its duration, the
 *     context-free ops run on the caller's current device (the one the stream belongs to).
 *   - "tokens" are token-major [B, N, D] row-major (N = 1 + gh*gw, CLS first).
 */
#ifndef SEGEARTH_HIP_H
#define SEGEARTH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sg_context sg_context;
typedef void* sg_stream;            /* hipStream_t */

enum sg_status {
  SG_OK = 0,
  SG_ERR_INVALID = -1,    /* bad argument / unsupported shape */
  SG_ERR_HIP = -2,        /* a HIP runtime call failed */
  SG_ERR_STATE = -3,      /* weights missing, text not set, workspace too small ... */
};

enum sg_precision {       /* arithmetic the ViT GEMMs / attention run in */
  SG_PREC_F32 = 0,        /* parity mode: f32 MFMA (exact fmaf chains), materialised attention */
  SG_PREC_BF16 = 1,       /* throughput mode: bf16 MFMA, f32 accumulate, f32 residual stream + LN */
  SG_PREC_FP8 = 2,        /* bf16 mode whose QKV / fc / proj linears of the ordinary blocks run on fp8 (OCP e4m3) MFMA with per-token and
                             per-output-channel absmax scales; attention, out-proj, the last block and everything else stay bf16 / f32 */
  SG_PREC_F16 = 3,        /* throughput mode on IEEE f16 operands (v_mfma_*_f16: the bf16 rate, 3 more mantissa bits): the reference's own
                             GPU arithmetic (segmentor.py:467 .half(), open_clip/model.py:142 fp32 LayerNorm).  f32 accumulate, f32
                             residual stream / LN / softmax statistics as in bf16 mode; stores saturate at +-65504 */
  SG_PREC_F16X2 = 4,      /* parity AT SPEED: every GEMM / attention operand is held as two f16 planes x = hi + lo (22 significant bits) and every
                             product is issued as hi.hi + hi.lo + lo.hi on the f16 matrix pipe into one f32 accumulator -- the error of an f32
                             fmaf chain (tools/h2_probe.hip) at a third of the f16 MFMA rate instead of the 1/16 of the f32 MFMA.  f32 residual
                             stream / LayerNorm / softmax statistics, exact expf / erff activations.  Logits within 1e-3 of the reference's fp32 CPU
                             path with arg-max identical up to fp32 ties, like SG_PREC_F32 (same reference bar: segmentor.py:467 .half() is the
                             reference's own GPU arithmetic; its CPU path, the oracle, is fp32).  Values beyond +-131008 are not representable. */
};

/* last-block attention variants: reference open_clip/transformer.py:858-932 (custom_attn),
 * SG_GEM: reference gem/gem_utils.py:60-199 */
enum sg_model_type {
  SG_VANILLA = 0, SG_MASKCLIP = 1, SG_CLEARCLIP = 2, SG_SCLIP = 3, SG_SEGEARTH = 4, SG_SFP = 5,
  SG_EXPERIMENTAL = 6, SG_NACLIP = 7, SG_NONLY = 8, SG_GAV = 9, SG_GEM = 10,
};

enum sg_image_format {
  SG_IMG_F32_NCHW = 0,    /* normalised float planes: what predict() receives (segmentor.py:453-467) */
  SG_IMG_U8_NHWC = 1,     /* raw RGB bytes; (x-mean)/std of segmentor.py:64-67 fused into the load */
};

/* Architecture of the vision tower: reference open_clip/model_configs/ViT-*.json,
 * open_clip/transformer.py:341-362 */
typedef struct sg_vit_desc {
  int32_t width;          /* D */
  int32_t layers;         /* L */
  int32_t heads;          /* H */
  int32_t patch;          /* P */
  int32_t embed_dim;      /* E */
  int32_t grid0;          /* native positional grid side (image_size / patch) */
  int32_t mlp_width;      /* 4D */
  int32_t quick_gelu;     /* 1: x*sigmoid(1.702x) (transformer.py:35-38), 0: exact erf GELU */
  int32_t precision;      /* enum sg_precision */
} sg_vit_desc;

/* Knobs of one forward: the kwargs of VisionTransformer.forward (transformer.py:538) plus the
 * refiner modules the reference hangs on net.visual (segmentor.py:196-274). */
typedef struct sg_forward_opts {
  int32_t model_type;             /* enum sg_model_type */
  int32_t ignore_residual;        /* transformer.py:627-643 */
  int32_t similarity_enabled;     /* similarity_enhancement.py; 0 = module not installed */
  float   similarity_weight;
  float   similarity_temperature;
  int32_t similarity_add_self;
  int32_t outlier_enabled;        /* outlier_suppression.py; captures block L-2 attention */
  int32_t outlier_top_k;
  float   outlier_contamination_temp;
  int32_t selfattn_enabled;       /* self_attention_enhancement.py (inert unless outlier_enabled, R6) */
  int32_t selfattn_mode;          /* 0 feature, 1 attention */
  int32_t selfattn_top_k;
  float   selfattn_strength;
  float   selfattn_threshold;
  int32_t gem_depth;              /* GEM: blocks -1..-(depth-1) are dual-stream (gem_wrapper.py:24-45) */
  int32_t layer_fusion_enabled;   /* apply_layer_fusion (transformer.py:598-607,630-637,647-690): EMA of the head-averaged attention of every
                                     block; with an outlier suppressor its top_k columns are zeroed, rows L1-renormalised and
                                     output = attn @ output -- the refiners are then skipped, as in the reference.  The reference's
                                     view(N, heads, L, L) only runs for heads == 1; that case pins the semantics. */
  float   layer_fusion_lambda;
} sg_forward_opts;

/* Where the tiles of one launch come from: one scene + a window per tile, so that cropping,
 * zero padding to a patch multiple (segmentor.py:418-431, 534-546) and im2col happen on the
 * device in one pass. */
typedef struct sg_tile_batch {
  const void* scene;              /* device: [C=3,H,W] f32 planes or [H,W,3] u8 */
  int32_t format;                 /* enum sg_image_format */
  int32_t scene_h, scene_w;
  const int32_t* windows;         /* device int32 [n_tiles][4] = y1,y2,x1,x2 */
  const int32_t* scene_index;     /* device int32 [n_tiles] image index of each tile, or NULL (all tiles from image 0) */
  int64_t scene_stride;           /* elements between consecutive images when `scene` holds a batch [B,3,H,W] / [B,H,W,3] */
  int32_t n_tiles;
  int32_t tile_h, tile_w;         /* window size (all tiles of a launch share it) */
  int32_t pad_l, pad_t;           /* compute_padsize(): zeros added left / top */
  int32_t grid_h, grid_w;         /* (tile + pad) / patch */
} sg_tile_batch;

const char* sg_last_error(void);
int sg_version(void);

/* ---- live kernel timing (measurement only; used by bench.py's roofline) -------------------------
 * HIP events bracket every launch of a kernel family on the stream it is launched on.
 * category: 0 = bf16 MFMA GEMM (small-shape tile variants), 1 = fused attention, 2 = f32 MFMA GEMM,
 *           3 = the persistent bf16 GEMM (every large ViT linear; the kernel bench.py's roofline prices: all three instantiations),
 *           4 = fp8 GEMM, 5 / 6 = the persistent GEMM's folded-LayerNorm consumer / producer instantiations alone (subsets of 3).
 *           Read after synchronising.  The state belongs to the calling thread (enable, launch and read on one thread). */
int sg_profile_enable(int capacity);
int sg_profile_disable(void);
int sg_set_gemm_config(int cfg);   /* tuning hook (calling thread only): bf16 GEMM tile variant, -1 = automatic; 33 = fp8 MLP without the MXFP8
                                    * hand-off, 34 = LayerNorm as its own pass (no folding), 36 = no small-launch dispatch (a few-tile GEMM stays on the
                                    * persistent kernel); 1000+ = tile order of the persistent kernel */
int sg_profile_read(int category, double* total_ms, double* total_flops, int64_t* launches, int64_t* dropped);

/* ---- context and weights ------------------------------------------------------------------
 * sg_create replaces create_model(...) + .eval().to(device) (segmentor.py:69-131) for the
 * vision tower only; weights arrive by their visual.* state-dict names
 * (e.g. "transformer.resblocks.3.attn.in_proj_weight", open_clip/transformer.py:372-442). */
int  sg_create(sg_context** out, int device, const sg_vit_desc* desc);
void sg_destroy(sg_context* ctx);
int  sg_vit_set_tensor(sg_context* ctx, const char* name, const float* dev_f32, int64_t numel, sg_stream s);
int  sg_vit_finalize(sg_context* ctx, sg_stream s);

/* ---- the vision tower -----------------------------------------------------------------------
 * sg_vit_forward replaces net.encode_image(img, model_type, ignore_residual, output_cls_token=True, ...)
 * (open_clip/model.py:265-286 -> open_clip/transformer.py:538-775) and, for SG_GEM,
 * net.visual(img) (gem/gem_utils.py:159-199).
 *   out_cls    [B,E] f32 (untouched for SG_GEM), out_tokens [B,gh*gw,E] f32. */
size_t sg_vit_workspace_bytes(const sg_context* ctx, int n_tiles, int grid_h, int grid_w, const sg_forward_opts* o);
int sg_vit_forward(sg_context* ctx, const sg_tile_batch* tiles, const sg_forward_opts* o,
                   float* out_cls, float* out_tokens, void* workspace, size_t workspace_bytes, sg_stream s);

/* ---- segmentation head ----------------------------------------------------------------------
 * sg_cosine_logits replaces segmentor.py:309-336,374-386: CLS normalise + cls_logits, global
 * debias tokens -= cls * (cos(tokens,cls) * factor), L2 normalise, tokens @ T^T, + lambda*cls_logits.
 *   tokens [B,n,E], cls [B,E] (may be NULL when both factors are 0), text [Q,E] -> logits [B,Q,n]. */
int sg_cosine_logits(const float* tokens, const float* cls, const float* text, int B, int n, int E, int Q,
                     float global_debias_factor, float cls_token_lambda, float* logits, sg_stream s);
/* The per-pixel logits behind the upsampler (segmentor.py:374-379, no global debias: it ran before the upsampler) for the exact tower mode
 * SG_PREC_F16X2: the [n, E] x [E, Q] product on the f16 matrix pipe with both operands as two f16 planes (f32-grade results, three MFMAs per
 * product; magnitudes beyond +-131 008 saturate, as everywhere in that mode).  Same layouts as sg_cosine_logits; shapes it does not take
 * (Q > 16, E % 32 != 0, n < 4096) are forwarded to sg_cosine_logits. */
int sg_cosine_logits_two_plane(const float* tokens, const float* cls, const float* text, int B, int n, int E, int Q,
                               float cls_token_lambda, float* logits, sg_stream s);

/* sg_stitch replaces the bilinear upsample + un-pad + overlap-add + count-normalise of
 * segmentor.py:388-391,436-447 in a write-once form: canvas[q,y,x] = mean over covering tiles
 * (raster order) of bilinear(tile_logits)[y - y1 + pad_t, x - x1 + pad_l].
 *   tile_logits [T,Q,gh,gw] f32, windows int32 [T][4] (y1,y2,x1,x2), canvas [Q,H,W] f32.
 *   up_h/up_w: the size tiles are bilinearly resized to (tile + pad). */
int sg_stitch(const float* tile_logits, const int32_t* windows, int T, int Q, int gh, int gw,
              int up_h, int up_w, int pad_t, int pad_l, int H, int W, float* canvas, sg_stream s);

/* F.interpolate(mode='bilinear', align_corners=False) on [C,h,w] -> [C,H,W] (segmentor.py:389,449) */
int sg_resize_bilinear(const float* src, int C, int h, int w, float* dst, int H, int W, sg_stream s);

/* sg_postprocess replaces postprocess_result (segmentor.py:475-489): x logit_scale, softmax over
 * queries, per-class max over synonyms, argmax, prob_thd -> bg_idx.
 *   logits [Q,H,W]; query_idx int32 [Q]; probs [K,H,W] f32 (may be NULL); labels int64 [H,W]. */
int sg_postprocess(const float* logits, const int32_t* query_idx, int Q, int K, int H, int W, float logit_scale,
                   float prob_thd, int bg_idx, float* probs, int64_t* labels, sg_stream s);
/* Label / confidence images of postprocess_result (segmentor.py:501-531): mask_rgb [H,W,3] = palette[clip(label)] (_colorize_mask,
 * :580-590); heat_rgb [H,W,3] = (g, 0, 255-g) with g = uint8(clip(max_k probs, 0, 1) * 255) (_to_colormap without OpenCV, :604-608;
 * OpenCV's JET table is not reproduced).  Either output may be NULL. */
int sg_render_maps(const int64_t* labels, const float* probs, const uint8_t* palette, int K, int H, int W, uint8_t* mask_rgb,
                   uint8_t* heat_rgb, sg_stream s);

/* ---- token refinements as stand-alone ops (same arithmetic as inside sg_vit_forward) ----------
 * sg_outlier_suppress replaces OutlierSuppressionModule.forward (outlier_suppression.py:83-214):
 *   feats [B,gh*gw,D] f32 in place; attn_cls [B,N] = head-averaged A[0,:], attn_diag [B,N] = diag(A);
 *   out_idx int32 [B,k] receives the selected tokens; scratch >= sg_outlier_scratch_bytes. */
size_t sg_outlier_scratch_bytes(int B, int D, int k);
int sg_outlier_suppress(float* feats, const float* attn_cls, const float* attn_diag, int B, int gh, int gw, int D,
                        int top_k, float contamination_temp, int32_t* out_idx, void* scratch, sg_stream s);
/* sg_cross_tile_fusion replaces CrossTileFusion.forward applied to every tile of a scene in raster order
 * (cross_tile_fusion.py:290-320; 'weighted' :185-236 adaptive branch, 'attention' :143-183).  The reference never calls the
 * module (dead code); the semantics are those of running it tile by tile with B=1 (oracle/refine.py).
 *   tokens [hg*wg, gh*gw, C] f32 patch tokens of the scene's tiles, updated in place; mode 0 = weighted, 1 = attention. */
size_t sg_cross_tile_scratch_bytes(int T, int gh, int gw, int C, int bw);
int sg_cross_tile_fusion(float* tokens, int hg, int wg, int gh, int gw, int C, int bw, int mode, float strength, void* scratch, sg_stream s);
/* The same fusion for a rank holding tiles [tile0, tile0+n_local) of the raster list: neighbour strips are read
 * from PACKED buffers indexed by the global tile id, which the caller all-gathers between the steps
 *   pack(which=0: original right columns [gh*bw,C]) -> gather -> fuse(pass 0) -> pack(which=1: final bottom rows [bw*gw,C],
 *   columns [0,bw) taken from left_result) -> gather -> fuse(pass 1) -> apply.
 * tokens / out / result / left_result / top_result are local ([n_local, ...]); nbr_strips is global ([hg*wg, S, C]). */
int sg_cross_tile_pack(const float* tokens, const float* left_result, int n_local, int tile0, int wg, int gh, int gw, int C, int bw,
                       int which, float* out, sg_stream s);
int sg_cross_tile_fuse(const float* tokens, const float* nbr_strips, int n_local, int tile0, int wg, int gh, int gw, int C, int bw,
                       int mode, float strength, int pass, float* result, sg_stream s);
int sg_cross_tile_apply(float* tokens, const float* left_result, const float* top_result, int n_local, int tile0, int wg, int gh, int gw,
                        int C, int bw, sg_stream s);
/* SelfAttentionEnhancementModule feature mode (self_attention_enhancement.py:71-150,247-324) */
int sg_weak_token_replace(float* feats, const float* attn_diag, int B, int gh, int gw, int D, int top_k,
                          int32_t* out_idx, void* scratch, sg_stream s);
/* SimilarityEnhancementModule.compute_similarity_map (similarity_enhancement.py:37-66):
 *   patches [B,n,D] f32 (row stride ld) -> sim [B,n,n] f32 */
int sg_similarity_map(const float* patches, int64_t batch_stride, int ld, int B, int n, int D, float temperature,
                      int add_self, int precision, float* sim, void* scratch, size_t scratch_bytes, sg_stream s);

/* ---- building-block ops exported for unit parity tests -----------------------------------------
 * C[M,N] = act(A[M,K] . W[N,K]^T + bias) (+ residual); f32 in/out at the boundary, computed in
 * `precision`.  act: 0 none, 1 QuickGELU, 2 erf GELU. */
int sg_op_linear(const float* A, const float* W, const float* bias, const float* residual, float* C,
                 int M, int N, int K, int act, int precision, void* scratch, size_t scratch_bytes, sg_stream s);
/* The residual GEMM -> LayerNorm -> GEMM chain of a transformer block (reference open_clip/transformer.py:234-254: x = x + out_proj(attn);
 * mlp(ln_2(x)), and the next block's attention(ln_1(x))):  x <- x + A.W1^T + b1 (in place),  y = act(LayerNorm(x; gamma, beta).W2^T + b2).
 * fold = 0 runs the LayerNorm as its own pass; fold = 1 is what the towers do in the 2-byte modes: the first GEMM's epilogue also writes the
 * 2-byte copy of x and per-64-column (sum, centred sum of squares), the second GEMM runs on that copy with W' = gamma o W2 and applies
 * rstd (acc - mean c) + b' in its epilogue (c = row sums of W', b' = b2 + W2.beta) -- no pass over x in between.
 * precision: SG_PREC_BF16 / SG_PREC_F16 / SG_PREC_F16X2; fold needs M >= 1024, D >= 512, D % 64 == 0, N2 >= 512.  All pointers are f32 device memory. */
size_t sg_op_ln_chain_scratch_bytes(int M, int K1, int D, int N2);
int sg_op_ln_chain(const float* A, const float* W1, const float* b1, float* x, const float* gamma, const float* beta, const float* W2,
                   const float* b2, float* y, int M, int K1, int D, int N2, int act, int precision, int fold, void* scratch,
                   size_t scratch_bytes, sg_stream s);
/* bf16 GEMM on caller-packed operands: A [M,K], W [N,K] bf16 (K % 64 == 0), C bf16 or f32 */
int sg_gemm_bf16_raw(const void* A, const void* W, const float* bias, const float* residual, void* C, int M, int N, int K,
                     int act, int c_is_bf16, sg_stream s);
/* fp8 (OCP e4m3, v_mfma_f32_16x16x128_f8f6f4) GEMM on quantised operands: C = act((A8 . W8^T) * sa[m] * sw[n] + bias) (+ residual);
 * A8 [M,K], W8 [N,K] bytes, K % 128 == 0; sg_quantize_rows_fp8 produces an operand and its per-row scales (absmax / 448). */
int sg_gemm_fp8_raw(const void* A8, const float* sa, const void* W8, const float* sw, const float* bias, const float* residual, void* C,
                    int M, int N, int K, int act, int c_is_bf16, sg_stream s);
/* The MXFP8 forms of the same GEMM (OCP microscaling: one E8M0 power-of-two scale, value 2^(byte - 127), per 32 consecutive K elements,
 * fed to the scale operands of v_mfma_scale_f32_16x16x128_f8f6f4).  Block scales of an [M, K] operand are laid out [K/128][M][4] bytes
 * (the four blocks of one 128-wide K step of a row form one dword).  Exactly one of `sa` (per-row f32 scales) / `a_mx` (block scales)
 * describes A8.  With `c_mx` the result act(..) is written as e4m3 bytes [M,N] with block scales `c_mx_scale` [N/128][M][4] -- the next
 * linear's MX operand straight out of the epilogue (the tower's fc -> proj hand-off in SG_PREC_FP8) -- and C / residual are unused.
 * Needs M >= 1024, N >= 256, N % 8 == 0 (N % 128 == 0 with c_mx), K % 128 == 0. */
int sg_gemm_fp8_mx_raw(const void* A8, const float* sa, const void* a_mx, const void* W8, const float* sw, const float* bias,
                       const float* residual, void* C, void* c_mx, void* c_mx_scale, int M, int N, int K, int act, int c_is_bf16, sg_stream s);
int sg_quantize_rows_fp8(const float* x, int64_t rows, int D, void* y, float* scale, sg_stream s);
int sg_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int rows, int D, float eps, sg_stream s);
/* multi-term attention over packed qkv [B,N,3D] (rows q|k|v, nn.MultiheadAttention order);
 * variant = enum sg_model_type (SG_VANILLA = ordinary softmax(q k^T) v). bias: [B,n,n] or NULL.
 * Optional outputs: attn_cls/attn_diag [B,N] head-averaged probabilities (vanilla only). */
size_t sg_op_attention_scratch_bytes(int B, int N, int D, int H, int precision);
int sg_op_attention(const float* qkv, int B, int N, int D, int H, int variant, const float* sim, float sim_weight,
                    float* ctx, float* attn_cls, float* attn_diag, int precision, void* scratch, size_t scratch_bytes,
                    sg_stream s);

/* ---- SimFeatUp joint bilateral upsampler --------------------------------------------------------
 * sg_adaptive_conv mirrors featup.adaptive_conv_cuda AdaptiveConv.apply as called at
 * simfeatup_dev/upsamplers.py:274 (semantics: adaptive_conv_py_simple, :14-25):
 *   input [B,C,h+d-1,w+d-1], filters [B,h,w,d,d] -> out [B,C,h,w], all f32. */
int sg_adaptive_conv(const float* input, const float* filters, int B, int C, int h, int w, int d, float* out, sg_stream s);

/* ---- Cluster-Then-Debias (reference CTD.py as segmentor.py:339-365 drives it) --------------------
 * sg_ctd_debias replaces, for every tile of a launch at once,
 *     _, labels = cluster_patch_tokens_dbscan(feats, grid_hw, {'metric': 'euclidean', 'eps': eps, 'min_samples': m})   CTD.py:147-296
 *     feats     = adaptive_debiasing(items=feats, labels=labels, bias=cls, factor=factor)                              CTD.py:299-366
 * (scikit-learn DBSCAN on the CPU in the reference).  tokens [B,n,C] f32 in/out, cls [B,C] = the CLS features (unit, or raw with
 * normalize_cls = 1: segmentor.py:310 normalises them first), labels_out int32 [B,n] (-1 = noise) or NULL.  n > 8192 leaves the tokens unchanged, as the reference's max_points does. */
size_t sg_ctd_scratch_bytes(int B, int n, int C);
int sg_ctd_debias(float* tokens, const float* cls, int B, int n, int C, double eps, int min_samples, float factor, int normalize_cls,
                  int32_t* labels_out, void* scratch, size_t scratch_bytes, sg_stream s);

/* ---- CLIP text tower (init-time producer of query_features) -------------------------------------
 * sg_text_encode replaces CLIP.encode_text(tokens) (open_clip/model.py:288-306): token + positional embedding, causal
 * residual blocks, ln_final, EOT pooling (argmax of the ids), @ text_projection.  Tensor names = the text part of the CLIP
 * state dict ("token_embedding.weight", "positional_embedding", "transformer.resblocks.N.*", "ln_final.*", "text_projection").
 *   tokens int32 [n_seq, context_length] -> out [n_seq, E] f32 (not normalised).  Init-time call: unlike the rest of the ABI it
 *   synchronises the stream before returning, so that an id outside [0, vocab_size) is SG_ERR_INVALID (the reference's
 *   nn.Embedding raises) instead of a silently clamped row. */
typedef struct sg_text sg_text;
int  sg_text_create(sg_text** out, int device, int width, int layers, int heads, int context_length, int vocab_size, int embed_dim,
                    int quick_gelu, int precision);
void sg_text_destroy(sg_text* t);
int  sg_text_set_tensor(sg_text* t, const char* name, const float* dev_f32, int64_t numel, sg_stream s);
size_t sg_text_workspace_bytes(const sg_text* t, int n_seq);
int  sg_text_encode(sg_text* t, const int32_t* tokens, int n_seq, float* out, void* workspace, size_t workspace_bytes, sg_stream s);

/* JBU context.  sg_jbu_create replaces get_upsampler(name, dim) (upsamplers.py:353-369; kind 0 = 'jbu_one', 1 = 'jbu_stack');
 * sg_jbu_set_tensor takes the tensors by their state-dict names ("up.range_temp", "up2.fixup_proj.0.weight",
 * "fixup_proj.1.weight" ...), i.e. load_state_dict (segmentor.py:281-283);
 * sg_jbu_upsample replaces self.upsampler(image_features, img) (segmentor.py:371 -> upsamplers.py:278-325):
 *   source [B, gh*gw, C] patch tokens (pixel-major), guidance [B,3,GH,GW] normalised tile -> out [B, 16gh*16gw, C].
 *   precision: SG_PREC_F32 (parity kernels), SG_PREC_F16X2 (f32-grade: the linears and the low-res adaptive convolution on three f16 MFMAs per
 *   product -- what an exact tower mode is paired with), SG_PREC_BF16 (throughput: low-res convolution on bf16 operands, f16 fixup chain). */
typedef struct sg_jbu sg_jbu;
int  sg_jbu_create(sg_jbu** out, int device, int kind, int feat_dim);
void sg_jbu_destroy(sg_jbu* j);
int  sg_jbu_set_tensor(sg_jbu* j, const char* name, const float* dev_f32, int64_t numel, sg_stream s);
size_t sg_jbu_workspace_bytes(const sg_jbu* j, int B, int gh, int gw);
int  sg_jbu_upsample(sg_jbu* j, const float* source, const float* guidance, int B, int gh, int gw, int GH, int GW, int precision,
                     float* out, void* workspace, size_t workspace_bytes, sg_stream s);

/* sg_jbu_logits (throughput mode, bf16, C % 64 == 0) replaces segmentor.py:368-379 for a batch of tiles in one call:
 *   feats = upsampler(tokens -> [1,C,g,g], img); feats /= |feats|; logits = feats @ T^T (+ cls_token_lambda * cls_logits)
 * with the JBU tail fused: out = x + 0.1 * fixup_proj(x) is never written -- the C x C 1x1 conv (upsamplers.py:301,325) runs as a GEMM whose
 * epilogue only accumulates |out|^2 per pixel, and out . T^T = x . (T^T + 0.1 Wf^T T^T) + 0.1 bf . T^T is a Q-wide f32 product.
 *   source [B, gh*gw, C] (global debias already applied), guidance [B,3,GH,GW], text [Q,C] (Q <= 32), cls [B,C] or NULL
 *   -> logits [B, Q, 16gh*16gw] f32.  Workspace as sg_jbu_workspace_bytes. */
int  sg_jbu_logits(sg_jbu* j, const float* source, const float* guidance, int B, int gh, int gw, int GH, int GW, int precision,
                   const float* text, int Q, const float* cls, float cls_token_lambda, float* logits, void* workspace, size_t workspace_bytes,
                   sg_stream s);

/* the normalised, zero-padded tile planes [T,3,up_h,up_w] f32 the reference hands to the upsampler as `img`
 * (segmentor.py:424-431 crop + pad, :371) */
int sg_extract_tiles(const sg_tile_batch* tiles, int up_h, int up_w, float* out, sg_stream s);

/* similarity-weighted global debias as a stand-alone op (segmentor.py:322-336), used ahead of the upsampler:
 *   out = tokens - cls_hat * (cos(tokens, cls_hat) * factor), tokens [B,n,E], cls [B,E] */
int sg_global_debias(const float* tokens, const float* cls, int B, int n, int E, float factor, float* out, sg_stream s);

#ifdef __cplusplus
}
#endif
#endif /* SEGEARTH_HIP_H */
