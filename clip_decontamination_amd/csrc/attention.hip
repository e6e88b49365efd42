// Fused multi-term attention for the ViT blocks (throughput mode, bf16 MFMA, f32 softmax).
// Replaces nn.MultiheadAttention's bmm/softmax/bmm (reference open_clip/transformer.py:204,230-232),
// the self-self variants of custom_attn (:858-908) and GEM's SelfSelfAttention (gem/gem_utils.py:60-123)
// without ever materialising an N x N score matrix in HBM.
//
// Layout (gfx950, wave64): a workgroup = 4 waves = 128 queries of one (image, head); each wave owns
// 32 queries.  Scores are computed TRANSPOSED with v_mfma_f32_32x32x16_bf16 (A port = K rows from
// LDS, B port = Q rows held in registers), so the query sits on the lane and the 32 keys of a
// sub-block sit in the 16 accumulator registers of the two half-waves: the row max / row sum need
// 15 local ops + ONE cross-half shuffle.  The probabilities, converted to bf16 in place, are exactly
// the B operand of the next MFMA (accumulator-as-operand, k order (j&3)+8(j>>2)+4h), which computes
// O^T = V^T.P^T with V^T fragments fetched by ds_read_b64_tr_b16 (hardware transpose) from a
// row-major V tile.  O^T keeps the query on the lane, so the online-softmax rescale is lane-local.
// This file is compiled TWICE: as is (bf16 operands, the kernels + attention_bf16 + the statistics kernels) and through
// attention_f16.hip with SG_ATTN_F16 = 1 (IEEE f16 operands on the v_mfma_*_f16 forms: same cycles, 3 more mantissa bits --
// the reference's own GPU arithmetic, SG_PREC_F16), which adds attention_f16_impl.  The operand kind is a template parameter of
// the kernels, so the two translation units instantiate different symbols.
#include "rowops.h"

// A THIRD compilation (attention_h2.hip, SG_ATTN_H2 = 1) is the two-plane f16 form of SG_PREC_F16X2 (common.h h2_t): every operand is
// hi + lo, every product three f16 MFMAs (K_hi.Q_hi + K_lo.Q_hi + K_hi.Q_lo; V_hi.P_hi + V_lo.P_hi + V_hi.P_lo) into the same f32
// accumulators, the probabilities are split into two planes in registers -- f32-grade scores and contexts at a third of the MFMA rate.
#ifndef SG_ATTN_F16
#define SG_ATTN_F16 0
#endif
#ifndef SG_ATTN_H2
#define SG_ATTN_H2 0
#endif

namespace sg {

constexpr bool AH2 = SG_ATTN_H2 != 0;
constexpr bool AF16 = SG_ATTN_F16 != 0 || AH2;
int attention_f16_impl(const AttnArgs& a, hipStream_t s);      // defined by the SG_ATTN_F16 translation unit
int attention_h2_impl(const AttnArgs& a, hipStream_t s);       // defined by the SG_ATTN_H2 translation unit

constexpr int QB = 128;        // queries per workgroup
constexpr int KT = 64;         // keys per LDS tile
constexpr float RESCALE_TAU = 8.0f;   // log2 units: probabilities may reach 2^8 before the running maximum is raised

typedef __attribute__((ext_vector_type(4))) short short4_;
typedef __attribute__((address_space(3))) short4_* lds_s4_ptr;

// Two-plane form (AH2): a head row is PW = 2 DH f16 in HBM ([8 hi | 8 lo] groups; the host passes strides in f16 units).  In LDS the planes
// are DE-INTERLEAVED (store_rows): a K row is [DH hi | DH lo] (+8 pad: the 16-row ds_read_b128 stays conflict-free, row stride 17 / 21 / ...
// 16-byte slots), a V row [DVT*32 hi | DVT*32 lo] (+32 pad: 80 / 112 / ... dwords = 16 mod 64, so the four rows of a transposed read
// cover four disjoint 16-dword bank groups) -- the fragment reads of one plane are exactly the plain kernel's.
template <int DH> struct AttnCfg {
  static constexpr int KS = DH / 16;                 // k-steps of the score MFMA
  static constexpr int DVT = (DH + 31) / 32;         // 32-row tiles of O^T
  static constexpr int PW = AH2 ? 2 * DH : DH;       // 2-byte elements per head row in HBM
  static constexpr int K_LO = DH;                    // AH2: offset of the lo plane inside an LDS row
  static constexpr int V_LO = DVT * 32;
  static constexpr int K_LD = PW + 8;                // LDS row strides (elements): conflict-free ds_read_b128 of 16 rows
  static constexpr int V_LD = (AH2 ? 2 : 1) * DVT * 32 + 32;   //   and 4-row ds_read_b64_tr_b16 blocks (48-dword stride at dh 64)
  static constexpr int CH = PW / 8;                  // 16-byte chunks per row
  static constexpr int NL = (KT * CH + 255) / 256;   // chunks per thread per tile
};

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// HBM -> registers (issued early, T14 async-stage split) and registers -> LDS (after the compute phase).
// No multiplies and no branches per tile: a chunk's element offset is min(tile base + its offset inside the tile, offset of the same
// column in the last row) -- rows past the end read row N-1 (masked in the scores).  RowMap holds the per-thread constants.
template <int DH> struct RowMap { int off[AttnCfg<DH>::NL], omax[AttnCfg<DH>::NL]; };
template <int DH>
__device__ __forceinline__ RowMap<DH> make_row_map(int64_t st, int N, int tid) {
  constexpr int CH = AttnCfg<DH>::CH;
  RowMap<DH> m;
#pragma unroll
  for (int i = 0; i < AttnCfg<DH>::NL; ++i) {
    int idx = tid + i * 256;
    idx = idx < KT * CH ? idx : KT * CH - 1;           // ragged chunk counts (dh 80): load a valid duplicate, never stored
    const int r = idx / CH, c = idx % CH;
    m.off[i] = r * (int)st + c * 8;
    m.omax[i] = (N - 1) * (int)st + c * 8;
  }
  return m;
}
template <int DH>
__device__ __forceinline__ void load_rows(const bf16_t* __restrict__ src, int64_t st, int row0, const RowMap<DH>& m,
                                          u32x4 (&reg)[AttnCfg<DH>::NL]) {
  const int base = row0 * (int)st;
#pragma unroll
  for (int i = 0; i < AttnCfg<DH>::NL; ++i) {
    const int o = base + m.off[i];
    reg[i] = *reinterpret_cast<const u32x4*>(src + (o < m.omax[i] ? o : m.omax[i]));
  }
}
// lo_off (AH2): HBM chunk 2 g + p of a row (plane p of storage group g) goes to LDS column p * lo_off + 8 g
template <int DH>
__device__ __forceinline__ void store_rows(bf16_t* lds, int ld, int tid, const u32x4 (&reg)[AttnCfg<DH>::NL], int lo_off) {
  constexpr int CH = AttnCfg<DH>::CH;
#pragma unroll
  for (int i = 0; i < AttnCfg<DH>::NL; ++i) {
    const int idx = tid + i * 256;
    const int cc = idx % CH;
    const int col = AH2 ? (cc & 1) * lo_off + (cc >> 1) * 8 : cc * 8;
    if (idx < KT * CH) *reinterpret_cast<u32x4*>(lds + (idx / CH) * ld + col) = reg[i];
  }
}

// GENERIC = additive bias and / or the 'Experimental' re-softmax (last block only); the 23 ordinary blocks run the lean path.
// MULTI = several separately soft-maxed streams are summed (SCLIP / SegEarth / GEM); otherwise no second accumulator.
// Lean variants are capped at 256 registers (VGPR-form MFMA, 2+ waves per SIMD); the register-hungry ones (bias + multi-stream,
// head_dim > 64) may take the whole file rather than spill.
// GK: 0 = lean (nothing added to the scores), 1 = generic with every option at run time (bias, Gaussian factors, re-softmax, causal mask),
// 2 = the 'Experimental' last block at compile time: similarity-map bias + re-softmax, no mask, no Gaussian factors.
// H2: nothing but a name -- the two-plane translation unit instantiates the SAME <DH, TS, GK, MULTI, F16 = true, PV> combinations as the f16 one
// with different bodies (AH2 is a file-level constant), so it must not share their symbols.
template <int DH, int TS, int GK, bool MULTI, bool F16, bool PV, bool H2 = AH2>
// (two-plane form: twice the fragment registers -- the variants with a bias / two summed terms and every head_dim > 64 get the whole file too)
__global__ __launch_bounds__(256, ((GK != 0 && MULTI) || (DH > 64 && (GK != 0 || MULTI || DH > 80)) || (AH2 && (DH > 64 || (GK != 0 && TS == 2)))) ? 1 : 2) void attn_kernel(AttnArgs a) {
  constexpr bool GENERIC = GK != 0, EXPER = GK == 2;
  using C = AttnCfg<DH>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BUF = TS * KT * C::K_LD + KT * C::V_LD;                 // elements per LDS buffer: [TS K tiles][V tile]
  constexpr int NB = (size_t)2 * BUF * sizeof(bf16_t) > 160 * 1024 ? 1 : 2;   // two buffers unless they exceed the CU's LDS (two-plane form, two
                                                                        // summed terms at head_dim 128): then ONE buffer and a second barrier per tile
  bf16_t* sbuf = reinterpret_cast<bf16_t*>(smem);                       // [NB][BUF]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  // XCD-aware order: workgroup w runs on XCD w % 8 (dispatch round-robin; used for speed only).  All query blocks of one
  // (image, head) pair are queued back to back on ONE XCD, so its K / V (350 kB at N=1370) are fetched into that XCD's L2 once
  // instead of once per query block through the fabric (rocprofv3 FETCH_SIZE: 6.1 GB -> see profiles/).
  // With an additive bias (similarity map, shared by the H heads) the order is turned round: the H heads of one (image, query block)
  // run back to back on one XCD, so the 128 x N bias slice (700 kB) is fetched once instead of once per head, which outweighs the
  // K / V re-reads (bias [B,n,n] f32 is 16x the size of K and V together).
  // Round 3: with enough images per launch (>= 16: two per XCD) an XCD takes WHOLE images and walks each as head groups of 4: for a group,
  // all query blocks, the 4 heads of a query block back to back.  The K / V of a head group (2 MB at ViT-L/14) then stay in the XCD's L2 for
  // all 11 query blocks and a bias slice is shared by 4 heads that start together, instead of every query block re-streaming the K / V of all
  // 16 heads (8.4 MB per image, more than the L2: rocprofv3 FETCH_SIZE 12.1 GB per 128-tile launch against ~2 GB of unique data).
  const int nq = (a.N + QB - 1) / QB;
  const int xw = blockIdx.x & 7, jw = blockIdx.x >> 3;
  int b, hd, qb;
  if (GENERIC && (EXPER || a.bias != nullptr) && a.B >= 16 && (a.H & 3) == 0) {
    const int hh = jw & 3, t1 = jw >> 2;
    qb = t1 % nq;
    const int t2 = t1 / nq, ngrp = a.H >> 2;
    hd = (t2 % ngrp) * 4 + hh;
    b = (t2 / ngrp) * 8 + xw;
    if (b >= a.B) return;
  } else if (GENERIC && (EXPER || a.bias != nullptr)) {
    const int unit = (jw / a.H) * 8 + xw;
    if (unit >= a.B * nq) return;
    b = unit / nq; qb = unit % nq; hd = jw % a.H;
  } else {
    const int grp = (jw / nq) * 8 + xw;
    if (grp >= a.H * a.B) return;
    b = grp / a.H; hd = grp % a.H; qb = jw % nq;
  }
  const int q_glob = qb * QB + wave * 32 + c;
  const int q_ld = q_glob < a.N ? q_glob : a.N - 1;
  const int n = a.N - 1;
  const float scale = a.scale_per_image ? a.scale_per_image[b] : a.scale;      // > 0
  const float c2 = scale * LOG2E;                                              // scores live in the exp2 domain
  constexpr bool do_pv = PV;                                                   // a context is wanted (false: the log-sum-exp pass alone) -- a template
                                                                               // parameter, so that the key-tile loop is free of run-time branches
  const int n_streams = MULTI ? a.n_terms : 1;
  const int64_t head_off = (int64_t)b * a.sb + (int64_t)hd * C::PW;
  const bf16_t* vbase = a.v + (int64_t)b * a.v_sb + (int64_t)hd * C::PW;
  const float lse1_2 = (GENERIC && (EXPER || a.resoftmax)) ? a.lse_in[((int64_t)b * a.H + hd) * a.N + q_ld] * LOG2E : 0.f;
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;               // transposed-read lane roles

  const RowMap<DH> kmap = make_row_map<DH>(a.st, a.N, tid), vmap = make_row_map<DH>(a.v_st, a.N, tid);

  f32x16 o_tot[MULTI ? C::DVT : 1];
  if (MULTI) {
#pragma unroll
    for (int t = 0; t < C::DVT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_tot[t][r] = 0.f;
  }
  f32x16 o_acc[C::DVT];
  float m_run = -INFINITY, l_run = 0.f;                     // m_run: the (stale) maximum the exponentials are taken against

  for (int sidx = 0; sidx < n_streams; ++sidx) {
    const bf16_t* kptr[TS];
    bf16x8 qf[TS][C::KS];                                    // B port: lane = query, 8 consecutive k per half-wave
    bf16x8 qfl[AH2 ? TS : 1][AH2 ? C::KS : 1];               // AH2: the lo plane of the same 8 k values
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      kptr[t] = a.k[a.sum_scores ? t : sidx] + head_off;
      const bf16_t* qp = a.q[a.sum_scores ? t : sidx] + head_off + (int64_t)q_ld * a.st;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        if constexpr (AH2) {                                 // storage group 2 ks + h: 32 bytes = [8 hi | 8 lo]
          qf[t][ks] = *reinterpret_cast<const bf16x8*>(qp + (2 * ks + h) * 16);
          qfl[t][ks] = *reinterpret_cast<const bf16x8*>(qp + (2 * ks + h) * 16 + 8);
        } else qf[t][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16 + h * 8);
      }
    }
    if (!GENERIC) {
      // Lean path: Q is pre-multiplied by scale * log2(e) (rounded back to bf16 once per stream) and the score accumulator starts at
      // -m_run, so the MFMA itself delivers `s * c2 - m` -- the 32 v_fma_f32 per key tile of the softmax disappear.
#pragma unroll
      for (int t = 0; t < TS; ++t)
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          if constexpr (AH2) {                               // (hi + lo) * c2 in f32, split again
            const f16x8 hq = __builtin_bit_cast(f16x8, qf[t][ks]), lq = __builtin_bit_cast(f16x8, qfl[t][ks]);
            float fq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) fq[j] = ((float)hq[j] + (float)lq[j]) * c2;
            uint4 nh, nl;
            split_h2x8(fq, nh, nl);
            qf[t][ks] = __builtin_bit_cast(bf16x8, nh); qfl[t][ks] = __builtin_bit_cast(bf16x8, nl);
          } else if constexpr (F16) {
            const f16x8 hq = __builtin_bit_cast(f16x8, qf[t][ks]);
            f32x8_t fq;
#pragma unroll
            for (int j = 0; j < 8; ++j) fq[j] = (float)hq[j] * c2;
            qf[t][ks] = __builtin_bit_cast(bf16x8, __builtin_convertvector(fq, f16x8));
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[t][ks][j] = (__bf16)((float)qf[t][ks][j] * c2);
          }
        }
    }
    // Retire the Q loads HERE: otherwise the compiler's counted vmcnt in front of the first QK^T MFMA also waits (every iteration)
    // for the K/V prefetch that was issued a few instructions earlier, exposing its full latency.
#pragma unroll
    for (int t = 0; t < TS; ++t)
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) { asm volatile("" : "+v"(qf[t][ks])); if constexpr (AH2) asm volatile("" : "+v"(qfl[t][ks])); }
#pragma unroll
    for (int t = 0; t < C::DVT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[t][r] = 0.f;
    m_run = -INFINITY; l_run = 0.f;

    u32x4 kreg[TS][C::NL], vreg[C::NL];
#pragma unroll
    for (int t = 0; t < TS; ++t) load_rows<DH>(kptr[t], a.st, 0, kmap, kreg[t]);
    if (do_pv) load_rows<DH>(vbase, a.v_st, 0, vmap, vreg);
#pragma unroll
    for (int t = 0; t < TS; ++t) store_rows<DH>(sbuf + t * KT * C::K_LD, C::K_LD, tid, kreg[t], C::K_LO);
    if (do_pv) store_rows<DH>(sbuf + TS * KT * C::K_LD, C::V_LD, tid, vreg, C::V_LO);
    __syncthreads();
    int cur = 0;

    // GENERIC: the additive bias (similarity map) of the NEXT key tile is fetched one tile ahead -- 32 coalesced dword loads per
    // lane (the map is symmetric, so it is read as bias[key][query] with the queries on the lanes)
    // Register-hungry variants (bias + several soft-maxed streams at head_dim > 64: ViT-H with SegEarth / SCLIP and a similarity map) fetch
    // the bias of a tile when it is used instead of one tile ahead: 32 fewer live registers, no scratch (they spilled 124-590 B per lane).
    constexpr bool BIAS_AHEAD = !(GENERIC && MULTI && DH > 64);
    float bnext[GENERIC ? 32 : 1];
    const float bias_rn = (GENERIC && a.bias_rn) ? a.bias_rn[((int64_t)b * a.H + hd) * a.N + q_ld] : 1.f;
    auto fetch_bias = [&](int kbase) {
      if (GENERIC && (EXPER || a.bias)) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const int key = kbase + 32 * (i >> 4) + (i & 3) + 8 * ((i & 15) >> 2) + 4 * h;
          float bvv = (key >= 1 && key < a.N && q_ld >= 1) ? a.bias[(int64_t)b * a.bias_bstride + (int64_t)(key - 1) * n + (q_ld - 1)] : 0.f;
          if (!EXPER && a.bias_cn) bvv *= a.bias_cn[((int64_t)b * a.H + hd) * a.N + (key < a.N ? key : a.N - 1)] * bias_rn;   // Gaussian variants: |q_i| |k_j|
          bnext[i] = bvv;
        }
      }
    };
    if (BIAS_AHEAD) fetch_bias(0);
    for (int k0 = 0; k0 < a.N; k0 += KT) {
      const bool has_next = k0 + KT < a.N;
      if (has_next) {                                        // next tile's loads fly while this tile computes
#pragma unroll
        for (int t = 0; t < TS; ++t) load_rows<DH>(kptr[t], a.st, k0 + KT, kmap, kreg[t]);
        if (do_pv) load_rows<DH>(vbase, a.v_st, k0 + KT, vmap, vreg);
      }
      const bf16_t* sK = sbuf + (NB == 2 ? cur * BUF : 0);
      const bf16_t* sV = sK + TS * KT * C::K_LD;

      // ---- scores of the whole 64-key tile: S^T[key][query], 2 sub-blocks x KS k-steps -----------------------------
      f32x16 sacc[2];
      {
        const float s_init = (!GENERIC && m_run != -INFINITY) ? -m_run : 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[sub][r] = s_init;
        if constexpr (AH2) {
          // two-plane form: the fragments of one k-step at a time (both planes, both key sub-blocks = 16 registers), three MFMAs each
#pragma unroll
          for (int t = 0; t < TS; ++t)
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
              bf16x8 kh[2], kl[2];
#pragma unroll
              for (int sub = 0; sub < 2; ++sub) {
                const bf16_t* kp = sK + t * KT * C::K_LD + (sub * 32 + c) * C::K_LD + ks * 16 + h * 8;
                kh[sub] = *reinterpret_cast<const bf16x8*>(kp);
                kl[sub] = *reinterpret_cast<const bf16x8*>(kp + C::K_LO);
              }
#pragma unroll
              for (int sub = 0; sub < 2; ++sub) sacc[sub] = mfma_32x32x16<true>(kh[sub], qf[t][ks], sacc[sub]);
#pragma unroll
              for (int sub = 0; sub < 2; ++sub) sacc[sub] = mfma_32x32x16<true>(kl[sub], qf[t][ks], sacc[sub]);
#pragma unroll
              for (int sub = 0; sub < 2; ++sub) sacc[sub] = mfma_32x32x16<true>(kh[sub], qfl[t][ks], sacc[sub]);
            }
        } else {
        bf16x8 kf[2][TS][C::KS];                           // all K fragments of the tile first: one exposed LDS latency, not eight
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int t = 0; t < TS; ++t)
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks)
              kf[sub][t][ks] = *reinterpret_cast<const bf16x8*>(sK + t * KT * C::K_LD + (sub * 32 + c) * C::K_LD + ks * 16 + h * 8);
#pragma unroll
        for (int t = 0; t < TS; ++t)
#pragma unroll
          for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)              // the two key sub-blocks alternate: no MFMA waits on its predecessor
              sacc[sub] = mfma_32x32x16<F16>(kf[sub][t][ks], qf[t][ks], sacc[sub]);
        }
      }
      // (b) V^T fragments of the tile are fetched NOW (transposed LDS reads, lane 4q+p of a 16-lane group addresses row q,
      // cols 4p..4p+3) so their latency hides under the softmax arithmetic below.
      typedef __attribute__((ext_vector_type(8))) short short8_;
      short8_ vfr[AH2 ? 1 : C::DVT][AH2 ? 1 : 4];
      auto read_vt = [&](int t, int f, int plane) -> short8_ {   // V^T fragment (d-tile t, key step f) of one plane
        const int key0 = f * 16 + 4 * h;
        const bf16_t* p0 = sV + (key0 + qq) * C::V_LD + plane * C::V_LO + t * 32 + 16 * (g & 1) + 4 * pp;
        const short4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p0));
        const short4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p0 + 8 * C::V_LD));
        return (short8_){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      };
      if (do_pv && !AH2) {
#pragma unroll
        for (int t = 0; t < C::DVT; ++t)
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            const int key0 = f * 16 + 4 * h;
            const bf16_t* p0 = sV + (key0 + qq) * C::V_LD + t * 32 + 16 * (g & 1) + 4 * pp;
            const short4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p0));
            const short4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p0 + 8 * C::V_LD));
            vfr[t][f] = (short8_){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
      }
      // log2-domain scores for query q_glob (lane); key of (sub, r) = k0 + 32 sub + (r&3) + 8 (r>>2) + 4 h
      float sc[32];
      float mloc;
      if (GENERIC) {
        if (!BIAS_AHEAD) fetch_bias(k0);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const int key = k0 + 32 * (i >> 4) + (i & 3) + 8 * ((i & 15) >> 2) + 4 * h;
          const float bv = (EXPER || a.bias) ? bnext[i] * a.bias_w : 0.f;
          float v = sacc[i >> 4][i & 15] * c2;
          if (EXPER || a.resoftmax) v = (__builtin_amdgcn_exp2f(v - lse1_2) + bv) * LOG2E; else v += bv * LOG2E;
          sc[i] = (key < a.N && !(!EXPER && a.causal && key > q_glob)) ? v : -INFINITY;   // causal: text tower (build_causal_mask)
        }
        if (BIAS_AHEAD && has_next) fetch_bias(k0 + KT);   // lands under this tile's softmax / PV and the next tile's QK^T
        mloc = sc[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) mloc = fmaxf(mloc, sc[i]);
      } else {
        if (k0 + KT > a.N) {                               // tail tile: mask keys past the end (block-uniform branch)
#pragma unroll
          for (int i = 0; i < 32; ++i)
            if (k0 + 32 * (i >> 4) + (i & 3) + 8 * ((i & 15) >> 2) + 4 * h >= a.N) sacc[i >> 4][i & 15] = -INFINITY;
        }
        mloc = sacc[0][0];
#pragma unroll
        for (int i = 1; i < 32; ++i) mloc = fmaxf(mloc, sacc[i >> 4][i & 15]);   // already relative to the running maximum
      }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      // Lazy online softmax: the exponentials use a STALE maximum m_run that is only raised (and the accumulators rescaled) when some
      // row's maximum has outgrown it by more than 2^RESCALE_TAU -- probabilities stay <= 2^TAU, exact in the final O / l ratio.
      // The branch is wave-uniform and, after the first tiles, almost never taken (the 32 accumulator multiplies per tile go away).
      float lsum = 0.f;
      if (GENERIC) {
        if (__any(mloc > m_run + RESCALE_TAU)) {
          const float m_new = fmaxf(m_run, mloc);
          const float alpha = m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
          for (int t = 0; t < C::DVT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[t][r] *= alpha;
          l_run *= alpha;
          m_run = m_new;
        }
        const float m_sub = m_run == -INFINITY ? 0.f : m_run;   // a row with nothing unmasked yet: exp2(-inf - 0) = 0, never inf - inf
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          sc[i] = __builtin_amdgcn_exp2f(sc[i] - m_sub);
          lsum += sc[i];
        }
      } else {
        // mloc is the tile maximum RELATIVE to m_run (absolute while m_run is still -inf, where the accumulator started at 0)
        if (__any(mloc > RESCALE_TAU || m_run == -INFINITY)) {
          const bool first = m_run == -INFINITY;
          const float delta = first ? mloc : fmaxf(mloc, 0.f);               // how far the running maximum moves up
          const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
          if (delta != -INFINITY) {
#pragma unroll
            for (int i = 0; i < 32; ++i) sacc[i >> 4][i & 15] -= delta;
            m_run = first ? delta : m_run + delta;
          }
#pragma unroll
          for (int t = 0; t < C::DVT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[t][r] *= alpha;
          l_run *= alpha;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          sc[i] = __builtin_amdgcn_exp2f(sacc[i >> 4][i & 15]);
          lsum += sc[i];
        }
      }
      lsum += __shfl_xor(lsum, 32, 64);
      l_run += lsum;
      // P (bf16) is already the B operand: element j of k-step s2 of sub-block sub = sc[16 sub + 8 s2 + j]
      bf16x8 pf[4];                                          // (bit pattern: f16 values when F16; probabilities are <= 2^RESCALE_TAU)
      bf16x8 pfl[AH2 ? 4 : 1];                               // AH2: the lo plane of the probabilities
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        if constexpr (AH2) {
          uint4 ph, pl;                                      // probabilities are <= 2^RESCALE_TAU: no saturation needed
          split_h2_bounded(sc[8 * f], sc[8 * f + 1], ph.x, pl.x); split_h2_bounded(sc[8 * f + 2], sc[8 * f + 3], ph.y, pl.y);
          split_h2_bounded(sc[8 * f + 4], sc[8 * f + 5], ph.z, pl.z); split_h2_bounded(sc[8 * f + 6], sc[8 * f + 7], ph.w, pl.w);
          pf[f] = __builtin_bit_cast(bf16x8, ph); pfl[f] = __builtin_bit_cast(bf16x8, pl);
        } else if constexpr (F16) {
          f32x8_t fp;                                        // 4 x v_cvt_pk_f16_f32
#pragma unroll
          for (int j = 0; j < 8; ++j) fp[j] = sc[8 * f + j];
          pf[f] = __builtin_bit_cast(bf16x8, __builtin_convertvector(fp, f16x8));
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[f][j] = (__bf16)sc[8 * f + j];
        }
      }
      if constexpr (AH2) {
        if (do_pv) {
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            short8_ vh[C::DVT], vl[C::DVT];
#pragma unroll
            for (int t = 0; t < C::DVT; ++t) { vh[t] = read_vt(t, f, 0); vl[t] = read_vt(t, f, 1); }
#pragma unroll
            for (int t = 0; t < C::DVT; ++t) o_acc[t] = mfma_32x32x16<true>(*reinterpret_cast<bf16x8*>(&vh[t]), pf[f], o_acc[t]);
#pragma unroll
            for (int t = 0; t < C::DVT; ++t) o_acc[t] = mfma_32x32x16<true>(*reinterpret_cast<bf16x8*>(&vl[t]), pf[f], o_acc[t]);
#pragma unroll
            for (int t = 0; t < C::DVT; ++t) o_acc[t] = mfma_32x32x16<true>(*reinterpret_cast<bf16x8*>(&vh[t]), pfl[f], o_acc[t]);
          }
        }
      } else if (do_pv) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int t = 0; t < C::DVT; ++t)                 // alternate the d-blocks of O^T for the same reason
            o_acc[t] = mfma_32x32x16<F16>(*reinterpret_cast<bf16x8*>(&vfr[t][f]), pf[f], o_acc[t]);
      }
      if (NB == 1) __syncthreads();                          // single buffer: every wave is done reading the tile before it is overwritten
      if (has_next) {                                        // the other buffer was last read one iteration ago
        bf16_t* nK = sbuf + (NB == 2 ? (cur ^ 1) * BUF : 0);
#pragma unroll
        for (int t = 0; t < TS; ++t) store_rows<DH>(nK + t * KT * C::K_LD, C::K_LD, tid, kreg[t], C::K_LO);
        if (do_pv) store_rows<DH>(nK + TS * KT * C::K_LD, C::V_LD, tid, vreg, C::V_LO);
      }
      __syncthreads();
      cur ^= 1;
    }
    if (do_pv) {
      const float inv = 1.0f / l_run;
#pragma unroll
      for (int t = 0; t < C::DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (MULTI) o_tot[t][r] += o_acc[t][r] * inv; else o_acc[t][r] *= inv;
        }
    }
  }

  if (a.lse_out && h == 0 && q_glob < a.N) a.lse_out[((int64_t)b * a.H + hd) * a.N + q_glob] = (m_run + __builtin_amdgcn_logf(l_run)) * LN2;
  if (do_pv && q_glob < a.N) {
    bf16_t* op = a.ctx + (int64_t)b * a.ctx_sb + (int64_t)q_glob * a.ctx_st + (int64_t)hd * C::PW;
#pragma unroll
    for (int t = 0; t < C::DVT; ++t)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int dv = t * 32 + 8 * g4 + 4 * h;
        if (dv < DH) {
          uint2 o;
          const f32x16& of = MULTI ? o_tot[t] : o_acc[t];
          if constexpr (AH2) {                               // elements dv .. dv+3 of storage group dv / 8: hi at [4 h, 4 h + 4), lo 8 elements on
            uint2 ol;
            split_h2(of[4 * g4 + 0] * a.out_scale, of[4 * g4 + 1] * a.out_scale, o.x, ol.x);
            split_h2(of[4 * g4 + 2] * a.out_scale, of[4 * g4 + 3] * a.out_scale, o.y, ol.y);
            bf16_t* gp = op + (dv >> 3) * 16 + (dv & 7);
            *reinterpret_cast<uint2*>(gp) = o;
            *reinterpret_cast<uint2*>(gp + 8) = ol;
            continue;
          }
          o.x = pack_half2<F16>(of[4 * g4 + 0] * a.out_scale, of[4 * g4 + 1] * a.out_scale);
          o.y = pack_half2<F16>(of[4 * g4 + 2] * a.out_scale, of[4 * g4 + 3] * a.out_scale);
          *reinterpret_cast<uint2*>(op + dv) = o;
        }
      }
  }
}

template <int DH, int TS>
static int launch_attn(const AttnArgs& a, hipStream_t s) {
  using C = AttnCfg<DH>;
  const size_t lds2 = (size_t)2 * (TS * KT * C::K_LD + KT * C::V_LD) * sizeof(bf16_t);
  const size_t lds = lds2 > 160 * 1024 ? lds2 / 2 : lds2;                 // the kernel's NB
  const bool generic = a.bias != nullptr || a.resoftmax != 0 || a.causal != 0;
  const bool multi = !a.sum_scores && a.n_terms > 1;
  // TS == 2 means two SUMMED score terms = one stream: the multi-stream variants exist for TS == 1 only
  constexpr bool CAN_MULTI = TS == 1;
  SG_REQUIRE(CAN_MULTI || !multi, "attention: summed terms and separate streams are exclusive");
  using Kern = void (*)(AttnArgs);
  const bool exper = generic && a.bias && a.resoftmax && !a.causal && !a.bias_cn && !a.bias_rn && !multi && a.ctx;
  const Kern kern = exper ? attn_kernel<DH, TS, 2, false, AF16, true>
                  : a.ctx ? (generic ? (multi ? attn_kernel<DH, TS, 1, CAN_MULTI, AF16, true> : attn_kernel<DH, TS, 1, false, AF16, true>)
                                     : (multi ? attn_kernel<DH, TS, 0, CAN_MULTI, AF16, true> : attn_kernel<DH, TS, 0, false, AF16, true>))
                          : (generic ? attn_kernel<DH, TS, 1, false, AF16, false> : attn_kernel<DH, TS, 0, false, AF16, false>);
  SG_REQUIRE(a.ctx || !multi, "attention: a log-sum-exp-only pass has one stream");
  if (lds > 64 * 1024) SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int64_t nqb = cdiv(a.N, QB);
  const int64_t nblk = (generic && a.bias) ? ((a.B >= 16 && (a.H & 3) == 0) ? cdiv((int64_t)a.B, 8) * 8 * a.H * nqb     // whole images per XCD (see the kernel)
                                                                              : cdiv((int64_t)a.B * nqb, 8) * 8 * a.H)
                                           : cdiv((int64_t)a.H * a.B, 8) * 8 * nqb;
  SG_REQUIRE(nblk < (1ll << 31), "attention: grid too large");
  dim3 grid((unsigned)nblk);
  // algorithmic FLOPs: 2*N*N*dh per (term score) + 2*N*N*dh per stream PV, per (image, head)
  const int streams = a.sum_scores ? 1 : a.n_terms;
  const double fl = (double)a.B * a.H * 2.0 * a.N * (double)a.N * DH * (a.n_terms + (a.ctx ? streams : 0));
  prof_begin(PROF_ATTENTION, fl, s);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  prof_end(PROF_ATTENTION, s);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

#if SG_ATTN_H2
int attention_h2_impl(const AttnArgs& a, hipStream_t s) {   // strides arrive in f16 units (the caller doubled the element strides)
#elif SG_ATTN_F16
int attention_f16_impl(const AttnArgs& a, hipStream_t s) {
#else
int attention_bf16(const AttnArgs& a, hipStream_t s) {
  if (a.h2) return attention_h2_impl(a, s);
  if (a.f16) return attention_f16_impl(a, s);
#endif
  SG_REQUIRE(a.n_terms >= 1 && a.n_terms <= 3, "attention: n_terms=%d", a.n_terms);
  SG_REQUIRE(a.B > 0 && a.N > 0 && a.H > 0, "attention: empty problem");
  SG_REQUIRE(a.B < 65536 && a.H < 65536, "attention: grid too large");
  SG_REQUIRE(a.sb % 8 == 0 && a.st % 8 == 0 && a.v_sb % 8 == 0 && a.v_st % 8 == 0, "attention: strides must be multiples of 8 elements");
  SG_REQUIRE(!a.resoftmax || a.lse_in, "attention: resoftmax needs lse_in");
  SG_REQUIRE(a.ctx || a.lse_out, "attention: nothing to compute");
  SG_REQUIRE((int64_t)(a.N + 2 * KT) * a.st < (1ll << 31) && (int64_t)(a.N + 2 * KT) * a.v_st < (1ll << 31), "attention: token stride too large for 32-bit row offsets");
  const int ts = a.sum_scores ? a.n_terms : 1;
  SG_REQUIRE(ts <= 2, "attention: at most 2 summed terms");
#define SG_ATTN_CASE(DHV)                                                   \
  case DHV: return ts == 2 ? launch_attn<DHV, 2>(a, s) : launch_attn<DHV, 1>(a, s);
  switch (a.dh) {
    SG_ATTN_CASE(32)
    SG_ATTN_CASE(64)
    SG_ATTN_CASE(80)
    SG_ATTN_CASE(128)
    default: return fail(SG_ERR_INVALID, "attention: head_dim %d not built (32, 64, 80, 128)", a.dh);
  }
#undef SG_ATTN_CASE
}

#if !SG_ATTN_F16 && !SG_ATTN_H2
// ---- head-averaged statistics for outlier detection ------------------------------------------------------
// One wave per token j: for every head, s_cls = scale * q[0].k[j], s_diag = scale * q[j].k[j];
// probabilities are recovered from the per-row log-sum-exp the attention kernel wrote.
template <typename T>
__global__ __launch_bounds__(256) void attn_stats_kernel(const T* __restrict__ qkv, int64_t sb, int64_t st, const float* __restrict__ lse,
                                                         int N, int H, int dh, float scale, float* __restrict__ attn_cls,
                                                         float* __restrict__ attn_diag) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= N) return;
  const int D = H * dh;
  const T* q0 = qkv + (int64_t)b * sb;
  const T* qj = q0 + (int64_t)j * st;
  const T* kj = qj + D;
  float pc = 0.f, pd = 0.f;
  for (int hd = 0; hd < H; ++hd) {
    float dc = 0.f, dd = 0.f;
    for (int d = lane; d < dh; d += 64) {
      const float kv = ld_elem<T>(kj, hd * dh + d);
      dc += ld_elem<T>(q0, hd * dh + d) * kv;
      dd += ld_elem<T>(qj, hd * dh + d) * kv;
    }
    dc = wave_sum(dc); dd = wave_sum(dd);
    const float* l = lse + ((int64_t)b * H + hd) * N;
    pc += expf(dc * scale - l[0]);
    pd += expf(dd * scale - l[j]);
  }
  if (lane == 0) {
    attn_cls[(int64_t)b * N + j] = pc / (float)H;
    attn_diag[(int64_t)b * N + j] = pd / (float)H;
  }
}

// Fast form for bf16 and head_dim 32 / 64 / 128: a lane owns 8 consecutive channels (one 16-byte load) of q[0], q[j], k[j]; the LPH = dh/8
// lanes of a head reduce with log2(LPH) shuffles, one pass over the row handles 64/LPH heads at once (the generic kernel above does
// two 6-step wave reductions per head).
template <int LPH, bool F16>
__global__ __launch_bounds__(256) void attn_stats_fast_kernel(const bf16_t* __restrict__ qkv, int64_t sb, int64_t st, const float* __restrict__ lse,
                                                              int N, int H, float scale, float* __restrict__ attn_cls, float* __restrict__ attn_diag) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= N) return;
  const int D = H * LPH * 8;
  const bf16_t* q0 = qkv + (int64_t)b * sb;
  const bf16_t* qj = q0 + (int64_t)j * st;
  const bf16_t* kj = qj + D;
  float pc = 0.f, pd = 0.f;
  for (int c0 = 0; c0 < D / 8; c0 += 64) {
    const int ch = c0 + lane;                               // 8-channel chunk; head = ch / LPH
    float dc = 0.f, dd = 0.f;
    if (ch < D / 8) {
      const uint4 a = *reinterpret_cast<const uint4*>(q0 + ch * 8), bq = *reinterpret_cast<const uint4*>(qj + ch * 8),
                  kk = *reinterpret_cast<const uint4*>(kj + ch * 8);
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {bq.x, bq.y, bq.z, bq.w}, kw[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        auto lo = [](uint32_t w) { if constexpr (F16) { f16_t h{(uint16_t)(w & 0xffffu)}; return h2f(h); } else return __uint_as_float(w << 16); };
        auto hi = [](uint32_t w) { if constexpr (F16) { f16_t h{(uint16_t)(w >> 16)}; return h2f(h); } else return __uint_as_float(w & 0xffff0000u); };
        const float k0 = lo(kw[e]), k1 = hi(kw[e]);
        dc += lo(aw[e]) * k0 + hi(aw[e]) * k1;
        dd += lo(bw[e]) * k0 + hi(bw[e]) * k1;
      }
    }
#pragma unroll
    for (int o = LPH / 2; o > 0; o >>= 1) { dc += __shfl_xor(dc, o, 64); dd += __shfl_xor(dd, o, 64); }
    if (ch < D / 8 && (lane % LPH) == 0) {
      const int hd = ch / LPH;
      const float* l = lse + ((int64_t)b * H + hd) * N;
      pc += expf(dc * scale - l[0]);
      pd += expf(dd * scale - l[j]);
    }
  }
  pc = wave_sum(pc); pd = wave_sum(pd);
  if (lane == 0) {
    attn_cls[(int64_t)b * N + j] = pc / (float)H;
    attn_diag[(int64_t)b * N + j] = pd / (float)H;
  }
}

int attention_stats(const void* qkv, int is_bf16, int64_t sb, int64_t st, const float* lse, int B, int N, int H, int dh,
                    float scale, float* attn_cls, float* attn_diag, hipStream_t s) {
  dim3 grid((unsigned)cdiv(N, 4), (unsigned)B);
  const bool aligned = (sb % 8 == 0) && (st % 8 == 0) && ((((uintptr_t)qkv) & 15) == 0);
#define SG_STATS_FAST(LPH, F)                                                                                                      \
  hipLaunchKernelGGL((attn_stats_fast_kernel<LPH, F>), grid, dim3(256), 0, s, (const bf16_t*)qkv, sb, st, lse, N, H, scale, attn_cls, attn_diag)
  if (is_bf16 && is_bf16 != HK_F16X2 && aligned && (dh == 32 || dh == 64 || dh == 128)) {
    const bool h = is_bf16 == HK_F16;
    if (dh == 32) { if (h) SG_STATS_FAST(4, true); else SG_STATS_FAST(4, false); }
    else if (dh == 64) { if (h) SG_STATS_FAST(8, true); else SG_STATS_FAST(8, false); }
    else { if (h) SG_STATS_FAST(16, true); else SG_STATS_FAST(16, false); }
    SG_LAUNCH_CHECK();
    return SG_OK;
  }
#undef SG_STATS_FAST
  if (is_bf16 == HK_F16X2) hipLaunchKernelGGL(attn_stats_kernel<h2_t>, grid, dim3(256), 0, s, (const h2_t*)qkv, sb, st, lse, N, H, dh, scale, attn_cls, attn_diag);
  else if (is_bf16 == HK_F16) hipLaunchKernelGGL(attn_stats_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)qkv, sb, st, lse, N, H, dh, scale, attn_cls, attn_diag);
  else if (is_bf16) hipLaunchKernelGGL(attn_stats_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)qkv, sb, st, lse, N, H, dh, scale, attn_cls, attn_diag);
  else hipLaunchKernelGGL(attn_stats_kernel<float>, grid, dim3(256), 0, s, (const float*)qkv, sb, st, lse, N, H, dh, scale, attn_cls, attn_diag);
  SG_LAUNCH_CHECK();
  return SG_OK;
}
#endif  // !SG_ATTN_F16 && !SG_ATTN_H2

}  // namespace sg
