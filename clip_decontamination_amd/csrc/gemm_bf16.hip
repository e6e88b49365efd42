// bf16 MFMA GEMM for the ViT linears (QKV / out-proj / c_fc / c_proj / patch-embed / proj) and
// the similarity map:  C[M,N] = act(alpha * A[M,K] . W[N,K]^T + bias) (+ residual)
// Reference ops replaced: F.linear inside nn.MultiheadAttention, mlp.c_fc / c_proj
// (open_clip/transformer.py:204-215,234-254), conv1 as a GEMM (:560), `@ self.proj` (:768-770).
//
// Design (gfx950): 128x128x64 block tile, 4 waves (2x2), each wave 64x64 as 4x4 tiles of
// v_mfma_f32_16x16x32_bf16.  Both operands are K-contiguous, staged HBM->LDS with
// global_load_lds (16 B/lane, no VGPR round trip) into a lane-linear image whose 16-B chunks are
// XOR-swizzled on the SOURCE side (chunk ^= (row>>1)&7) so the ds_read_b128 fragment reads of 16
// different rows hit 16 different bank groups.  The MFMA operands are swapped (W rows feed the
// A port, activation rows the B port) so every lane ends up owning 4 CONSECUTIVE output columns
// of one output row: bias / residual / store are 8- or 16-byte vector accesses, no transpose.
// Double-buffered LDS, one barrier per K tile.
#include "common.h"
#include <type_traits>

#ifndef SG_PS_ABL
#define SG_PS_ABL 0                                         // tuning builds only (tools/ablate_persist.sh): parts of the persistent GEMM switched off
#endif

#ifndef SG_H2_PROD_ABL
#define SG_H2_PROD_ABL 0   // tuning builds only (tools): the two-plane producer epilogue without its copy (1), its statistics (2), both (3)
#endif
namespace sg {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Stage a 128 x 64 bf16 tile: 4 passes, each wave-instruction writes 1 KiB = 8 rows x 128 B.
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ P, int64_t ld, int row0, int max_row, int k0,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int slab = p * 4 + wave;                         // 1 KiB slab index (wave-uniform)
    const int r = slab * 8 + (lane >> 3);
    const int c = lane & 7;
    const int g = c ^ ((r >> 1) & 7);                      // source chunk that lives at LDS chunk c
    int grow = row0 + r;
    grow = grow < max_row ? grow : max_row;                // clamp: rows past the edge are never stored
    const bf16_t* src = P + (int64_t)grow * ld + k0 + g * 8;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(lds_tile + slab * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <int ACT, bool C_BF16, bool VEC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmBf16Args a) {
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];   // [buf][A|W]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int tiles_n = (a.N + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.y;
  const bf16_t* A = a.A + (int64_t)z * a.strideA;
  const bf16_t* W = a.W + (int64_t)z * a.strideW;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = a.K / BK;
  stage_tile(A, a.lda, m0, a.M - 1, 0, lds, wave, lane);
  stage_tile(W, a.ldw, n0, a.N - 1, 0, lds + TILE_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    char* bufA = lds + cur * 2 * TILE_BYTES;
    char* bufW = bufA + TILE_BYTES;
    if (t + 1 < nt) {
      char* nA = lds + (cur ^ 1) * 2 * TILE_BYTES;
      stage_tile(A, a.lda, m0, a.M - 1, (t + 1) * BK, nA, wave, lane);
      stage_tile(W, a.ldw, n0, a.N - 1, (t + 1) * BK, nA + TILE_BYTES, wave, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int chunk = kk * 4 + (lane >> 4);
      bf16x8 fa[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag(bufA, wave_m * 64 + i * 16 + (lane & 15), chunk);
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[j] = read_frag(bufW, wave_n * 64 + j * 16 + (lane & 15), chunk);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: lane owns row m = .. + (lane & 15), columns n = .. + (lane >> 4) * 4 + {0..3}
  const float* res = a.residual ? a.residual + (int64_t)z * a.strideC : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wave_m * 64 + i * 16 + (lane & 15);
    if (m >= a.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wave_n * 64 + j * 16 + (lane >> 4) * 4;
      if (n >= a.N) continue;
      float v[4] = {acc[i][j][0] * a.alpha, acc[i][j][1] * a.alpha, acc[i][j][2] * a.alpha, acc[i][j][3] * a.alpha};
      if (VEC) {
        if (a.bias) {
          const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (ACT == ACT_QUICK_GELU) v[e] = quick_gelu(v[e]);
          if (ACT == ACT_GELU) v[e] = erf_gelu_fast(v[e]);
        }
        if (res) {
          const float4 r = *reinterpret_cast<const float4*>(res + (int64_t)m * a.ldr + n);
          v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        if (C_BF16) {
          bf16_t* C = reinterpret_cast<bf16_t*>(a.C) + (int64_t)z * a.strideC + (int64_t)m * a.ldc + n;
          uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
          *reinterpret_cast<uint2*>(C) = o;
        } else {
          float* C = reinterpret_cast<float*>(a.C) + (int64_t)z * a.strideC + (int64_t)m * a.ldc + n;
          *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= a.N) break;
          float x = v[e];
          if (a.bias) x += a.bias[n + e];
          if (ACT == ACT_QUICK_GELU) x = quick_gelu(x);
          if (ACT == ACT_GELU) x = erf_gelu_fast(x);
          if (res) x += res[(int64_t)m * a.ldr + n + e];
          if (C_BF16) reinterpret_cast<bf16_t*>(a.C)[(int64_t)z * a.strideC + (int64_t)m * a.ldc + n + e] = f2bf(x);
          else reinterpret_cast<float*>(a.C)[(int64_t)z * a.strideC + (int64_t)m * a.ldc + n + e] = x;
        }
      }
    }
  }
}

// ---- coalescing epilogue shared by the ring / ping-pong kernels ----------------------------------------------------------------
// The MFMA fragment leaves each lane with 4 consecutive columns of one row: stored directly that is 32-byte (bf16) / 64-byte (f32)
// pieces of 16 different rows per instruction, and the measured store rate was 1.3 TB/s.  Instead every 16-row strip of the wave's
// tile goes through a private LDS patch (row stride TN+4 floats: conflict-free 16-byte writes) and comes back row-contiguous, so a
// store instruction writes WHOLE 128-byte lines (8 rows x 128 B bf16, 4 rows x 256 B f32); bias / activation are applied before
// the patch, the f32 residual is added on the way out with equally coalesced loads.
// SPEC > 0 fixes the hot combinations at compile time (as gemm_bf16_persist's epilogue does): 1 = 2-byte output, no activation (QKV);
// 2 / 3 = MXFP8 output after QuickGELU / GELU (fp8 fc); 4 = f32 output + f32 residual, no activation (out-proj, proj);
// 6 / 7 = 2-byte output after QuickGELU / GELU (fc on the small-launch path).
// SPLIT (SG_PREC_F16X2): a "2-byte" C is the two-plane f16 form (8 columns = one 32-byte storage group per lane, 256 contiguous bytes per
// row and instruction); activations are the exact forms of parity mode (expf / erff), not the hardware-approximation ones.
template <int MI, int NI, bool F16 = false, int SPEC = 0, bool SPLIT = false>
__device__ __forceinline__ void epilogue_store(f32x4 (&acc)[MI][NI], const GemmBf16Args& a, int act_rt, int c_bf16_rt, int z, int row0,
                                               int col0, float* patch, int lane) {
  constexpr int TN = NI * 16, LDP = TN + 4;
  const int act = SPEC == 0 ? act_rt : ((SPEC == 2 || SPEC == 6) ? (int)ACT_QUICK_GELU : (SPEC == 3 || SPEC == 7) ? (int)ACT_GELU : (int)ACT_NONE);
  const int c_bf16 = SPEC == 0 ? c_bf16_rt : (SPEC != 4);
  const bool mx_out = SPEC == 0 ? a.c_mx != nullptr : (SPEC == 2 || SPEC == 3);
  const float* res = (SPEC == 0 || SPEC == 4) ? (a.residual ? a.residual + (int64_t)z * a.strideC : nullptr) : nullptr;
  float4 bias4[NI], cs4[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = col0 + j * 16 + (lane >> 4) * 4;
    bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    cs4[j] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.bias && n + 3 < a.N) bias4[j] = *reinterpret_cast<const float4*>(a.bias + n);
    if (a.col_scale && n + 3 < a.N) cs4[j] = *reinterpret_cast<const float4*>(a.col_scale + n);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float al = a.alpha;
    if (a.row_scale) { int m = row0 + i * 16 + (lane & 15); m = m < a.M ? m : a.M - 1; al *= a.row_scale[m]; }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float v[4] = {acc[i][j][0] * (al * cs4[j].x) + bias4[j].x, acc[i][j][1] * (al * cs4[j].y) + bias4[j].y,
                    acc[i][j][2] * (al * cs4[j].z) + bias4[j].z, acc[i][j][3] * (al * cs4[j].w) + bias4[j].w};
      if (act == ACT_QUICK_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = SPLIT ? quick_gelu_split(v[e]) : quick_gelu(v[e]);
      } else if (act == ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = SPLIT ? erf_gelu(v[e]) : erf_gelu_fast(v[e]);
      }
      *reinterpret_cast<float4*>(patch + (lane & 15) * LDP + j * 16 + (lane >> 4) * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    const int rbase = row0 + i * 16;
    if (c_bf16) {                                          // 8 lanes x 16 B per row, 8 rows per instruction
      constexpr int LPR = TN / 8;                          // lanes per row
      constexpr int RPP = 64 / LPR;                        // rows per pass
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += RPP) {
        const int r = r0 + lane / LPR, cq = (lane % LPR) * 8;
        const int m = rbase + r, n = col0 + cq;
        if (mx_out) {                                       // MXFP8 output: e4m3 + one E8M0 scale per 32 columns (4 adjacent lanes of a row)
          static_assert(TN % 32 == 0, "MX blocks are 32 columns");
          const bool ok = r < 16 && m < a.M && n < a.N;
          const float4 x0 = *reinterpret_cast<const float4*>(patch + r * LDP + cq);
          const float4 x1 = *reinterpret_cast<const float4*>(patch + r * LDP + cq + 4);
          float am = fmaxf(fmaxf(fmaxf(fabsf(x0.x), fabsf(x0.y)), fmaxf(fabsf(x0.z), fabsf(x0.w))),
                           fmaxf(fmaxf(fabsf(x1.x), fabsf(x1.y)), fmaxf(fabsf(x1.z), fabsf(x1.w))));
          am = fmaxf(am, dpp_f32<0xB1>(am)); am = fmaxf(am, dpp_f32<0x4E>(am));   // quad xor 1, quad xor 2
          // smallest power of two 2^(E-127) with amax / 2^(E-127) <= 448 = 1.75 * 2^8: exponent of amax - 8, + 1 if its mantissa exceeds 1.75's
          const uint32_t ab = __float_as_uint(am);
          int E = (int)(ab >> 23) - 8 + ((ab & 0x7fffffu) > 0x600000u ? 1 : 0);
          E = E < 0 ? 0 : E;
          const float inv = __uint_as_float((uint32_t)(254 - E) << 23);           // 2^(127 - E)
          if (ok) {
            int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x0.x * inv, x0.y * inv, 0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x0.z * inv, x0.w * inv, w0, true);
            int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x1.x * inv, x1.y * inv, 0, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x1.z * inv, x1.w * inv, w1, true);
            *reinterpret_cast<uint2*>(a.c_mx + (int64_t)m * a.ldc + n) = make_uint2((uint32_t)w0, (uint32_t)w1);
            if ((lane & 3) == 0) { const int nb = n >> 5; a.c_mx_scale[((int64_t)(nb >> 2) * a.M + m) * 4 + (nb & 3)] = (uint8_t)E; }
          }
        } else if (r < 16 && m < a.M && n < a.N) {
          float4 x0 = *reinterpret_cast<const float4*>(patch + r * LDP + cq);
          float4 x1 = *reinterpret_cast<const float4*>(patch + r * LDP + cq + 4);
          if (SPEC == 0 && !SPLIT && res && a.res_half) {  // 2-byte residual in the operands' type (JBU fixup chain in f16)
            const uint4 rr = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.residual) + (int64_t)z * a.strideC + (int64_t)m * a.ldr + n);
            const uint32_t w[4] = {rr.x, rr.y, rr.z, rr.w};
            float rv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (F16) { rv[2 * e] = h2f(f16_t{(uint16_t)(w[e] & 0xffffu)}); rv[2 * e + 1] = h2f(f16_t{(uint16_t)(w[e] >> 16)}); }
              else { rv[2 * e] = __uint_as_float(w[e] << 16); rv[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
            }
            x0.x += rv[0]; x0.y += rv[1]; x0.z += rv[2]; x0.w += rv[3]; x1.x += rv[4]; x1.y += rv[5]; x1.z += rv[6]; x1.w += rv[7];
          }
          if constexpr (SPLIT) {
            const float v8[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            store_h2x8(reinterpret_cast<h2_t*>(a.C) + (int64_t)z * a.strideC + (int64_t)m * a.ldc + n, v8);
            continue;
          }
          uint4 o; o.x = pack_half2<F16>(x0.x, x0.y); o.y = pack_half2<F16>(x0.z, x0.w); o.z = pack_half2<F16>(x1.x, x1.y); o.w = pack_half2<F16>(x1.z, x1.w);
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.C) + (int64_t)z * a.strideC + (int64_t)m * a.ldc + n) = o;
        }
      }
    } else {                                               // 16 lanes x 16 B per row, 4 rows per instruction
      constexpr int LPR = TN / 4;
      constexpr int RPP = 64 / LPR;
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += RPP) {
        const int r = r0 + lane / LPR, cq = (lane % LPR) * 4;
        const int m = rbase + r, n = col0 + cq;
        if (r < 16 && m < a.M && n < a.N) {
          float4 x = *reinterpret_cast<const float4*>(patch + r * LDP + cq);
          if (SPEC == 4 || res) {
            const float4 rr = *reinterpret_cast<const float4*>(res + (int64_t)m * a.ldr + n);
            x.x += rr.x; x.y += rr.y; x.z += rr.z; x.w += rr.w;
          }
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + (int64_t)z * a.strideC + (int64_t)m * a.ldc + n) = x;
        }
      }
    }
  }
}

// ---- ring-buffered variant ---------------------------------------------------------------------------------------------
// Same staging image, fragment reads and swapped-operand MFMAs as above, but (1) the block tile and wave grid are template
// parameters (256x128 / 256x256 tiles with 8 waves halve the L2 bytes per FLOP), (2) the LDS is a ring of STAGES K-tiles
// filled by global_load_lds that stay IN FLIGHT across the barrier: a counted `s_waitcnt vmcnt(N)` (never 0 in steady state)
// retires only the tile about to be read, a raw s_barrier publishes it, and the slot freed by the previous iteration is
// refilled immediately -- so STAGES-1 tiles of HBM/L2 latency are hidden behind the MFMA phase instead of one.
//   iteration t:  vmcnt((STAGES-2) * G) ; s_barrier ; stage(t + STAGES - 1) ; ds_read + MFMA on slot t % STAGES
// Slot (t-1) % STAGES is rewritten only after every wave has passed the barrier that follows its last read of it.
__device__ __forceinline__ int swz32(int r) { return (0 - (r >> 2)) & 3; }
__device__ __forceinline__ bf16x8 read_frag32(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((chunk ^ swz32(row)) << 4));
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// FP8: the operands are OCP e4m3 bytes.  The kernel is handed the SAME byte image as a bf16 matrix of half the width (K tile of
// 64 'bf16' = 128 fp8 per row), so staging, swizzle and fragment reads are unchanged; the two 16-byte fragment reads of a K tile
// are concatenated into the 32-byte operand of ONE v_mfma_f32_16x16x128_f8f6f4 (both operands use the same K permutation).
// The legacy v_mfma_f32_16x16x32_fp8_fp8 runs at the bf16 rate on gfx950 (tools/mfma_rate.hip: 2.1 vs 4.8 PFLOP/s), so it is not used.
// MXA (fp8 only): A carries MX block scales (GemmBf16Args::a_mx): every K tile stages one more piece -- the dwords holding the four E8M0
// scales of each A row -- behind the operand image, and the MFMA's second scale operand gets the lane's own byte.
// SPLIT (SG_PREC_F16X2): the operands are two-plane f16 rows seen as f16 matrices of twice the width (host side: K, lda, ldw doubled), so a
// 128-byte K tile holds 32 elements as four [8 hi | 8 lo] groups.  The SOURCE-side chunk permutation also de-interleaves the planes: LDS
// chunks 0-3 of a row are the four hi chunks, 4-7 the four lo chunks -- the fragment reads are then exactly the plain kernel's kk = 0 / 1
// reads (conflict-free as they stand), and a K tile is ONE k-step of three MFMAs: W_hi.A_hi + W_lo.A_hi + W_hi.A_lo.
template <int BM_, int BN_, int WM, int WN, int STAGES, int ABLATE = 0, int BKT = 64, bool FP8 = false, bool F16 = false, bool MXA = false, int SPEC = 0, bool SPLIT = false>   // ABLATE (tuning only): 1 = no loads in the loop, 2 = no MFMA
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_bf16_ring(GemmBf16Args a, int act, int c_bf16, int vec) {
  static_assert(!SPLIT || (BKT == 64 && F16 && !FP8 && !MXA), "two-plane f16: 128-byte K tiles on the f16 MFMA");
  constexpr int NW = WM * WN, TM = BM_ / WM, TN = BN_ / WN, MI = TM / 16, NI = TN / 16;
  constexpr int RPS = BKT == 64 ? 8 : 16;                // rows per 1 KiB slab (row = BKT * 2 bytes)
  constexpr int SLABS = (BM_ + BN_) / RPS;               // 1 KiB slabs per K tile
  constexpr int G = SLABS / NW + (MXA ? 1 : 0);          // global_load_lds per thread per K tile
  constexpr int OPER_BYTES = (BM_ + BN_) * BKT * 2;
  constexpr int STAGE_BYTES = OPER_BYTES + (MXA ? BM_ * 4 : 0);
  static_assert(!MXA || (FP8 && BM_ == 256 && NW >= 4), "MX block scales: fp8 256-row tiles");
  static_assert(SLABS % NW == 0, "slabs must divide over the waves");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WN, wave_n = wave % WN;
  const int tiles_n = (a.N + BN_ - 1) / BN_;
  const int tiles_m = (a.M + BM_ - 1) / BM_;
  // XCD-aware order: consecutive ids of ONE XCD walk the n tiles of one m tile (shared A rows stay in that XCD's L2)
  const int nwg = tiles_m * tiles_n;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + seq;
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int m0 = tile_m * BM_, n0 = tile_n * BN_;
  const int z = blockIdx.y;
  const bf16_t* A = a.A + (int64_t)z * a.strideA;
  const bf16_t* W = a.W + (int64_t)z * a.strideW;

  auto stage = [&](int t, int slot) {
    char* base = lds + slot * STAGE_BYTES;
    if constexpr (MXA) {                                   // 64 rows x 4 scale bytes per wave-instruction; waves >= 4 repeat the rows of waves 0-3 (same bytes, same place)
      int gr = m0 + (wave & 3) * 64 + lane; gr = gr < a.M ? gr : a.M - 1;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a.a_mx + ((int64_t)t * a.M + gr) * 4), (lds_ptr_t)(base + OPER_BYTES + (wave & 3) * 256), 4, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < G - (MXA ? 1 : 0); ++p) {
      const int slab = p * NW + wave;                      // wave-uniform
      const int r = slab * RPS + (BKT == 64 ? (lane >> 3) : (lane >> 2));   // row inside the stacked [A rows | W rows] image
      const int c = BKT == 64 ? (lane & 7) : (lane & 3);
      const bool is_a = slab * RPS < BM_;
      const int rl = is_a ? r : r - BM_;                   // the swizzle is a function of the row inside its own tile
      int gch = BKT == 64 ? (c ^ ((rl >> 1) & 7)) : (c ^ swz32(rl));
      if constexpr (SPLIT) gch = ((gch & 3) << 1) | (gch >> 2);   // logical chunk (plane p, group g) = 4 p + g lives at global chunk 2 g + p
      const bf16_t* src;
      if (is_a) { int gr = m0 + r; gr = gr < a.M ? gr : a.M - 1; src = A + (int64_t)gr * a.lda + t * BKT + gch * 8; }
      else { int gr = n0 + rl; gr = gr < a.N ? gr : a.N - 1; src = W + (int64_t)gr * a.ldw + t * BKT + gch * 8; }
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + slab * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = ABLATE == 4 ? 1 : a.K / BKT;
#pragma unroll
  for (int sidx = 0; sidx < STAGES - 1; ++sidx)
    if (sidx < nt) stage(sidx, sidx);

  for (int t = 0; t < nt; ++t) {
    const int ahead = nt - 1 - t;                          // tiles already issued beyond t
    if (STAGES >= 4 && ahead >= 2) wait_vmcnt<2 * G>();
    else if (STAGES >= 3 && ahead >= 1) wait_vmcnt<(STAGES >= 3 ? G : 0)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ABLATE != 1 && t + STAGES - 1 < nt) stage(t + STAGES - 1, (t + STAGES - 1) % STAGES);
    const char* bufA = lds + (t % STAGES) * STAGE_BYTES;
    const char* bufW = bufA + BM_ * BKT * 2;
    if constexpr (FP8) {
      static_assert(!FP8 || BKT == 64, "fp8 uses the 128-byte K tile");
      typedef __attribute__((ext_vector_type(8))) int i32x8;
      union Op { bf16x8 h[2]; i32x8 v; };
      Op fa8[MI], fw8[NI];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int chunk = kk * 4 + (lane >> 4);            // the instruction's own K order: dwords 0-3 of lane block g are K 16g..16g+15, dwords 4-7 are K 64+16g.. (tools/mx_probe.hip)
#pragma unroll
        for (int i = 0; i < MI; ++i) fa8[i].h[kk] = read_frag(bufA, wave_m * TM + i * 16 + (lane & 15), chunk);
#pragma unroll
        for (int j = 0; j < NI; ++j) fw8[j].h[kk] = read_frag(bufW, wave_n * TN + j * 16 + (lane & 15), chunk);
      }
      if constexpr (MXA) {
        // the scale of K block b (K 32b..32b+31) of row r is taken from lane r + 16 b (NOT "the lane's own 32 bytes": those straddle
        // blocks g/2 and 2 + g/2 -- tools/mx_probe.hip run 4): lane (r, g) supplies byte g of the row's scale dword, in byte 0 (opsel 0)
        const uint32_t* sS = reinterpret_cast<const uint32_t*>(bufA + OPER_BYTES);
        int sa[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) sa[i] = (int)(sS[wave_m * TM + i * 16 + (lane & 15)] >> (8 * (lane >> 4)));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)   // operands are swapped (W rows on the first port): A's block scale is the SECOND scale operand
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw8[j].v, fa8[i].v, acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, sa[i]);
        continue;
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)   // cbsz = blgp = 0: both operands e4m3; scales 0x7f = 2^0 (E8M0), i.e. plain fp8 x fp8
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw8[j].v, fa8[i].v, acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      continue;
    }
    if constexpr (SPLIT) {
      bf16x8 fah[MI], fal[MI], fwh[NI], fwl[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        fwh[j] = read_frag(bufW, wave_n * TN + j * 16 + (lane & 15), lane >> 4);
        fwl[j] = read_frag(bufW, wave_n * TN + j * 16 + (lane & 15), 4 + (lane >> 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        fah[i] = read_frag(bufA, wave_m * TM + i * 16 + (lane & 15), lane >> 4);
        fal[i] = read_frag(bufA, wave_m * TM + i * 16 + (lane & 15), 4 + (lane >> 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          acc[i][j] = mfma_16x16x32<true>(fwh[j], fah[i], acc[i][j]);
          acc[i][j] = mfma_16x16x32<true>(fwl[j], fah[i], acc[i][j]);
          acc[i][j] = mfma_16x16x32<true>(fwh[j], fal[i], acc[i][j]);
        }
      continue;
    }
#pragma unroll
    for (int kk = 0; kk < BKT / 32; ++kk) {
      const int chunk = kk * 4 + (lane >> 4);
      bf16x8 fa[MI], fw[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        fa[i] = BKT == 64 ? read_frag(bufA, wave_m * TM + i * 16 + (lane & 15), chunk) : read_frag32(bufA, wave_m * TM + i * 16 + (lane & 15), chunk);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        fw[j] = BKT == 64 ? read_frag(bufW, wave_n * TN + j * 16 + (lane & 15), chunk) : read_frag32(bufW, wave_n * TN + j * 16 + (lane & 15), chunk);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          if (ABLATE != 2) acc[i][j] = mfma_16x16x32<F16>(fw[j], fa[i], acc[i][j]);
          else { asm volatile("" ::"v"(fw[j]), "v"(fa[i])); }
    }
  }

  if (ABLATE == 3) {                                       // tuning only: keep the accumulators live, store (almost) nothing
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 1234.5678f) reinterpret_cast<float*>(a.C)[0] = sum;
    return;
  }
  if (vec) {                                               // N % 8 == 0, aligned: coalesced path through an LDS patch
    __syncthreads();                                       // every wave is done reading the staging buffers
    epilogue_store<MI, NI, F16, SPEC, SPLIT>(acc, a, act, c_bf16, z, m0 + wave_m * TM, n0 + wave_n * TN, reinterpret_cast<float*>(lds) + wave * 16 * (TN + 4), lane);
    return;
  }
  // scalar path (ragged N / unaligned C): f32 C only in SPLIT mode (checked on the host), exact activations there
  const float* res = a.residual ? a.residual + (int64_t)z * a.strideC : nullptr;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wave_m * TM + i * 16 + (lane & 15);
    if (m >= a.M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wave_n * TN + j * 16 + (lane >> 4) * 4;
      if (n >= a.N) continue;
      float v[4] = {acc[i][j][0] * a.alpha, acc[i][j][1] * a.alpha, acc[i][j][2] * a.alpha, acc[i][j][3] * a.alpha};
      if (a.row_scale) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= a.row_scale[m];
      }
      if (a.col_scale) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= a.col_scale[n + e < a.N ? n + e : a.N - 1];
      }
      if (vec) {
        if (a.bias) {
          const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (act == ACT_QUICK_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
        } else if (act == ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = erf_gelu_fast(v[e]);
        }
        if (res) {
          const float4 r = *reinterpret_cast<const float4*>(res + (int64_t)m * a.ldr + n);
          v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        if (c_bf16) {
          bf16_t* C = reinterpret_cast<bf16_t*>(a.C) + (int64_t)z * a.strideC + (int64_t)m * a.ldc + n;
          uint2 o; o.x = pack_half2<F16>(v[0], v[1]); o.y = pack_half2<F16>(v[2], v[3]);
          *reinterpret_cast<uint2*>(C) = o;
        } else {
          float* C = reinterpret_cast<float*>(a.C) + (int64_t)z * a.strideC + (int64_t)m * a.ldc + n;
          *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= a.N) break;
          float x = v[e];
          if (a.bias) x += a.bias[n + e];
          if (act == ACT_QUICK_GELU) x = SPLIT ? quick_gelu_split(x) : quick_gelu(x);
          else if (act == ACT_GELU) x = SPLIT ? erf_gelu(x) : erf_gelu_fast(x);
          if (res) x += res[(int64_t)m * a.ldr + n + e];
          if (c_bf16) reinterpret_cast<bf16_t*>(a.C)[(int64_t)z * a.strideC + (int64_t)m * a.ldc + n + e] = F16 ? f2h(x).bits : f2bf(x);
          else reinterpret_cast<float*>(a.C)[(int64_t)z * a.strideC + (int64_t)m * a.ldc + n + e] = x;
        }
      }
    }
  }
}

// ---- ping-pong variant: 256x256x64 tile, 8 waves = two groups of four that ALTERNATE on the matrix pipe ---------------------
// Group g (waves 4g..4g+3, one per SIMD) owns output rows [128g, 128g+128); wave i of a group owns columns [64i, 64i+64).
// A K tile is four phases of 16 MFMAs (one 64x32 quadrant x K=64).  Every phase is two barrier-delimited segments:
//     READ  : ds_read_b128 of the operands the phase needs (4-12 reads) + this wave's share of the next tile's global_load_lds
//     MFMA  : 16 x v_mfma_f32_16x16x32_bf16
// Group 1 runs ONE barrier behind group 0, so on every SIMD one wave is in its MFMA segment while its partner reads:
// the matrix pipe never waits for LDS latency and the LDS/TA never wait for the MFMAs.
// LDS: 2 buffers x [A 256x64 | W 256x64] bf16 = 128 KiB, same swizzled image as above.  Hand-off rules (slot = barrier interval;
// group 0's READ(t,p) is slot 8t+2p, group 1's is 8t+2p+1):
//   * who loads what for tile t+1: wave (g,i) loads A rows [128g+32i,+32) in READ(t,0) and W rows [64i+32g,+32) in READ(t,1).
//     WAR: those A rows were last read by group g in READ(t-1,2) (>= 4 slots earlier); those W rows in READ(t-1,3) of either
//     group (slots 8t-2 / 8t-1, retired by the lgkmcnt wait that opens slots 8t-1 / 8t) -- the write is issued at slot >= 8t+2.
//   * RAW: every wave drains its own loads (vmcnt(0)) at the end of MFMA(t,3) (slot 8t+7 / 8t+8) before the barrier; the first
//     readers of another group's rows come >= 1 slot after that barrier (group 0 reads group-1-loaded W rows in READ(t+1,1),
//     slot 8t+10; group 1 reads group-0-loaded W rows in READ(t+1,0), slot 8t+9).
// SPLIT (SG_PREC_F16X2): as in gemm_bf16_ring -- the planes are de-interleaved by the source-side chunk permutation, fragment [.][0] is the
// hi plane and [.][1] the lo plane of the tile's 32 elements, and a phase issues 8 x 3 MFMAs (W_hi.A_hi + W_lo.A_hi + W_hi.A_lo).
template <bool F16, bool SPLIT = false>
__global__ __launch_bounds__(512) void gemm_bf16_pingpong(GemmBf16Args a, int act, int c_bf16, int vec) {
  constexpr int PBM = 256, PBN = 256;
  constexpr int BUF_BYTES = (PBM + PBN) * BK * 2;          // 64 KiB
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wi = wave & 3;
  const int tiles_n = (a.N + PBN - 1) / PBN, tiles_m = (a.M + PBM - 1) / PBM;
  const int nwg = tiles_m * tiles_n;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + seq;
  const int m0 = (tile / tiles_n) * PBM, n0 = (tile % tiles_n) * PBN;
  const int z = blockIdx.y;
  const bf16_t* A = a.A + (int64_t)z * a.strideA;
  const bf16_t* W = a.W + (int64_t)z * a.strideW;
  const int nt = a.K / BK;

  // this lane's 4 A rows and 4 W rows of every K tile (one 1 KiB slab = 8 rows per instruction)
  const int sub = lane >> 3, cpos = lane & 7;
  const bf16_t* a_src[4]; const bf16_t* w_src[4]; int a_dst[4], w_dst[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ra = 128 * g + 32 * wi + 8 * p + sub;                       // row inside the A tile
    const int rw = 64 * wi + 32 * g + 8 * p + sub;                        // row inside the W tile
    int gra = m0 + ra; gra = gra < a.M ? gra : a.M - 1;
    int grw = n0 + rw; grw = grw < a.N ? grw : a.N - 1;
    int ca = cpos ^ ((ra >> 1) & 7), cw = cpos ^ ((rw >> 1) & 7);
    if constexpr (SPLIT) { ca = ((ca & 3) << 1) | (ca >> 2); cw = ((cw & 3) << 1) | (cw >> 2); }   // logical chunk 4 p + g <- global chunk 2 g + p
    a_src[p] = A + (int64_t)gra * a.lda + (ca << 3);
    w_src[p] = W + (int64_t)grw * a.ldw + (cw << 3);
    a_dst[p] = (128 * g + 32 * wi + 8 * p) * 128;                         // wave-uniform slab base
    w_dst[p] = PBM * BK * 2 + (64 * wi + 32 * g + 8 * p) * 128;
  }
  auto load_a = [&](int t, int buf) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[p] + t * BK), (lds_ptr_t)(lds + buf * BUF_BYTES + a_dst[p]), 16, 0, 0);
  };
  auto load_w = [&](int t, int buf) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[p] + t * BK), (lds_ptr_t)(lds + buf * BUF_BYTES + w_dst[p]), 16, 0, 0);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[4][2], fw[2][2];                                              // [frag][kk]

  auto read_a = [&](const char* buf, int half) {                           // 64 rows of this group's A half
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fa[i][kk] = read_frag(buf, 128 * g + 64 * half + 16 * i + (lane & 15), kk * 4 + (lane >> 4));
  };
  auto read_w = [&](const char* buf, int half) {                           // 32 rows of this wave's W slice
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fw[j][kk] = read_frag(buf + PBM * BK * 2, 64 * wi + 32 * half + 16 * j + (lane & 15), kk * 4 + (lane >> 4));
  };
#define SG_PP_MFMA(MH, NH)                                                                               \
  do {                                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int kk = 0; kk < (SPLIT ? 3 : 2); ++kk)                                       \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
          acc[4 * (MH) + i][2 * (NH) + j] =                                                              \
              mfma_16x16x32<F16>(fw[j][SPLIT ? (kk == 1) : kk], fa[i][SPLIT ? (kk == 2) : kk], acc[4 * (MH) + i][2 * (NH) + j]); \
    __builtin_amdgcn_s_setprio(0);                                                                       \
  } while (0)
#define SG_PP_SYNC()                                 \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_barrier();                    \
    __builtin_amdgcn_sched_barrier(0);               \
  } while (0)

  // prologue: tile 0 completely, then the one-barrier stagger of group 1
  load_a(0, 0); load_w(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SG_PP_SYNC();
  if (g == 1) SG_PP_SYNC();

  for (int t = 0; t < nt; ++t) {
    const char* buf = lds + (t & 1) * BUF_BYTES;
    const bool more = t + 1 < nt;
    // phase 0: quadrant (rows 0-63, cols 0-31)
    read_w(buf, 0); read_a(buf, 0);
    if (more) load_a(t + 1, (t + 1) & 1);
    SG_PP_SYNC();
    SG_PP_MFMA(0, 0);
    SG_PP_SYNC();
    // phase 1: (rows 0-63, cols 32-63)
    read_w(buf, 1);
    if (more) load_w(t + 1, (t + 1) & 1);
    SG_PP_SYNC();
    SG_PP_MFMA(0, 1);
    SG_PP_SYNC();
    // phase 2: (rows 64-127, cols 32-63)
    read_a(buf, 1);
    SG_PP_SYNC();
    SG_PP_MFMA(1, 1);
    SG_PP_SYNC();
    // phase 3: (rows 64-127, cols 0-31)
    read_w(buf, 0);
    SG_PP_SYNC();
    SG_PP_MFMA(1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of tile t+1 has landed
    SG_PP_SYNC();
  }
  if (g == 0) SG_PP_SYNC();                               // balance the stagger barrier
#undef SG_PP_MFMA
#undef SG_PP_SYNC

  if (vec) {
    __syncthreads();
    epilogue_store<8, 4, F16, 0, SPLIT>(acc, a, act, c_bf16, z, m0 + 128 * g, n0 + 64 * wi, reinterpret_cast<float*>(lds) + wave * 16 * 68, lane);
    return;
  }
  const float* res = a.residual ? a.residual + (int64_t)z * a.strideC : nullptr;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + 128 * g + i * 16 + (lane & 15);
    if (m >= a.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 64 * wi + j * 16 + (lane >> 4) * 4;
      if (n >= a.N) continue;
      float v[4] = {acc[i][j][0] * a.alpha, acc[i][j][1] * a.alpha, acc[i][j][2] * a.alpha, acc[i][j][3] * a.alpha};
      if (vec) {
        if (a.bias) {
          const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (act == ACT_QUICK_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
        } else if (act == ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = erf_gelu_fast(v[e]);
        }
        if (res) {
          const float4 r = *reinterpret_cast<const float4*>(res + (int64_t)m * a.ldr + n);
          v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        if (c_bf16) {
          bf16_t* C = reinterpret_cast<bf16_t*>(a.C) + (int64_t)z * a.strideC + (int64_t)m * a.ldc + n;
          uint2 o; o.x = pack_half2<F16>(v[0], v[1]); o.y = pack_half2<F16>(v[2], v[3]);
          *reinterpret_cast<uint2*>(C) = o;
        } else {
          float* C = reinterpret_cast<float*>(a.C) + (int64_t)z * a.strideC + (int64_t)m * a.ldc + n;
          *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= a.N) break;
          float x = v[e];
          if (a.bias) x += a.bias[n + e];
          if (act == ACT_QUICK_GELU) x = SPLIT ? quick_gelu_split(x) : quick_gelu(x);
          else if (act == ACT_GELU) x = SPLIT ? erf_gelu(x) : erf_gelu_fast(x);
          if (res) x += res[(int64_t)m * a.ldr + n + e];
          if (c_bf16) reinterpret_cast<bf16_t*>(a.C)[(int64_t)z * a.strideC + (int64_t)m * a.ldc + n + e] = F16 ? f2h(x).bits : f2bf(x);
          else reinterpret_cast<float*>(a.C)[(int64_t)z * a.strideC + (int64_t)m * a.ldc + n + e] = x;
        }
      }
    }
  }
}

static int launch_pingpong(const GemmBf16Args& a, int vec, hipStream_t s) {
  const size_t lds = 2 * (256 + 256) * BK * 2;
  auto kern = a.h2 ? gemm_bf16_pingpong<true, true> : a.f16 ? gemm_bf16_pingpong<true> : gemm_bf16_pingpong<false>;
  SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int64_t tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
  SG_REQUIRE(tiles < (1ll << 31) && a.batch < 65536, "gemm_bf16: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)a.batch), dim3(512), lds, s, a, a.act, a.c_is_bf16, vec);
  return SG_OK;
}

// ---- ping-pong, K tile 32, FOUR-deep LDS ring ---------------------------------------------------------------------------------
// Same two-group alternation as gemm_bf16_pingpong, but a K tile is 32 deep (2 phases x 16 MFMAs) and the 128 KiB of LDS hold a
// ring of 4 tiles, so the loads of tile t+3 are issued while tile t computes: three tiles (~3000 cycles) of HBM / fabric latency
// are covered instead of one, with only 2 global_load_lds per wave per phase (their issue cost stays inside a 256-cycle segment).
// LDS image per tile: [A 256 rows x 64 B | W 256 rows x 64 B]; 16-byte chunk c of row r sits at position c ^ swz(r),
// swz(r) = (-(r >> 2)) & 3, which makes every ds_read_b128 lane group of a 16-row fragment read hit 16 distinct bank slots.
// Slot = barrier interval; group 0: READ(t,0) = 4t, MFMA(t,0) = 4t+1, READ(t,1) = 4t+2, MFMA(t,1) = 4t+3; group 1 one later.
//   loads : wave (g,i) issues for tile t+3   A rows [128g+32i,+32) in READ(t,0),   W rows [64i+32g,+32) in READ(t,1)
//   WAR   : ring slot (t+3)&3 was last read for tile t-1: A-half g by group g in READ(t-1,1) (slot 4t-2+g, retired when the
//           next segment opens), W rows in READ(t-1,0) (slots 4t-4 / 4t-3); the writes are issued at slots >= 4t+g / 4t+2+g.
//   RAW   : each wave retires its own pieces of tile t+1 with a COUNTED vmcnt (tiles t+2, t+3 stay in flight) just before
//           the barrier that closes slot 4t+3 (group 0: after MFMA(t,1); group 1: after READ(t,1)); tile t+1 is first read
//           in slot 4t+4.

template <int ABLATE>
__global__ __launch_bounds__(512) void gemm_bf16_pp32(GemmBf16Args a, int act, int c_bf16, int vec) {
  constexpr int PBM = 256, PBN = 256, KT32 = 32;
  constexpr int TILE_B = (PBM + PBN) * KT32 * 2;             // 32 KiB per ring slot
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wi = wave & 3;
  const int tiles_n = (a.N + PBN - 1) / PBN, tiles_m = (a.M + PBM - 1) / PBM;
  const int nwg = tiles_m * tiles_n;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + seq;
  const int m0 = (tile / tiles_n) * PBM, n0 = (tile % tiles_n) * PBN;
  const int z = blockIdx.y;
  const bf16_t* A = a.A + (int64_t)z * a.strideA;
  const bf16_t* W = a.W + (int64_t)z * a.strideW;
  const int nt = ABLATE == 4 ? 4 : a.K / KT32;

  // this lane's source rows: 2 A slabs and 2 W slabs (16 rows x 64 B each) per K tile
  const int srow = lane >> 2, cpos = lane & 3;
  const bf16_t* a_src[2]; const bf16_t* w_src[2]; int a_dst[2], w_dst[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int ra = 128 * g + 32 * wi + 16 * p + srow;
    const int rw = 64 * wi + 32 * g + 16 * p + srow;
    int gra = m0 + ra; gra = gra < a.M ? gra : a.M - 1;
    int grw = n0 + rw; grw = grw < a.N ? grw : a.N - 1;
    a_src[p] = A + (int64_t)gra * a.lda + ((cpos ^ swz32(ra)) << 3);
    w_src[p] = W + (int64_t)grw * a.ldw + ((cpos ^ swz32(rw)) << 3);
    a_dst[p] = (128 * g + 32 * wi + 16 * p) * 64;
    w_dst[p] = PBM * KT32 * 2 + (64 * wi + 32 * g + 16 * p) * 64;
  }
  auto load_a = [&](int t) {
    char* base = lds + (t & 3) * TILE_B;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[p] + t * KT32), (lds_ptr_t)(base + a_dst[p]), 16, 0, 0);
  };
  auto load_w = [&](int t) {
    char* base = lds + (t & 3) * TILE_B;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[p] + t * KT32), (lds_ptr_t)(base + w_dst[p]), 16, 0, 0);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[4], fw[4];

#define SG_P32_SYNC()                                \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_barrier();                    \
    __builtin_amdgcn_sched_barrier(0);               \
  } while (0)
#define SG_P32_MFMA(MH)                                                                                  \
  do {                                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
        if (ABLATE != 2) acc[4 * (MH) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[4 * (MH) + i][j], 0, 0, 0); \
        else asm volatile("" ::"v"(fw[j]), "v"(fa[i]));                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                       \
  } while (0)
#define SG_P32_WAIT(T)                                                                                   \
  do {                                                                                                   \
    const int beyond = nt - 2 - (T);                                                                     \
    if (beyond >= 2) wait_vmcnt<8>(); else if (beyond == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();       \
  } while (0)

  // prologue: tiles 0..2 in flight, tile 0 retired; then the one-barrier stagger of group 1
#pragma unroll
  for (int t = 0; t < 3; ++t)
    if (t < nt) { load_a(t); load_w(t); }
  SG_P32_WAIT(-1);
  SG_P32_SYNC();
  if (g == 1) SG_P32_SYNC();

  for (int t = 0; t < nt; ++t) {
    const char* tA = lds + (t & 3) * TILE_B;
    const char* tW = tA + PBM * KT32 * 2;
    const bool more = (ABLATE != 1) && (t + 3 < nt);
    // phase 0: rows 0-63 of the group's half x the wave's 64 columns
#pragma unroll
    for (int j = 0; j < 4; ++j) fw[j] = read_frag32(tW, 64 * wi + 16 * j + (lane & 15), lane >> 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = read_frag32(tA, 128 * g + 16 * i + (lane & 15), lane >> 4);
    if (more) load_a(t + 3);
    SG_P32_SYNC();
    SG_P32_MFMA(0);
    SG_P32_SYNC();
    // phase 1: rows 64-127
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = read_frag32(tA, 128 * g + 64 + 16 * i + (lane & 15), lane >> 4);
    if (more) load_w(t + 3);
    if (g == 1) SG_P32_WAIT(t);
    SG_P32_SYNC();
    SG_P32_MFMA(1);
    if (g == 0) SG_P32_WAIT(t);
    SG_P32_SYNC();
  }
  if (g == 0) SG_P32_SYNC();
#undef SG_P32_SYNC
#undef SG_P32_MFMA
#undef SG_P32_WAIT

  if (ABLATE == 3) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 1234.5678f) reinterpret_cast<float*>(a.C)[0] = sum;
    return;
  }
  __syncthreads();
  epilogue_store<8, 4>(acc, a, act, c_bf16, z, m0 + 128 * g, n0 + 64 * wi, reinterpret_cast<float*>(lds) + wave * 16 * 68, lane);
}

template <int ABLATE>
static int launch_pp32(const GemmBf16Args& a, hipStream_t s) {
  const size_t lds = 4 * (256 + 256) * 32 * 2;
  auto kern = gemm_bf16_pp32<ABLATE>;
  SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int64_t tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
  SG_REQUIRE(tiles < (1ll << 31) && a.batch < 65536, "gemm_bf16: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)a.batch), dim3(512), lds, s, a, a.act, a.c_is_bf16, 1);
  return SG_OK;
}

static thread_local int g_persist_grid_cap = 0;             // tuning (sg_set_gemm_config(2000 + n)): at most n workgroups for the persistent kernel (0 = one per CU)
static thread_local int g_gemm_order = -1;                 // tuning (sg_set_gemm_config(1000 + v)): -1 automatic N-group size, 0 raster order, v > 0 forced N-group size
// ---- persistent ping-pong: the production kernel for the large ViT linears ---------------------------------------------------------
// gemm_bf16_pp32's ring (K tile 32, four slots) with three changes measured to matter:
//   * ONE phase of 32 MFMAs per K tile (12 ds_read_b128 + 4 global_load_lds per wave per READ segment): the barrier + LDS-latency
//     overhead of a READ segment is amortised over a 512-cycle MFMA segment instead of 256;
//   * PERSISTENT workgroups (grid = #CUs): the K-tile stream runs straight across output tiles, so the loads of the next
//     tile's first K tiles are already in flight while the current tile's epilogue drains -- no exposed prologue per tile;
//   * asymmetric issue so every piece gets >= 3 slots of flight:   group 0 READ(s): W(s+2), A(s+3)    group 1 READ(s): A(s+3), W(s+3)
// Slots (barrier intervals): group 0 READ(s) = 2s, MFMA(s) = 2s+1; group 1 READ(s) = 2s+1, MFMA(s) = 2s+2.
//   WAR  ring slot (s+3)&3 = (s-1)&3: A-half g last read by group g in READ(s-1) (slot 2s-2+g), retired when the next segment opens
//        (2s-1+g); written at slot 2s+g.  W rows of slot (s-1)&3 last read at slots 2s-2 / 2s-1, retired by the start of slot 2s;
//        group 1 writes them at slot 2s+1.  Group 0's W(s+2) goes to slot (s-2)&3, idle since slot 2s-3.
//   RAW  K tile s+1 is first read in slot 2s+2.  Group 0 retires its pieces of it after MFMA(s) (allowed outstanding:
//        A(s+2), W(s+2), A(s+3) = 6), group 1 after READ(s) (tiles s+2, s+3 = 8); both before the barrier closing slot 2s+1.
//   tile end: group 0 takes one extra barrier (both groups are then past every read of the tile's last K tile), every wave runs the
//        coalescing epilogue through a private patch inside the just-consumed ring slot, one barrier, group 1 re-staggers.
// MODE 1 = the consumer side of a folded LayerNorm (2-byte output only), MODE 2 = the producer side (f32 output + 2-byte copy + slice
// statistics): their own instantiations, so that the per-row / per-column factors of the one, the statistics of the other and the plain
// form's deeper residual pipeline never hold registers at the same time.
// SPLIT (two-plane f16, MODE 0 / SPEC 0 only): a "2-byte" C is written as [8 hi | 8 lo] storage groups; exact activations.
template <int MI, int NI, bool F16, int MODE = 0, int SPEC = 0, bool SPLIT = false>
__device__ __forceinline__ void epilogue_store8(f32x4 (&acc)[MI][NI], const GemmBf16Args& a, int act_rt, int c_bf16_rt, int z, int row0,
                                                int col0, float* patch, int lane, float* statbuf = nullptr) {
  static_assert(!SPLIT || SPEC != 5, "two-plane f16: no row-dot form");
  constexpr bool LN = MODE == 1, PROD = MODE == 2;
  // SPEC > 0: activation (SPEC - 1), output type (2-byte in MODE 0 / 1, f32 in MODE 2) and the presence of a residual (MODE 2 only) are
  // compile-time constants -- straight-line strips without the run-time branches (measured on the QKV shape: -3.3 %; overlapping the
  // strips' LDS round trips on top of that brought nothing: the epilogue is bound by store issue).
  // SPEC == 0 keeps every choice at run time (the rarely used combinations and the row-dot form).
  // SPEC == 4 (MODE 0 only): f32 output + f32 residual, no activation -- the plain residual GEMM (fp8 mode's out-proj, the last block, the text tower).
  // SPEC == 5 (MODE 0 only): the row-dot form of the JBU tail (sg_jbu_logits) -- 2-byte residual, no activation, nothing stored but the slice sums.
  constexpr bool RES32 = MODE == 0 && SPEC == 4;
  constexpr bool RDOT = MODE == 0 && SPEC == 5;
  const int act = SPEC > 0 ? ((RES32 || RDOT) ? (int)ACT_NONE : SPEC - 1) : act_rt;
  const int c_bf16 = (LN || (SPEC > 0 && !PROD && !RES32 && !RDOT)) ? 1 : ((PROD || RES32 || RDOT) ? 0 : c_bf16_rt);
  constexpr int TN = NI * 16, LDP = TN + 4;
  const float* res = a.residual ? a.residual + (int64_t)z * a.strideC : nullptr;
  float4 bias4[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = col0 + j * 16 + (lane >> 4) * 4;
    bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias && n + 3 < a.N) bias4[j] = *reinterpret_cast<const float4*>(a.bias + n);
  }
  // folded LayerNorm, consumer side (wave-uniform): per-row (mean, rstd) of the 8 rows this lane's accumulators belong to, per-column c
  float4 lnc4[LN ? NI : 1];
  auto ln_row = [&](int i) -> float2 {                     // (mean, rstd) of the row this lane's accumulators of strip pair i belong to
    int m = row0 + i * 16 + (lane & 15); m = m < a.M ? m : a.M - 1;
    return *reinterpret_cast<const float2*>(a.ln_stats + 2 * (int64_t)m);
  };
  float2 ln_cur = make_float2(0.f, 1.f), ln_nxt = ln_cur;  // fetched two strip pairs ahead
  if constexpr (LN) {
    ln_cur = ln_row(0); ln_nxt = ln_row(1);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = col0 + j * 16 + (lane >> 4) * 4;
      lnc4[j] = (n + 3 < a.N) ? *reinterpret_cast<const float4*>(a.ln_c + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // f32 output with residual: the residual of strip t + RD is requested while strip t goes through the patch, so RD strips of HBM
  // latency are in flight per wave instead of one dependent load -> add -> store chain per strip (addresses clamped, stores guarded)
  constexpr int RD = PROD ? 4 : 5;
  constexpr int LPRF = TN / 4, RPPF = 64 / LPRF, NPASS = 8 / RPPF;
  float4 rbuf[RD][NPASS];
  auto fetch_res = [&](int t, float4 (&dst)[NPASS]) {
    const int rb = row0 + (t >> 1) * 16 + (t & 1) * 8;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      int m = rb + ps * RPPF + lane / LPRF; m = m < a.M ? m : a.M - 1;
      int n = col0 + (lane % LPRF) * 4; n = n < a.N ? n : a.N - 4;
      if (RDOT || (SPEC == 0 && a.rowdot && a.rowdot_res_bf16)) {   // the JBU tail keeps x in bf16 only (wave-uniform branch)
        const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(res) + (int64_t)m * a.ldr + n);
        dst[ps] = make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                              __uint_as_float(raw.y & 0xffff0000u));
      } else dst[ps] = *reinterpret_cast<const float4*>(res + (int64_t)m * a.ldr + n);
    }
  };
  const bool pipe_res = ((PROD && SPEC > 0) || RES32 || RDOT) ? true : (!LN && SPEC == 0 && res != nullptr && !c_bf16);
  if (pipe_res) {
#pragma unroll
    for (int t = 0; t < RD; ++t) fetch_res(t, rbuf[t]);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float4 v4[NI];
    const float2 ln_use = ln_cur;
    if constexpr (LN) { ln_cur = ln_nxt; if (i + 2 < MI) ln_nxt = ln_row(i + 2); }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float v[4] = {acc[i][j][0] * a.alpha + bias4[j].x, acc[i][j][1] * a.alpha + bias4[j].y, acc[i][j][2] * a.alpha + bias4[j].z,
                    acc[i][j][3] * a.alpha + bias4[j].w};
      if constexpr (LN) {                                  // rstd (x.W'^T - mean c) + b'
        const float mu = ln_use.x, rs = ln_use.y;
        v[0] = (acc[i][j][0] - mu * lnc4[j].x) * rs + bias4[j].x; v[1] = (acc[i][j][1] - mu * lnc4[j].y) * rs + bias4[j].y;
        v[2] = (acc[i][j][2] - mu * lnc4[j].z) * rs + bias4[j].z; v[3] = (acc[i][j][3] - mu * lnc4[j].w) * rs + bias4[j].w;
      }
      if (act == ACT_QUICK_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = SPLIT ? quick_gelu_split(v[e]) : quick_gelu(v[e]);
      } else if (act == ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = SPLIT ? erf_gelu(v[e]) : erf_gelu_fast(v[e]);
      }
      v4[j] = make_float4(v[0], v[1], v[2], v[3]);
    }
#if SG_PS_ABL == 6
    // tuning build (WRONG element order by design): what would an epilogue cost that transposes in registers instead of through the LDS
    // patch?  Same bytes, same full-line store pattern (8 rows x 128 B per instruction), no LDS round trip.
    if (c_bf16 && NI == 4) {
      uint4 o0, o1;
      o0.x = pack_half2<F16>(v4[0].x, v4[0].y); o0.y = pack_half2<F16>(v4[0].z, v4[0].w); o0.z = pack_half2<F16>(v4[1].x, v4[1].y); o0.w = pack_half2<F16>(v4[1].z, v4[1].w);
      o1.x = pack_half2<F16>(v4[2].x, v4[2].y); o1.y = pack_half2<F16>(v4[2].z, v4[2].w); o1.z = pack_half2<F16>(v4[3].x, v4[3].y); o1.w = pack_half2<F16>(v4[3].z, v4[3].w);
      const int m = row0 + i * 16 + (lane >> 3), n = col0 + (lane & 7) * 8;
      if (m < a.M && n < a.N) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.C) + (int64_t)m * a.ldc + n) = o0;
      if (m + 8 < a.M && n < a.N) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.C) + (int64_t)(m + 8) * a.ldc + n) = o1;
      continue;
    }
#endif
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {                       // two 8-row strips per 16-row MFMA tile
      if (((lane >> 3) & 1) == hh) {
#pragma unroll
        for (int j = 0; j < NI; ++j) *reinterpret_cast<float4*>(patch + (lane & 7) * LDP + j * 16 + (lane >> 4) * 4) = v4[j];
      }
      // Keep the patch READS below out of the half-wave block above: with the output type a compile-time constant (MODE 1 / 2) hipcc 7.2
      // sank the first ds_read_b128 of a strip into the exec-masked write block, so the other half of the wave kept the previous strip's
      // values (found as rows 8-11, 16-19, ... of the producer form carrying the GEMM part of rows 0-3, 8-11, ...).
      int pofs = 0;                                        // an opaque 0 added to every patch read address below: the reads depend on a
      if constexpr (MODE != 0 || SPEC != 0) asm volatile("" : "+v"(pofs));   // statement that follows the block, so they cannot be moved into it
      const int rbase = row0 + i * 16 + hh * 8;
      if (c_bf16) {                                        // 8 lanes x 16 B per row: one instruction stores the whole strip
        constexpr int LPR = TN / 8;
        const int r = lane / LPR, cq = (lane % LPR) * 8;
        const int m = rbase + r, n = col0 + cq;
        if (r < 8 && m < a.M && n < a.N) {
          const float4 x0 = *reinterpret_cast<const float4*>(patch + r * LDP + cq + pofs);
          const float4 x1 = *reinterpret_cast<const float4*>(patch + r * LDP + cq + 4 + pofs);
          if constexpr (SPLIT) {
            const float v8[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            store_h2x8(reinterpret_cast<h2_t*>(a.C) + (int64_t)z * a.strideC + (int64_t)m * a.ldc + n, v8);
          } else {
          uint4 o; o.x = pack_half2<F16>(x0.x, x0.y); o.y = pack_half2<F16>(x0.z, x0.w); o.z = pack_half2<F16>(x1.x, x1.y); o.w = pack_half2<F16>(x1.z, x1.w);
#if SG_PS_ABL == 5
          if (o.x == 0x12345678u)                              // tuning build: the patch round trip and the packing stay, the store (almost) never happens
#endif
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.C) + (int64_t)z * a.strideC + (int64_t)m * a.ldc + n) = o;
          }
        }
      } else {
        constexpr int LPR = TN / 4;
        constexpr int RPP = 64 / LPR;
        const int t = i * 2 + hh;
#pragma unroll
        for (int r0 = 0; r0 < 8; r0 += RPP) {
          const int r = r0 + lane / LPR, cq = (lane % LPR) * 4;
          const int m = rbase + r, n = col0 + cq;
          float4 x = *reinterpret_cast<const float4*>(patch + r * LDP + cq + pofs);
          if (RDOT || (SPEC == 0 && !PROD && a.rowdot)) {    // row-dot epilogue (wave-uniform): v * (2 r + v) summed over this wave's 64 columns
            const float4 rr = rbuf[t % RD][r0 / RPP];
            float d = x.x * (2.f * rr.x + x.x) + x.y * (2.f * rr.y + x.y) + x.z * (2.f * rr.z + x.z) + x.w * (2.f * rr.w + x.w);
            if constexpr (LPR == 16) d = sum16_dpp(d);
            else {
#pragma unroll
              for (int o = 1; o < LPR; o <<= 1) d += __shfl_xor(d, o, 64);
            }
            if ((lane % LPR) == 0 && m < a.M && col0 < a.N) a.rowdot[(int64_t)m * a.rowdot_ld + (col0 >> 6)] = d;   // col0 >= N: a wave past the last 64-column slice (N % 256 != 0) owns no slot
            continue;
          }
          if (pipe_res) {
            const float4 rr = rbuf[t % RD][r0 / RPP];
            x.x += rr.x; x.y += rr.y; x.z += rr.z; x.w += rr.w;
          }
          if (r < 8 && m < a.M && n < a.N)
#if SG_PS_ABL == 5
            if (x.x == 1234.5678f)
#endif
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + (int64_t)z * a.strideC + (int64_t)m * a.ldc + n) = x;
          if constexpr (PROD) *reinterpret_cast<float4*>(patch + r * LDP + cq) = x;   // the finished values go back into the patch (same lane, same place)
        }
        if (pipe_res && t + RD < 2 * MI) fetch_res(t + RD, rbuf[t % RD]);
        if constexpr (PROD) {
          // folded LayerNorm, producer side: the strip's finished rows are read back in the 2-byte form's layout (8 lanes x 8 columns per row,
          // 8 rows per instruction): one 16-byte store per lane for the next GEMM's operand, and the slice statistics from 3 + 3 shuffles
          static_assert(TN == 64, "row statistics are kept per 64-column slice");
          const int r8 = lane >> 3, c8 = (lane & 7) * 8;
          const int m8 = rbase + r8, n8 = col0 + c8;
          const float4 y0 = *reinterpret_cast<const float4*>(patch + r8 * LDP + c8);
          const float4 y1 = *reinterpret_cast<const float4*>(patch + r8 * LDP + c8 + 4);
          if (m8 < a.M && n8 < a.N) {
            if constexpr (SPLIT) {
              const float v8[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#if SG_H2_PROD_ABL == 1 || SG_H2_PROD_ABL == 3
              if (y0.x == 1234.5678f)
#endif
              store_h2x8(reinterpret_cast<h2_t*>(a.copy16) + (int64_t)m8 * a.ld16 + n8, v8);
            } else {
              uint4 o; o.x = pack_half2<F16>(y0.x, y0.y); o.y = pack_half2<F16>(y0.z, y0.w); o.z = pack_half2<F16>(y1.x, y1.y); o.w = pack_half2<F16>(y1.z, y1.w);
              *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.copy16) + (int64_t)m8 * a.ld16 + n8) = o;
            }
          }
#if SG_H2_PROD_ABL == 2 || SG_H2_PROD_ABL == 3
          if constexpr (SPLIT) continue;
#endif
          float sm = ((y0.x + y0.y) + (y0.z + y0.w)) + ((y1.x + y1.y) + (y1.z + y1.w));
          sm = sum8_dpp(sm);
          const float mu = sm * (1.0f / 64.0f);
          const float e0 = y0.x - mu, e1 = y0.y - mu, e2 = y0.z - mu, e3 = y0.w - mu, e4 = y1.x - mu, e5 = y1.y - mu, e6 = y1.z - mu, e7 = y1.w - mu;
          float sq = ((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3)) + ((e4 * e4 + e5 * e5) + (e6 * e6 + e7 * e7));
          sq = sum8_dpp(sq);
          // the 8 row leaders park (sum, centred squares) of rows t * 8 + r8 of this wave's 16 * MI rows in LDS; written out once, below
          if ((lane & 7) == 0) *reinterpret_cast<float2*>(statbuf + 2 * (t * 8 + r8)) = make_float2(sm, sq);
        }
      }
    }
  }
  if constexpr (PROD) {
    // slice statistics, SLICE-MAJOR [N / 64][M][2]: the wave's 16 * MI rows of its one slice are contiguous, so they leave as full lines
    // (round 3; as [M][N / 64][2] every row leader wrote 8 bytes into a line of its own -- 2.6 M partial-line writes per launch at the
    // bench shape, which cost the two-plane producer 0.2 ms per launch)
    if (col0 < a.N) {                                        // col0 >= N (N % 256 != 0): this wave's slice does not exist
#pragma unroll
      for (int rr = lane; rr < 16 * MI; rr += 64) {
        const int m = row0 + rr;
        if (m < a.M) *reinterpret_cast<float2*>(a.row_stats + ((int64_t)(col0 >> 6) * a.M + m) * 2) = *reinterpret_cast<const float2*>(statbuf + 2 * rr);
      }
    }
  }
}

// EPI: 0 = the plain epilogue, 1 / 2 = the consumer / producer side of a folded LayerNorm (epilogue_store8's MODE): separate kernels, so
// that the plain one keeps exactly the register allocation it was tuned with.
template <bool F16, int EPI = 0, int SPEC = 0>
__global__ __launch_bounds__(512) void gemm_bf16_persist(GemmBf16Args a, int act, int c_bf16) {
  constexpr int PBM = 256, PBN = 256, KT32 = 32;
  constexpr int TILE_B = (PBM + PBN) * KT32 * 2;             // 32 KiB per ring slot
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wi = wave & 3;
  const int tiles_n = (a.N + PBN - 1) / PBN, tiles_m = (a.M + PBM - 1) / PBM;
  const int nwg = tiles_m * tiles_n;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int z = blockIdx.y;
  const bf16_t* A = a.A + (int64_t)z * a.strideA;
  const bf16_t* W = a.W + (int64_t)z * a.strideW;
  const int nt = a.K / KT32;
  // tile order (a.ngroup > 0): XCD x = blockIdx.x & 7 owns M tiles [mlo, mlo + mcnt) and walks its mcnt * tiles_n tiles N-group by N-group
  const int NG = a.ngroup;
  const int wpx = (int)gridDim.x >> 3, wx = (int)blockIdx.x >> 3, xc = (int)blockIdx.x & 7;
  const int mq8 = tiles_m >> 3, mr8 = tiles_m & 7;
  const int mlo = xc * mq8 + (xc < mr8 ? xc : mr8), mcnt = mq8 + (xc < mr8 ? 1 : 0), xcnt = mcnt * tiles_n;
  const int my_tiles = NG > 0 ? (xcnt - wx + wpx - 1) / wpx : (nwg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nt;                          // length of this workgroup's K-tile stream

  const int srow = lane >> 2, cpos = lane & 3;
  struct Src { const bf16_t* a[2]; const bf16_t* w[2]; int m0, n0; };
  auto make_src = [&](int j) {
    Src sp;
    if (NG > 0) {
      int Lx = wx + j * wpx;                                   // index inside this XCD's tile set
      Lx = Lx < xcnt ? Lx : xcnt - 1;
      const int gsz = mcnt * NG, ngroups = (tiles_n + NG - 1) / NG;
      int grp = Lx / gsz; grp = grp < ngroups - 1 ? grp : ngroups - 1;
      const int rem = Lx - grp * gsz;
      const int ncols = grp == ngroups - 1 ? tiles_n - grp * NG : NG;
      sp.m0 = (mlo + rem / ncols) * PBM; sp.n0 = (grp * NG + rem % ncols) * PBN;
    } else {
      const int v = (int)blockIdx.x + j * (int)gridDim.x;    // virtual id; XCD x = v & 7 walks a contiguous chunk of tile ids
      const int xcd = v & 7, seq = v >> 3;
      int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + seq;
      tile = tile < nwg ? tile : nwg - 1;
      sp.m0 = (tile / tiles_n) * PBM; sp.n0 = (tile % tiles_n) * PBN;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int ra = 128 * g + 32 * wi + 16 * p + srow;
      const int rw = 64 * wi + 32 * g + 16 * p + srow;
      int gra = sp.m0 + ra; gra = gra < a.M ? gra : a.M - 1;
      int grw = sp.n0 + rw; grw = grw < a.N ? grw : a.N - 1;
      sp.a[p] = A + (int64_t)gra * a.lda + ((cpos ^ swz32(ra)) << 3);
      sp.w[p] = W + (int64_t)grw * a.ldw + ((cpos ^ swz32(rw)) << 3);
    }
    return sp;
  };
  int a_dst[2], w_dst[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    a_dst[p] = (128 * g + 32 * wi + 16 * p) * 64;
    w_dst[p] = PBM * KT32 * 2 + (64 * wi + 32 * g + 16 * p) * 64;
  }
  Src cur = make_src(0);
  Src nxt = make_src(my_tiles > 1 ? 1 : 0);
  int cur_end = nt;                                          // stream index where `nxt` begins
  auto load_a = [&](int u) {
    if (u >= total) return;
    const Src& sp = u >= cur_end ? nxt : cur;
    const int kt = u >= cur_end ? u - cur_end : u - (cur_end - nt);
    char* base = lds + (u & 3) * TILE_B;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(sp.a[p] + kt * KT32), (lds_ptr_t)(base + a_dst[p]), 16, 0, 0);
  };
  auto load_w = [&](int u) {
    if (u >= total) return;
    const Src& sp = u >= cur_end ? nxt : cur;
    const int kt = u >= cur_end ? u - cur_end : u - (cur_end - nt);
    char* base = lds + (u & 3) * TILE_B;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(sp.w[p] + kt * KT32), (lds_ptr_t)(base + w_dst[p]), 16, 0, 0);
  };
  // retire this wave's pieces of K tile u, leaving younger pieces in flight
  auto wait_tile = [&](int u) {
    if (g == 0) { if (u + 2 < total) wait_vmcnt<6>(); else if (u + 1 < total) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
    else { if (u + 2 < total) wait_vmcnt<8>(); else if (u + 1 < total) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
  };
#define SG_PS_SYNC()                                 \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_barrier();                    \
    __builtin_amdgcn_sched_barrier(0);               \
  } while (0)

  // prologue (group 0 holds W(2) back: it is issued in READ(0) as W(s+2))
  load_a(0); load_w(0); load_a(1); load_w(1); load_a(2);
  if (g == 1) load_w(2);
  wait_tile(0);
  SG_PS_SYNC();
  if (g == 1) SG_PS_SYNC();

  f32x4 acc[8][4];
  bf16x8 fa[8], fw[4];
  int s = 0;
  for (int j = 0; j < my_tiles; ++j) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // One K step.  STEADY: every piece this step requests (K tiles s+2 / s+3) still belongs to the CURRENT output tile and the stream is
    // far from its end -- no source select, no end-of-stream test, constant wait counts; the last three steps of a tile take the general form.
    int kt = 0;
    auto kstep = [&](auto steady_tag) {
      constexpr bool STEADY = decltype(steady_tag)::value;
      const char* tA = lds + (s & 3) * TILE_B;
      const char* tW = tA + PBM * KT32 * 2;
      // READ(s)   (SG_PS_ABL, tuning builds only -- WRONG results by design: 1 = no LDS-DMA issue in the steady loop, 2 = no fragment reads,
      //            3 = no MFMAs, 4 = no epilogue: what each part of a slot costs, tools/ablate_persist.sh)
#if SG_PS_ABL == 2
      if (!STEADY || kt == 0) {
#endif
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) fw[jj] = read_frag32(tW, 64 * wi + 16 * jj + (lane & 15), lane >> 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = read_frag32(tA, 128 * g + 16 * i + (lane & 15), lane >> 4);
#if SG_PS_ABL == 2
      }
#endif
      if constexpr (STEADY && SG_PS_ABL == 1) {
        if (g == 1) wait_vmcnt<8>();
      } else if constexpr (STEADY) {
        auto piece = [&](const bf16_t* const (&src)[2], const int (&dst)[2], int ahead) {
          char* base = lds + ((s + ahead) & 3) * TILE_B;
#pragma unroll
          for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[p] + (kt + ahead) * KT32), (lds_ptr_t)(base + dst[p]), 16, 0, 0);
        };
        if (g == 0) { piece(cur.w, w_dst, 2); piece(cur.a, a_dst, 3); }
        else { piece(cur.a, a_dst, 3); piece(cur.w, w_dst, 3); wait_vmcnt<8>(); }
      } else {
        if (g == 0) { load_w(s + 2); load_a(s + 3); }
        else { load_a(s + 3); load_w(s + 3); wait_tile(s + 1); }
      }
      SG_PS_SYNC();
      // MFMA(s)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#if SG_PS_ABL == 3
          if (STEADY) { asm volatile("" ::"v"(fw[jj]), "v"(fa[i])); continue; }
#endif
          acc[i][jj] = mfma_16x16x32<F16>(fw[jj], fa[i], acc[i][jj]);
        }
      __builtin_amdgcn_s_setprio(0);
      if (g == 0) { if constexpr (STEADY) wait_vmcnt<6>(); else wait_tile(s + 1); }
      SG_PS_SYNC();
      ++kt; ++s;
    };
    // (the producer instantiation, EPI 2, keeps the general form throughout: measured 1.5 % slower with the split, the others 2 % faster)
    if constexpr (EPI != 2) { for (; kt + 3 < nt; ) kstep(std::true_type{}); }
    for (; kt < nt; ) kstep(std::false_type{});
    // ---- tile end ----
    if (g == 0) SG_PS_SYNC();                               // align: every read of this tile's last K tile has retired
    {
      float* patch = reinterpret_cast<float*>(lds + ((s - 1) & 3) * TILE_B) + wave * 576;   // 8 rows x 68 floats (+pad) per wave
#if SG_PS_ABL == 4
      if (acc[0][0][0] == 1234.5678f)                                                       // keep the accumulators live, store (almost) nothing
#endif
      epilogue_store8<8, 4, F16, EPI, SPEC>(acc, a, act, c_bf16, z, cur.m0 + 128 * g, cur.n0 + 64 * wi, patch, lane,
                                            reinterpret_cast<float*>(lds + ((s - 1) & 3) * TILE_B) + 8 * 576 + wave * 256);   // 1 KiB per wave behind the 8 patches
    }
    cur = nxt; cur_end += nt;
    if (j + 2 < my_tiles) nxt = make_src(j + 2);
    SG_PS_SYNC();                                           // the ring slot used as patch may be refilled from here on
    if (g == 1 && j + 1 < my_tiles) SG_PS_SYNC();           // re-stagger
  }
#undef SG_PS_SYNC
}



// Epilogue of the persistent fp8 kernel.  Same patch transposition as epilogue_store8, but de-quantisation (acc * row_scale[m] *
// col_scale[n]), bias and activation are applied AFTER it, where a lane owns the same 4 (f32) / 8 (bf16) columns for every strip:
// one column-scale / bias quad per lane instead of one per 16-column block, and the wave's 128 row scales sit in LDS.
template <int MI, int NI>
__device__ __forceinline__ void epilogue_store8_fp8(f32x4 (&acc)[MI][NI], const GemmBf16Args& a, int act, int c_bf16, int row0, int col0,
                                                    float* patch, float* srs, int lane, const float (&rs_pre)[MI * 16 / 64]) {
  constexpr int TN = NI * 16, LDP = TN + 4;
  const float* res = a.residual;
#pragma unroll
  for (int u = 0; u < MI * 16 / 64; ++u) srs[u * 64 + lane] = rs_pre[u] * a.alpha;   // the wave's MI*16 row scales (fetched during the last K step) -> LDS
  constexpr int RD = 4;
  constexpr int LPRF = TN / 4, RPPF = 64 / LPRF, NPASS = 8 / RPPF;
  float4 rbuf[RD][NPASS];
  auto fetch_res = [&](int t, float4 (&dst)[NPASS]) {
    const int rb = row0 + (t >> 1) * 16 + (t & 1) * 8;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      int m = rb + ps * RPPF + lane / LPRF; m = m < a.M ? m : a.M - 1;
      int n = col0 + (lane % LPRF) * 4; n = n < a.N ? n : a.N - 4;
      dst[ps] = *reinterpret_cast<const float4*>(res + (int64_t)m * a.ldr + n);
    }
  };
  const bool pipe_res = res != nullptr && !c_bf16;
  if (pipe_res) {
#pragma unroll
    for (int t = 0; t < RD; ++t) fetch_res(t, rbuf[t]);
  }
  auto finish = [&](float4 x, float rs, float4 cs, float4 bi) {
    float v[4] = {x.x * (rs * cs.x) + bi.x, x.y * (rs * cs.y) + bi.y, x.z * (rs * cs.z) + bi.z, x.w * (rs * cs.w) + bi.w};
    if (act == ACT_QUICK_GELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
    } else if (act == ACT_GELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = erf_gelu_fast(v[e]);
    }
    return make_float4(v[0], v[1], v[2], v[3]);
  };
  auto quad = [&](const float* p, int n) { return (p && n + 3 < a.N) ? *reinterpret_cast<const float4*>(p + n) : make_float4(0.f, 0.f, 0.f, 0.f); };
  // this lane's columns after the transposition
  const int cq16 = (lane % (TN / 8)) * 8, cq32 = (lane % LPRF) * 4;
  float4 cs0, cs1, bi0, bi1;
  if (c_bf16) {
    cs0 = quad(a.col_scale, col0 + cq16); cs1 = quad(a.col_scale, col0 + cq16 + 4); bi0 = quad(a.bias, col0 + cq16); bi1 = quad(a.bias, col0 + cq16 + 4);
  } else {
    cs0 = quad(a.col_scale, col0 + cq32); bi0 = quad(a.bias, col0 + cq32); cs1 = cs0; bi1 = bi0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {                       // two 8-row strips per 16-row MFMA tile
      if (((lane >> 3) & 1) == hh) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
          *reinterpret_cast<float4*>(patch + (lane & 7) * LDP + j * 16 + (lane >> 4) * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
      const int rloc = i * 16 + hh * 8;
      const int rbase = row0 + rloc;
      if (c_bf16) {                                        // 8 lanes x 16 B per row: one instruction stores the whole strip
        const int r = lane / (TN / 8);
        const int m = rbase + r, n = col0 + cq16;
        const float rs = srs[rloc + r];
        const float4 x0 = finish(*reinterpret_cast<const float4*>(patch + r * LDP + cq16), rs, cs0, bi0);
        const float4 x1 = finish(*reinterpret_cast<const float4*>(patch + r * LDP + cq16 + 4), rs, cs1, bi1);
        if (m < a.M && n < a.N) {
          uint4 o; o.x = pack_bf2(x0.x, x0.y); o.y = pack_bf2(x0.z, x0.w); o.z = pack_bf2(x1.x, x1.y); o.w = pack_bf2(x1.z, x1.w);
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.C) + (int64_t)m * a.ldc + n) = o;
        }
      } else {
        const int t = i * 2 + hh;
#pragma unroll
        for (int r0 = 0; r0 < 8; r0 += RPPF) {
          const int r = r0 + lane / LPRF;
          const int m = rbase + r, n = col0 + cq32;
          float4 x = finish(*reinterpret_cast<const float4*>(patch + r * LDP + cq32), srs[rloc + r], cs0, bi0);
          if (pipe_res) {
            const float4 rr = rbuf[t % RD][r0 / RPPF];
            x.x += rr.x; x.y += rr.y; x.z += rr.z; x.w += rr.w;
          }
          if (m < a.M && n < a.N) *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + (int64_t)m * a.ldc + n) = x;
        }
        if (pipe_res && t + RD < 2 * MI) fetch_res(t + RD, rbuf[t % RD]);
      }
    }
  }
}

// ---- persistent fp8 ping-pong (SG_PREC_FP8: the QKV / fc / proj linears of the ordinary blocks) ------------------------------------
// The bf16 persistent kernel's structure carried to OCP e4m3 operands: 256 x 256 output tile, 8 waves in two groups that alternate READ /
// MFMA segments, persistent workgroups whose K-step stream runs across output tiles, epilogue through a per-wave LDS patch.
// A K step is 128 fp8 per row = ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 output (32 cycles: twice the bf16 form at 4x the K).
// An operand fragment is 32 B per lane, so the 12 fragments of a wave (8 A + 4 W) no longer fit next to 128 accumulators: a K step is
// TWO phases (rows 0-63 / 64-127 of the group's half), the 4 W fragments stay in registers across both.
//   slots (barrier intervals), group 0: READ0(s) = 4s, MFMA0(s) = 4s+1, READ1(s) = 4s+2, MFMA1(s) = 4s+3; group 1 one later.
// LDS, all 160 KiB: the A ring has 2 slots, the W ring 3 (32 KiB each: 256 rows x 128 B, 16-byte chunks XOR-swizzled as in the 128-byte
// bf16 image).  A-half g is only ever read by group g, so two slots suffice; W is read by both groups, which costs it one more slot:
//   loads : READ0(t): group g issues A_g(t+1)                    (position (t+1)&1, last read in READ1(t-1), retired at slot 4t-1+g)
//           READ1(t): group 0 issues its rows of W(t+2), group 1 its rows of W(t+3)
//             (W(t) is read in READ0(t) = slots 4t / 4t+1 and retired when slot 4t+2 opens: group 1 (slot 4t+3) may overwrite it, group 0
//              (slot 4t+2) may not yet and refills the position of W(t-1) instead)
//   RAW   : every wave ends MFMA1(t) with vmcnt(4): everything but the W pieces it issued in READ1(t) has landed -- its A(t+1) pieces and
//           its pieces of W(t+1) (issued one or two steps earlier) -- before the barrier that precedes READ0(t+1).
//   tile end: as in the bf16 kernel (group 0 takes one extra barrier, epilogue through a patch inside the just-consumed A slot, barrier,
//           group 1 re-staggers).  Look-ahead is at most 3 K steps, so K >= 512 (4 steps per tile) keeps loads within the next tile.
template <bool DUMMY = false>
__global__ __launch_bounds__(512) void gemm_fp8_persist(GemmBf16Args a, int act, int c_bf16) {
  constexpr int PBM = 256, PBN = 256, KB = 128;                          // K step in bytes (= fp8 elements)
  constexpr int SLOT = 256 * KB;                                         // 32 KiB
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* ldsA = lds;                                                      // 2 slots
  char* ldsW = lds + 2 * SLOT;                                           // 3 slots
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wi = wave & 3;
  const int tiles_n = (a.N + PBN - 1) / PBN, tiles_m = (a.M + PBM - 1) / PBM;
  const int nwg = tiles_m * tiles_n;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const uint8_t* A = reinterpret_cast<const uint8_t*>(a.A);
  const uint8_t* W = reinterpret_cast<const uint8_t*>(a.W);
  const int nt = a.K / KB;
  const int my_tiles = (nwg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nt;                                       // length of this workgroup's K-step stream

  const int srow = lane >> 3, cpos = lane & 7;
  // Per tile a lane only keeps the first row of its 4 A pieces and of its 4 W pieces (pieces are 8 rows apart); the byte offset of a piece is
  // rebuilt at issue time (row clamp + one 32-bit multiply: operands < 2 GiB, checked on the host), the swizzle term depends on the lane and on
  // the parity of the piece only.  (Eight 64-bit pointers per tile for the current and the next tile do not fit next to 128 accumulators and
  // 64 fragment registers.)
  struct Src { int ra0, rw0, m0, n0; };
  const int swz_e = (cpos ^ (srow >> 1)) << 4, swz_o = (cpos ^ (4 + (srow >> 1))) << 4;
  auto make_src = [&](int j) {
    Src sp;
    const int v = (int)blockIdx.x + j * (int)gridDim.x;                  // XCD x = v & 7 walks a contiguous chunk of tile ids
    const int xcd = v & 7, seq = v >> 3;
    int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + seq;
    tile = tile < nwg ? tile : nwg - 1;
    sp.m0 = (tile / tiles_n) * PBM; sp.n0 = (tile % tiles_n) * PBN;
    sp.ra0 = sp.m0 + 128 * g + 32 * wi + srow;
    sp.rw0 = sp.n0 + 64 * wi + 32 * g + srow;
    return sp;
  };
  Src cur = make_src(0);
  Src nxt = make_src(my_tiles > 1 ? 1 : 0);
  int cur_end = nt;                                                      // stream index where `nxt` begins
  const int lda = (int)a.lda, ldw = (int)a.ldw, Mm1 = a.M - 1, Nm1 = a.N - 1;
  auto load_a = [&](int u) {                                             // this wave's pieces of A_g(u)
    if (u >= total) return;
    const bool nx = u >= cur_end;
    const int r0 = nx ? nxt.ra0 : cur.ra0;
    const int kt = nx ? u - cur_end : u - (cur_end - nt);
    char* base = ldsA + (u & 1) * SLOT + (128 * g + 32 * wi) * KB;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int r = r0 + 8 * p; r = r < Mm1 ? r : Mm1;
      const int off = r * lda + ((p & 1) ? swz_o : swz_e) + kt * KB;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(A + off), (lds_ptr_t)(base + p * 8 * KB), 16, 0, 0);
    }
  };
  auto load_w = [&](int u) {                                             // this wave's rows of W(u)
    if (u >= total) return;
    const bool nx = u >= cur_end;
    const int r0 = nx ? nxt.rw0 : cur.rw0;
    const int kt = nx ? u - cur_end : u - (cur_end - nt);
    char* base = ldsW + (u % 3) * SLOT + (64 * wi + 32 * g) * KB;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int r = r0 + 8 * p; r = r < Nm1 ? r : Nm1;
      const int off = r * ldw + ((p & 1) ? swz_o : swz_e) + kt * KB;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(W + off), (lds_ptr_t)(base + p * 8 * KB), 16, 0, 0);
    }
  };
#define SG_F8_SYNC()                                 \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_barrier();                    \
    __builtin_amdgcn_sched_barrier(0);               \
  } while (0)

  // prologue: A(0), W(0), W(1) from everybody, W(2) from group 1 (group 0 issues its rows of W(2) in READ1(0))
  load_a(0); load_w(0); load_w(1);
  if (g == 1) { load_w(2); if (2 < total) wait_vmcnt<8>(); else if (1 < total) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
  else { if (1 < total) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
  SG_F8_SYNC();
  if (g == 1) SG_F8_SYNC();

  typedef __attribute__((ext_vector_type(8))) int i32x8;
  union Op { bf16x8 h[2]; i32x8 v; };
  f32x4 acc[8][4];
  Op fa8[4], fw8[4];
  float rs_pre[2] = {1.f, 1.f};
  int s = 0;
  for (int j = 0; j < my_tiles; ++j) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nt; ++kt, ++s) {
      const char* tA = ldsA + (s & 1) * SLOT;
      const char* tW = ldsW + (s % 3) * SLOT;
      // READ0(s): W fragments (kept for both phases) + A rows 0-63 of the group's half; issue A_g(s+1)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) fw8[jj].h[kk] = read_frag(tW, 64 * wi + 16 * jj + (lane & 15), kk * 4 + (lane >> 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) fa8[i].h[kk] = read_frag(tA, 128 * g + 16 * i + (lane & 15), kk * 4 + (lane >> 4));
      }
      load_a(s + 1);
      SG_F8_SYNC();
      // MFMA0(s)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)   // cbsz = blgp = 0: both operands e4m3; scales 0x7f = 2^0 (E8M0): plain fp8 x fp8
          acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw8[jj].v, fa8[i].v, acc[i][jj], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      __builtin_amdgcn_s_setprio(0);
      SG_F8_SYNC();
      // READ1(s): A rows 64-127; issue this group's rows of W(s+2) / W(s+3)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) fa8[i].h[kk] = read_frag(tA, 128 * g + 64 + 16 * i + (lane & 15), kk * 4 + (lane >> 4));
      const int uw = s + 2 + g;
      load_w(uw);
      if (kt == nt - 1) {                                                  // last K step of the tile: this wave's 128 row scales for the epilogue
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) { int m = cur.m0 + 128 * g + u2 * 64 + lane; m = m < a.M ? m : a.M - 1; rs_pre[u2] = a.row_scale[m]; }
      }
      SG_F8_SYNC();
      // MFMA1(s)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          acc[4 + i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw8[jj].v, fa8[i].v, acc[4 + i][jj], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      __builtin_amdgcn_s_setprio(0);
      if (kt == nt - 1) wait_vmcnt<0>();                                 // tile end: the row scales too (they were issued after the W pieces)
      else if (uw < total) wait_vmcnt<4>(); else wait_vmcnt<0>();        // all but the W pieces just issued
      SG_F8_SYNC();
    }
    // ---- tile end ----
    if (g == 0) SG_F8_SYNC();                                            // align: every read of this tile's last K step has retired
    {
      float* pbase = reinterpret_cast<float*>(ldsA + ((s - 1) & 1) * SLOT);    // the consumed A slot: 8 patches of 8 rows x 68 floats (+pad), then 8 x 128 row scales
      epilogue_store8_fp8<8, 4>(acc, a, act, c_bf16, cur.m0 + 128 * g, cur.n0 + 64 * wi, pbase + wave * 576, pbase + 8 * 576 + wave * 128, lane, rs_pre);
    }
    cur = nxt; cur_end += nt;
    if (j + 2 < my_tiles) nxt = make_src(j + 2);
    SG_F8_SYNC();                                                        // the A slot used as patch is refilled from READ0 of the next step on
    if (g == 1 && j + 1 < my_tiles) SG_F8_SYNC();                        // re-stagger
  }
#undef SG_F8_SYNC
}

static int launch_fp8_persist(const GemmBf16Args& a, hipStream_t s) {
  const size_t lds = 5 * 256 * 128;                                      // 160 KiB: the whole LDS of a CU
  auto kern = gemm_fp8_persist<false>;
  SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int n_cu = device_cu_count();
  const int64_t tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
  SG_REQUIRE(tiles < (1ll << 31), "gemm_fp8: grid too large");
  const unsigned grid = (unsigned)(tiles < n_cu ? tiles : n_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, a.act, a.c_is_bf16);
  return SG_OK;
}


// ---- persistent two-plane f16 ping-pong (SG_PREC_F16X2: every large ViT linear of the exact mode) ------------------------------------------
// gemm_fp8_persist's structure on two-plane f16 operands: a K step is again 128 bytes per row -- 32 elements as four [8 hi | 8 lo] storage
// groups -- so the rings (A: 2 slots, W: 3 slots = all 160 KiB of LDS), the two MFMA phases per K step, the hand-off rules and the tile
// switch are those of the fp8 kernel, comment for comment.  What differs:
//   * the source-side chunk permutation also DE-INTERLEAVES the planes (as in gemm_bf16_ring<SPLIT>): LDS chunks 0-3 of a row hold the four
//     hi chunks, 4-7 the four lo chunks, so fragment h[0] (chunk g) is the hi plane and h[1] (chunk 4 + g) the lo plane of the lane's 8 K values;
//   * a 16 x 16 output takes THREE v_mfma_f32_16x16x32_f16 per K step (W_hi.A_hi + W_lo.A_hi + W_hi.A_lo): 48 MFMAs = 768 cycles per phase
//     against the same READ segments as the fp8 kernel (16 / 8 ds_read_b128 + 4 LDS-DMA pieces) -- the matrix pipe has 1.5x the cover;
//   * operand rows are 4 bytes per element (the fc output of a 128-tile launch is 2.9 GB), so a tile keeps a 64-bit base per operand and
//     32-bit offsets inside its 256 rows;
//   * the epilogue is epilogue_store8<.., SPLIT>: bias / exact activation / f32 residual / f32 or two-plane output, no scales.
// `a` arrives as gemm_h2 prepared it: K, lda, ldw in f16 UNITS (2 x the element counts), C / residual strides in elements.
// SPEC as epilogue_store8's: 0 = every choice at run time, 1 / 2 / 3 = two-plane output after no activation / QuickGELU / GELU (QKV, fc),
// 4 = f32 output + f32 residual (out-proj, proj).
template <int SPEC, int EPI = 0>
__global__ __launch_bounds__(512) void gemm_h2_persist(GemmBf16Args a, int act, int c_bf16) {
  constexpr int PBM = 256, PBN = 256, KB = 128;                          // K step in bytes
  constexpr int SLOT = 256 * KB;                                         // 32 KiB
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* ldsA = lds;                                                      // 2 slots
  char* ldsW = lds + 2 * SLOT;                                           // 3 slots
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wi = wave & 3;
  const int tiles_n = (a.N + PBN - 1) / PBN, tiles_m = (a.M + PBM - 1) / PBM;
  const int nwg = tiles_m * tiles_n;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const char* A = reinterpret_cast<const char*>(a.A);
  const char* W = reinterpret_cast<const char*>(a.W);
  const int64_t lda_b = a.lda * 2, ldw_b = a.ldw * 2;                    // row strides in bytes
  const int nt = a.K * 2 / KB;
  const int my_tiles = (nwg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nt;                                       // length of this workgroup's K-step stream

  const int srow = lane >> 3, cpos = lane & 7;
  // plane-de-interleaving swizzle: LDS chunk cpos of row r holds logical chunk L = cpos ^ ((r >> 1) & 7) = 4 * plane + group, which lives at
  // global chunk 2 * group + plane.  Pieces are 8 rows apart, so the swizzle term depends on the lane and on the parity of the piece only.
  auto gchunk = [](int L) { return ((L & 3) << 1) | (L >> 2); };
  const int swz_e = gchunk(cpos ^ (srow >> 1)) << 4, swz_o = gchunk(cpos ^ (4 + (srow >> 1))) << 4;
  struct Src { const char* Ab; const char* Wb; int m0, n0; };            // 64-bit tile bases (wave-uniform)
  const int ra0 = 128 * g + 32 * wi + srow, rw0 = 64 * wi + 32 * g + srow;   // this lane's first piece row inside any tile
  auto make_src = [&](int j) {
    Src sp;
    const int v = (int)blockIdx.x + j * (int)gridDim.x;                  // XCD x = v & 7 walks a contiguous chunk of tile ids
    const int xcd = v & 7, seq = v >> 3;
    int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + seq;
    tile = tile < nwg ? tile : nwg - 1;
    sp.m0 = (tile / tiles_n) * PBM; sp.n0 = (tile % tiles_n) * PBN;
    sp.Ab = A + (int64_t)sp.m0 * lda_b; sp.Wb = W + (int64_t)sp.n0 * ldw_b;
    return sp;
  };
  Src cur = make_src(0);
  Src nxt = make_src(my_tiles > 1 ? 1 : 0);
  int cur_end = nt;                                                      // stream index where `nxt` begins
  const int lda_i = (int)lda_b, ldw_i = (int)ldw_b;                      // < 2^23 (checked on the host): offsets inside a 256-row tile fit 32 bits
  auto load_a = [&](int u) {                                             // this wave's pieces of A_g(u)
    if (u >= total) return;
    const bool nx = u >= cur_end;
    const Src& sp = nx ? nxt : cur;
    const int kt = nx ? u - cur_end : u - (cur_end - nt);
    const int rmax = a.M - 1 - sp.m0;                                    // rows past the edge re-read the last row (never stored)
    char* base = ldsA + (u & 1) * SLOT + (128 * g + 32 * wi) * KB;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int r = ra0 + 8 * p; r = r < rmax ? r : rmax;
      const int off = r * lda_i + ((p & 1) ? swz_o : swz_e) + kt * KB;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(sp.Ab + off), (lds_ptr_t)(base + p * 8 * KB), 16, 0, 0);
    }
  };
  auto load_w = [&](int u) {                                             // this wave's rows of W(u)
    if (u >= total) return;
    const bool nx = u >= cur_end;
    const Src& sp = nx ? nxt : cur;
    const int kt = nx ? u - cur_end : u - (cur_end - nt);
    const int rmax = a.N - 1 - sp.n0;
    char* base = ldsW + (u % 3) * SLOT + (64 * wi + 32 * g) * KB;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int r = rw0 + 8 * p; r = r < rmax ? r : rmax;
      const int off = r * ldw_i + ((p & 1) ? swz_o : swz_e) + kt * KB;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(sp.Wb + off), (lds_ptr_t)(base + p * 8 * KB), 16, 0, 0);
    }
  };
#define SG_H2_SYNC()                                 \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_barrier();                    \
    __builtin_amdgcn_sched_barrier(0);               \
  } while (0)

  // prologue: A(0), W(0), W(1) from everybody, W(2) from group 1 (group 0 issues its rows of W(2) in READ1(0))
  load_a(0); load_w(0); load_w(1);
  if (g == 1) { load_w(2); if (2 < total) wait_vmcnt<8>(); else if (1 < total) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
  else { if (1 < total) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
  SG_H2_SYNC();
  if (g == 1) SG_H2_SYNC();

  f32x4 acc[8][4];
  bf16x8 fah[4], fal[4], fwh[4], fwl[4];                                 // hi / lo planes of the lane's 8 K values
  int s = 0;
  for (int j = 0; j < my_tiles; ++j) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nt; ++kt, ++s) {
      const char* tA = ldsA + (s & 1) * SLOT;
      const char* tW = ldsW + (s % 3) * SLOT;
      // READ0(s): W fragments (kept for both phases) + A rows 0-63 of the group's half; issue A_g(s+1)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        fwh[jj] = read_frag(tW, 64 * wi + 16 * jj + (lane & 15), lane >> 4);
        fwl[jj] = read_frag(tW, 64 * wi + 16 * jj + (lane & 15), 4 + (lane >> 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fah[i] = read_frag(tA, 128 * g + 16 * i + (lane & 15), lane >> 4);
        fal[i] = read_frag(tA, 128 * g + 16 * i + (lane & 15), 4 + (lane >> 4));
      }
      load_a(s + 1);
      SG_H2_SYNC();
      // MFMA0(s)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          acc[i][jj] = mfma_16x16x32<true>(fwh[jj], fah[i], acc[i][jj]);
          acc[i][jj] = mfma_16x16x32<true>(fwl[jj], fah[i], acc[i][jj]);
          acc[i][jj] = mfma_16x16x32<true>(fwh[jj], fal[i], acc[i][jj]);
        }
      __builtin_amdgcn_s_setprio(0);
      SG_H2_SYNC();
      // READ1(s): A rows 64-127; issue this group's rows of W(s+2) / W(s+3)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fah[i] = read_frag(tA, 128 * g + 64 + 16 * i + (lane & 15), lane >> 4);
        fal[i] = read_frag(tA, 128 * g + 64 + 16 * i + (lane & 15), 4 + (lane >> 4));
      }
      const int uw = s + 2 + g;
      load_w(uw);
      SG_H2_SYNC();
      // MFMA1(s)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          acc[4 + i][jj] = mfma_16x16x32<true>(fwh[jj], fah[i], acc[4 + i][jj]);
          acc[4 + i][jj] = mfma_16x16x32<true>(fwl[jj], fah[i], acc[4 + i][jj]);
          acc[4 + i][jj] = mfma_16x16x32<true>(fwh[jj], fal[i], acc[4 + i][jj]);
        }
      __builtin_amdgcn_s_setprio(0);
      if (uw < total) wait_vmcnt<4>(); else wait_vmcnt<0>();             // all but the W pieces just issued
      SG_H2_SYNC();
    }
    // ---- tile end ----
    if (g == 0) SG_H2_SYNC();                                            // align: every read of this tile's last K step has retired
    {
      float* pbase = reinterpret_cast<float*>(ldsA + ((s - 1) & 1) * SLOT);    // the consumed A slot: 8 patches of 8 rows x 68 floats (+pad)
      epilogue_store8<8, 4, true, EPI, SPEC, true>(acc, a, act, c_bf16, 0, cur.m0 + 128 * g, cur.n0 + 64 * wi, pbase + wave * 576, lane, pbase + 8 * 576 + wave * 256);
    }
    cur = nxt; cur_end += nt;
    if (j + 2 < my_tiles) nxt = make_src(j + 2);
    SG_H2_SYNC();                                                        // the A slot used as patch is refilled from READ0 of the next step on
    if (g == 1 && j + 1 < my_tiles) SG_H2_SYNC();                        // re-stagger
  }
#undef SG_H2_SYNC
}

static int launch_h2_persist(const GemmBf16Args& h, hipStream_t s) {
  const size_t lds = 5 * 256 * 128;                                      // 160 KiB: the whole LDS of a CU
  using Kern = void (*)(GemmBf16Args, int, int);
  // epilogue form: folded-LayerNorm consumer (two-plane output, activation per layer kind) / producer (f32 + residual, two-plane copy, slice
  // statistics) / plain
  Kern kern;
  if (h.ln_stats) kern = h.act == ACT_NONE ? gemm_h2_persist<1, 1> : h.act == ACT_QUICK_GELU ? gemm_h2_persist<2, 1> : gemm_h2_persist<3, 1>;
  else if (h.copy16) kern = gemm_h2_persist<1, 2>;
  else kern = (h.c_is_bf16 && !h.residual) ? (h.act == ACT_NONE ? gemm_h2_persist<1> : h.act == ACT_QUICK_GELU ? gemm_h2_persist<2> : gemm_h2_persist<3>)
            : (!h.c_is_bf16 && h.residual && h.act == ACT_NONE) ? gemm_h2_persist<4> : gemm_h2_persist<0>;
  SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int n_cu = device_cu_count();
  const int64_t tiles = cdiv(h.M, 256) * cdiv(h.N, 256);
  SG_REQUIRE(tiles < (1ll << 31), "gemm_h2: grid too large");
  const unsigned grid = (unsigned)(tiles < n_cu ? tiles : n_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, h, h.act, h.c_is_bf16);
  return SG_OK;
}

static int launch_persist(const GemmBf16Args& a, hipStream_t s) {
  const size_t lds = 4 * (256 + 256) * 32 * 2;
  // instantiation: epilogue form (plain / folded-LayerNorm consumer / producer) x compile-time specialisation of the hot combinations
  using Kern = void (*)(GemmBf16Args, int, int);
  Kern kern;
  if (a.ln_stats) {                                         // consumer: 2-byte output, activation per layer kind
    kern = a.act == ACT_NONE ? (a.f16 ? gemm_bf16_persist<true, 1, 1> : gemm_bf16_persist<false, 1, 1>)
         : a.act == ACT_QUICK_GELU ? (a.f16 ? gemm_bf16_persist<true, 1, 2> : gemm_bf16_persist<false, 1, 2>)
                                   : (a.f16 ? gemm_bf16_persist<true, 1, 3> : gemm_bf16_persist<false, 1, 3>);
  } else if (a.copy16) {                                    // producer: f32 output + residual, no activation
    SG_REQUIRE(a.residual && a.act == ACT_NONE, "gemm_bf16: the folded-LayerNorm producer form is a residual GEMM without activation");
    kern = a.f16 ? gemm_bf16_persist<true, 2, 1> : gemm_bf16_persist<false, 2, 1>;
  } else if (a.c_is_bf16 && !a.residual && !a.rowdot && a.act == ACT_NONE) {
    kern = a.f16 ? gemm_bf16_persist<true, 0, 1> : gemm_bf16_persist<false, 0, 1>;
  } else if (!a.c_is_bf16 && a.residual && !a.rowdot && a.act == ACT_NONE) {
    kern = a.f16 ? gemm_bf16_persist<true, 0, 4> : gemm_bf16_persist<false, 0, 4>;
  } else if (a.rowdot && a.rowdot_res_bf16 && !a.f16 && a.act == ACT_NONE) {
    kern = gemm_bf16_persist<false, 0, 5>;
  } else {
    kern = a.f16 ? gemm_bf16_persist<true, 0, 0> : gemm_bf16_persist<false, 0, 0>;
  }
  SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int n_cu = device_cu_count();
  const int64_t tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
  SG_REQUIRE(tiles < (1ll << 31) && a.batch < 65536, "gemm_bf16: grid too large");
  unsigned grid = (unsigned)(tiles < n_cu ? tiles : n_cu);
  if (g_persist_grid_cap > 0 && grid > (unsigned)g_persist_grid_cap) grid = (unsigned)g_persist_grid_cap;
  GemmBf16Args b = a;
  // W-panel-resident order where it pays: more N tiles than fit the L2 together and a short K (re-reading the A panels once per
  // N-group must cost less than re-streaming every W panel once per round): the K = 1024 linears with N = 3072 / 4096
  const int64_t tiles_m = cdiv(a.M, 256), tiles_n = cdiv(a.N, 256);
  const int64_t panel = (int64_t)256 * a.K * 2;
  b.ngroup = 0;
  if (g_gemm_order != 0 && a.batch == 1 && grid == (unsigned)n_cu && n_cu % 8 == 0 && (tiles_m / 8) * tiles_n >= n_cu / 8 && a.K <= 2048) {
    const int ng = (int)((3 << 20) / panel);               // 3 MiB of W panels per XCD next to the streaming A panels (measured: 6 N tiles at K = 1024
                                                           // run at the raster order's speed with 40 % less fabric traffic; 4 and 2 are 3-7 % slower: the A
                                                           // panels are re-read once per N-group, so few large groups beat many small ones)
    if (ng >= 1 && tiles_n > ng) b.ngroup = g_gemm_order > 0 ? g_gemm_order : ng;
  }
  hipLaunchKernelGGL(kern, dim3(grid, (unsigned)a.batch), dim3(512), lds, s, b, a.act, a.c_is_bf16);
  return SG_OK;
}

static thread_local int g_gemm_config = -1;                // -1 = pick per shape (tuning override, per calling thread)
int get_gemm_config() { return g_gemm_config; }
// A launch whose 256 x 256 tiles would not even fill half the CUs (one or two image tiles per call: the reference's own tile-by-tile loop)
// runs on the 128 x 128 ring kernel instead -- four times the workgroups, measured 1.4-2x faster there; tuning code 36 switches this off.
static bool few_tiles(int M, int N) {
  return g_gemm_config != 36 && (int64_t)cdiv(M, 256) * cdiv(N, 256) * 2 < device_cu_count();
}
bool gemm_bf16_ln_fold_ok(int M, int N, int K) { return M >= 1024 && N >= 512 && N % 64 == 0 && K % 32 == 0 && K / 32 >= 4; }
bool gemm_bf16_prefers_persistent(int M, int N) { return M >= 1024 && N >= 512 && !few_tiles(M, N); }
void set_gemm_config(int c) {
  if (c >= 2000) { g_persist_grid_cap = c - 2000; return; } // experiment: fewer persistent workgroups than CUs (2000 = no cap)
  if (c >= 1000) { g_gemm_order = c - 1001; return; }      // 1000 -> -1 (automatic), 1001 -> 0 (raster), 1001 + v -> N-group size v
  g_gemm_config = c;
}


template <int BM_, int BN_, int WM, int WN, int STAGES, int ABLATE = 0, int BKT = 64, bool FP8 = false, bool F16 = false, bool MXA = false, int SPEC = 0, bool SPLIT = false>
static int launch_ring(const GemmBf16Args& a, int vec, hipStream_t s) {
  auto kern = gemm_bf16_ring<BM_, BN_, WM, WN, STAGES, ABLATE, BKT, FP8, F16, MXA, SPEC, SPLIT>;
  const size_t lds = (size_t)STAGES * ((BM_ + BN_) * BKT * 2 + (MXA ? BM_ * 4 : 0));
  if (lds > 48 * 1024) SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int64_t tiles = cdiv(a.M, BM_) * cdiv(a.N, BN_);
  SG_REQUIRE(tiles < (1ll << 31) && a.batch < 65536, "gemm_bf16: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)a.batch), dim3(WM * WN * 64), lds, s, a, a.act, a.c_is_bf16, vec);
  return SG_OK;
}

template <int ACT, bool C_BF16>
static void launch(const GemmBf16Args& a, bool vec, dim3 grid, hipStream_t s) {
  if (vec) hipLaunchKernelGGL((gemm_bf16_kernel<ACT, C_BF16, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((gemm_bf16_kernel<ACT, C_BF16, false>), grid, dim3(256), 0, s, a);
}

static int gemm_fp8(const GemmBf16Args& a, hipStream_t s) {
  SG_REQUIRE(a.K % 128 == 0, "gemm_fp8: K=%d must be a multiple of 128", a.K);
  SG_REQUIRE(a.lda % 16 == 0 && a.ldw % 16 == 0 && a.strideA % 16 == 0 && a.strideW % 16 == 0, "gemm_fp8: operand strides must be multiples of 16 bytes");
  SG_REQUIRE((((uintptr_t)a.A) & 15) == 0 && (((uintptr_t)a.W) & 15) == 0, "gemm_fp8: operands must be 16-byte aligned");
  SG_REQUIRE(a.col_scale && (a.row_scale || a.a_mx), "gemm_fp8: col_scale and (row_scale or MX block scales) are required");
  if (a.c_mx) SG_REQUIRE(a.c_mx_scale && a.N % 128 == 0 && a.ldc % 8 == 0 && !a.residual, "gemm_fp8: MX output needs c_mx_scale, N %% 128 == 0, no residual");
  SG_REQUIRE(a.act >= 0 && a.act <= 2, "gemm_fp8: bad act %d", a.act);
  bool vec = (a.N % 8 == 0) && (a.ldc % 8 == 0) && (a.strideC % 8 == 0) && ((((uintptr_t)a.C) & 15) == 0) && ((((uintptr_t)a.col_scale) & 15) == 0);
  if (a.bias) vec = vec && ((((uintptr_t)a.bias) & 15) == 0);
  if (a.residual) vec = vec && (a.ldr % 4 == 0) && ((((uintptr_t)a.residual) & 15) == 0);
  GemmBf16Args h = a;                                       // the same bytes seen as a bf16 matrix of half the width
  h.K = a.K / 2; h.lda = a.lda / 2; h.ldw = a.ldw / 2; h.strideA = a.strideA / 2; h.strideW = a.strideW / 2;
  prof_begin(PROF_GEMM_FP8, 2.0 * a.M * (double)a.N * a.K * a.batch, s);
  // 256 x 256 x 128 B, two stages (1.47 / 1.32 / 1.64 PFLOP/s on the QKV / fc / proj shapes; the 256 x 128 three-stage tile 1.31 / 1.04 / 1.45)
  // large shapes: the persistent ping-pong kernel (byte strides, original K); cfg 31 (tuning) forces the two-stage ring kernel instead
  // measured (tools/bench_gemm_fp8.py, R = 175 360): proj (K 4096) 1.76 vs 1.65 PFLOP/s for the persistent kernel; QKV / fc (K 1024: 8 K steps
  // per tile, the tile switch weighs twice what it does in bf16) 1.25 / 1.23 vs 1.51 / 1.31 for the ring kernel -> persistent for long K only
  const bool mx = a.a_mx != nullptr || a.c_mx != nullptr;
  if (mx) {                                               // MX operands / MX output live in the ring kernel (the persistent kernel's LDS is full)
    SG_REQUIRE(vec && a.batch == 1 && a.M >= 1024 && a.N >= 256, "gemm_fp8: the MX forms need the large-shape vector path");
    // the tower's MLP hand-off runs on compile-time-specialised epilogues: fc -> MX output after the activation, proj <- MX operand, f32 + residual
    const bool proj_form = a.a_mx && !a.c_mx && !a.c_is_bf16 && a.residual && a.act == ACT_NONE;
    const bool fc_form = !a.a_mx && a.c_mx && (a.act == ACT_QUICK_GELU || a.act == ACT_GELU);
    const int rcm = proj_form ? launch_ring<256, 256, 2, 4, 2, 0, 64, true, false, true, 4>(h, vec, s)
                  : a.a_mx ? launch_ring<256, 256, 2, 4, 2, 0, 64, true, false, true>(h, vec, s)
                  : fc_form ? (a.act == ACT_QUICK_GELU ? launch_ring<256, 256, 2, 4, 2, 0, 64, true, false, false, 2>(h, vec, s)
                                                       : launch_ring<256, 256, 2, 4, 2, 0, 64, true, false, false, 3>(h, vec, s))
                            : launch_ring<256, 256, 2, 4, 2, 0, 64, true>(h, vec, s);
    prof_end(PROF_GEMM_FP8, s);
    if (rcm != SG_OK) return rcm;
    SG_LAUNCH_CHECK();
    return SG_OK;
  }
  const bool persist = vec && a.batch == 1 && a.M >= 1024 && a.N >= 512 && (a.K >= 2048 || g_gemm_config == 32) && a.K >= 512 && g_gemm_config != 31 &&
                       (int64_t)a.M * a.lda < (1ll << 31) && (int64_t)a.N * a.ldw < (1ll << 31);     // 32-bit byte offsets inside the kernel
  const int rc = persist ? launch_fp8_persist(a, s)
               : (a.M >= 1024 && a.N >= 256) ? ((vec && a.c_is_bf16 && !a.residual && a.act == ACT_NONE) ? launch_ring<256, 256, 2, 4, 2, 0, 64, true, false, false, 1>(h, vec, s)
                                                                                                      : launch_ring<256, 256, 2, 4, 2, 0, 64, true>(h, vec, s))
                                             : launch_ring<128, 128, 2, 2, 3, 0, 64, true>(h, vec, s);
  prof_end(PROF_GEMM_FP8, s);
  if (rc != SG_OK) return rc;
  SG_LAUNCH_CHECK();
  return SG_OK;
}

// SG_PREC_F16X2: two-plane f16 operands (common.h h2_t).  The kernels see the same bytes as f16 matrices of twice the width.
static int gemm_h2(const GemmBf16Args& a, hipStream_t s) {
  SG_REQUIRE(!a.fp8 && !a.rowdot && !a.row_scale && !a.col_scale && !a.a_mx && !a.c_mx,
             "gemm_h2: the fp8 / row-dot forms do not exist for two-plane f16 operands");
  SG_REQUIRE(a.K % 32 == 0, "gemm_h2: K=%d must be a multiple of 32 (pad the operands)", a.K);
  SG_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.strideA % 8 == 0 && a.strideW % 8 == 0, "gemm_h2: operand strides must be multiples of 8 elements (32-byte storage groups)");
  SG_REQUIRE((((uintptr_t)a.A) & 31) == 0 && (((uintptr_t)a.W) & 31) == 0, "gemm_h2: operands must be 32-byte aligned");
  SG_REQUIRE(a.act >= 0 && a.act <= 2, "gemm_h2: bad act %d", a.act);
  bool vec = (a.N % 8 == 0) && (a.ldc % 8 == 0) && (a.strideC % 8 == 0) && ((((uintptr_t)a.C) & 31) == 0);
  if (a.bias) vec = vec && ((((uintptr_t)a.bias) & 15) == 0);
  if (a.residual) vec = vec && (a.ldr % 4 == 0) && ((((uintptr_t)a.residual) & 15) == 0);
  if (a.c_is_bf16) SG_REQUIRE(vec && !a.residual, "gemm_h2: a two-plane C needs N %% 8 == 0, ldc %% 8 == 0, a 32-byte aligned C and no residual");
  GemmBf16Args h = a;
  h.K = a.K * 2; h.lda = a.lda * 2; h.ldw = a.ldw * 2; h.strideA = a.strideA * 2; h.strideW = a.strideW * 2; h.f16 = 1;
  const bool big = a.M >= 1024 && a.N >= 512 && !(few_tiles(a.M, a.N) && a.batch == 1);
  // the persistent kernel: batch 1, the vector epilogue, >= 4 K steps per tile (its look-ahead is 3), tile-relative 32-bit offsets;
  // tuning code 37 keeps the plain ping-pong kernel (A/B measurements)
  const bool hot_form = (a.c_is_bf16 && !a.residual) || (!a.c_is_bf16 && a.residual && a.act == ACT_NONE);   // the forms with a compile-time epilogue (the run-time one spills)
  const bool fits = a.batch == 1 && a.K >= 128 && a.lda * 4 < (1 << 23) && a.ldw * 4 < (1 << 23);
  const bool ln_fold = a.copy16 != nullptr || a.ln_stats != nullptr;   // folded LayerNorm (round 3): the persistent kernel's own epilogue forms, as in the 2-byte modes
  if (ln_fold) {
    SG_REQUIRE(vec && fits && gemm_bf16_ln_fold_ok(a.M, a.N, a.K), "gemm_h2: the folded-LayerNorm epilogues need the persistent kernel (M >= 1024, N >= 512, K >= 128, batch 1)");
    if (a.copy16) SG_REQUIRE(a.row_stats && !a.c_is_bf16 && a.residual && a.act == ACT_NONE && a.N % 64 == 0 && a.ld16 % 8 == 0 && ((((uintptr_t)a.copy16) & 31) == 0),
                             "gemm_h2: copy16 needs row_stats, an f32 C with residual, no activation and N %% 64 == 0");
    if (a.ln_stats) SG_REQUIRE(a.ln_c && a.alpha == 1.f && a.c_is_bf16 && !a.residual && ((((uintptr_t)a.ln_c) & 15) == 0) && ((((uintptr_t)a.ln_stats) & 7) == 0),
                               "gemm_h2: ln_stats needs ln_c, alpha 1 and a two-plane C");
  }
  const bool persist = ln_fold || (big && vec && hot_form && fits && g_gemm_config != 37);
  prof_begin(PROF_GEMM_H2, 2.0 * a.M * (double)a.N * a.K * a.batch, s);
  const int rc = persist ? launch_h2_persist(h, s) : big ? launch_pingpong(h, vec, s) : launch_ring<128, 128, 2, 2, 2, 0, 64, false, true, false, 0, true>(h, vec, s);
  prof_end(PROF_GEMM_H2, s);
  if (rc != SG_OK) return rc;
  SG_LAUNCH_CHECK();
  return SG_OK;
}

int gemm_bf16(const GemmBf16Args& a, hipStream_t s) {
  SG_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.batch > 0, "gemm_bf16: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  if (a.h2) return gemm_h2(a, s);
  if (a.fp8) return gemm_fp8(a, s);
  SG_REQUIRE(a.K % BK == 0, "gemm_bf16: K=%d must be a multiple of %d (pad the operands)", a.K, BK);
  SG_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.strideA % 8 == 0 && a.strideW % 8 == 0,
             "gemm_bf16: operand strides must be multiples of 8 elements (16-byte chunks)");
  SG_REQUIRE((((uintptr_t)a.A) & 15) == 0 && (((uintptr_t)a.W) & 15) == 0, "gemm_bf16: operands must be 16-byte aligned");
  const int csz = a.c_is_bf16 ? 2 : 4;
  bool vec = (a.N % 8 == 0) && (a.ldc % 8 == 0) && (a.strideC % 8 == 0) && ((((uintptr_t)a.C) & 15) == 0);
  if (a.bias) vec = vec && ((((uintptr_t)a.bias) & 15) == 0);
  if (a.residual) vec = vec && (a.ldr % 4 == 0) && ((((uintptr_t)a.residual) & 15) == 0);
  (void)csz;
  SG_REQUIRE(a.act >= 0 && a.act <= 2, "gemm_bf16: bad act %d", a.act);
  if (a.rowdot)
    SG_REQUIRE(vec && a.residual && !a.c_is_bf16 && a.batch == 1 && a.M >= 1024 && a.N >= 512 && a.N % 64 == 0 && a.K / 32 >= 4,
               "gemm_bf16: the row-dot epilogue needs the persistent kernel (M >= 1024, N >= 512, N %% 64 == 0), an f32 residual and batch 1");
  const bool ln_fold = a.copy16 != nullptr || a.ln_stats != nullptr;
  if (ln_fold) {
    SG_REQUIRE(vec && a.batch == 1 && gemm_bf16_ln_fold_ok(a.M, a.N, a.K), "gemm_bf16: the folded-LayerNorm epilogues need the persistent kernel (M >= 1024, N >= 512, batch 1)");
    if (a.copy16) SG_REQUIRE(a.row_stats && !a.c_is_bf16 && !a.rowdot && a.N % 64 == 0 && a.ld16 % 4 == 0 && ((((uintptr_t)a.copy16) & 7) == 0), "gemm_bf16: copy16 needs row_stats, an f32 C and N %% 64 == 0");
    if (a.ln_stats) SG_REQUIRE(a.ln_c && a.alpha == 1.f && a.c_is_bf16 && !a.residual && !a.rowdot && ((((uintptr_t)a.ln_c) & 15) == 0) && ((((uintptr_t)a.ln_stats) & 7) == 0), "gemm_bf16: ln_stats needs ln_c and alpha 1");
  }
  int cfg = g_gemm_config;
  if (cfg == 33 || cfg == 34 || cfg == 36 || cfg == 37) cfg = -1;                    // tuning codes read by capi.hip (MX hand-off / LayerNorm folding off), not tile configurations
  if (a.rowdot || ln_fold) cfg = 30;
  if (a.res_half) {                                          // 2-byte residual: the small-tile kernel's run-time epilogue only
    SG_REQUIRE(vec && a.residual && a.c_is_bf16 && !a.rowdot && !ln_fold && a.ldr % 8 == 0, "gemm_bf16: res_half needs a 2-byte C, N %% 8 == 0 and ldr %% 8 == 0");
    cfg = 4;
  }
  if (!a.res_half && (cfg < 0 || a.f16)) cfg = (a.rowdot || ln_fold || (a.M >= 1024 && a.N >= 512 && !(few_tiles(a.M, a.N) && a.batch == 1))) ? 30 : 4;  // large: persistent ping-pong; small: 128x128 tiles (more workgroups); f16 operands: these two only
  if (cfg > 0) {
    const int pcat = (cfg == 30 && vec && a.K / 32 >= 4) ? (a.ln_stats ? PROF_GEMM_PERSIST_LN_CONSUMER : a.copy16 ? PROF_GEMM_PERSIST_LN_PRODUCER : PROF_GEMM_PERSIST)
                                                         : PROF_GEMM_BF16;
    prof_begin(pcat, 2.0 * a.M * (double)a.N * a.K * a.batch, s);
    int rc;
    switch (cfg) {
      case 1: rc = launch_ring<128, 128, 2, 2, 3>(a, vec, s); break;
      case 2: rc = launch_ring<256, 128, 4, 2, 3>(a, vec, s); break;
      case 3: rc = launch_ring<256, 256, 2, 4, 2>(a, vec, s); break;
      case 4: {                                             // 128 x 128 tiles: small shapes and small launches; hot epilogue combinations specialised
        const int sp = !vec || a.c_mx || a.row_scale || a.col_scale ? 0
                     : (a.c_is_bf16 && !a.residual) ? (a.act == ACT_NONE ? 1 : a.act == ACT_QUICK_GELU ? 6 : 7)
                     : (!a.c_is_bf16 && a.residual && a.act == ACT_NONE) ? 4 : 0;
#define SG_RING128(SP) (a.f16 ? launch_ring<128, 128, 2, 2, 2, 0, 64, false, true, false, SP>(a, vec, s) : launch_ring<128, 128, 2, 2, 2, 0, 64, false, false, false, SP>(a, vec, s))
        rc = sp == 1 ? SG_RING128(1) : sp == 4 ? SG_RING128(4) : sp == 6 ? SG_RING128(6) : sp == 7 ? SG_RING128(7) : SG_RING128(0);
#undef SG_RING128
        break;
      }
      case 5: rc = launch_ring<256, 128, 4, 2, 2>(a, vec, s); break;
      case 6: rc = launch_ring<128, 256, 2, 4, 3>(a, vec, s); break;
      case 7: rc = launch_pingpong(a, vec, s); break;
      case 30: rc = (vec && a.K / 32 >= 4) ? launch_persist(a, s) : launch_pingpong(a, vec, s); break;
      case 9: rc = launch_ring<128, 256, 1, 4, 3, 0, 32>(a, vec, s); break;    // 72 KiB LDS: two workgroups per CU
      case 10: rc = launch_ring<256, 128, 4, 1, 3, 0, 32>(a, vec, s); break;
      case 8: rc = vec ? launch_pp32<0>(a, s) : launch_pingpong(a, vec, s); break;
#ifdef SG_GEMM_ABLATIONS                                   // tuning builds only: these variants drop loads / MFMAs / stores and return WRONG results
      case 21: rc = launch_pp32<1>(a, s); break;          // ablations of the pp32 kernel (wrong results by design)
      case 22: rc = launch_pp32<2>(a, s); break;
      case 23: rc = launch_pp32<3>(a, s); break;
      case 24: rc = launch_pp32<4>(a, s); break;
      case 11: rc = launch_ring<256, 256, 2, 4, 2, 1>(a, vec, s); break;   // ablations (wrong results by design)
      case 12: rc = launch_ring<256, 256, 2, 4, 2, 2>(a, vec, s); break;
      case 13: rc = launch_ring<128, 128, 2, 2, 2, 1>(a, vec, s); break;
      case 15: rc = launch_ring<256, 256, 2, 4, 2, 3>(a, vec, s); break;
      case 16: rc = launch_ring<256, 256, 2, 4, 2, 4>(a, vec, s); break;
      case 17: rc = launch_ring<128, 128, 2, 2, 2, 3>(a, vec, s); break;
      case 18: rc = launch_ring<128, 128, 2, 2, 2, 4>(a, vec, s); break;
      case 14: rc = launch_ring<128, 128, 2, 2, 2, 2>(a, vec, s); break;
#endif
      default: return fail(SG_ERR_INVALID, "gemm_bf16: unknown tile config %d", cfg);
    }
    prof_end(pcat, s);
    if (rc != SG_OK) return rc;
    SG_LAUNCH_CHECK();
    return SG_OK;
  }
  const int64_t tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
  SG_REQUIRE(tiles < (1ll << 31) && a.batch < 65536, "gemm_bf16: grid too large");
  dim3 grid((unsigned)tiles, (unsigned)a.batch);
  prof_begin(PROF_GEMM_BF16, 2.0 * a.M * (double)a.N * a.K * a.batch, s);
  switch (a.act * 2 + (a.c_is_bf16 ? 1 : 0)) {
    case 0: launch<ACT_NONE, false>(a, vec, grid, s); break;
    case 1: launch<ACT_NONE, true>(a, vec, grid, s); break;
    case 2: launch<ACT_QUICK_GELU, false>(a, vec, grid, s); break;
    case 3: launch<ACT_QUICK_GELU, true>(a, vec, grid, s); break;
    case 4: launch<ACT_GELU, false>(a, vec, grid, s); break;
    case 5: launch<ACT_GELU, true>(a, vec, grid, s); break;
    default: return fail(SG_ERR_INVALID, "gemm_bf16: bad act %d", a.act);
  }
  prof_end(PROF_GEMM_BF16, s);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

}  // namespace sg
