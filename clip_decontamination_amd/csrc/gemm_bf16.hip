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
          if (ACT == ACT_GELU) v[e] = erf_gelu(v[e]);
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
          if (ACT == ACT_GELU) x = erf_gelu(x);
          if (res) x += res[(int64_t)m * a.ldr + n + e];
          if (C_BF16) reinterpret_cast<bf16_t*>(a.C)[(int64_t)z * a.strideC + (int64_t)m * a.ldc + n + e] = f2bf(x);
          else reinterpret_cast<float*>(a.C)[(int64_t)z * a.strideC + (int64_t)m * a.ldc + n + e] = x;
        }
      }
    }
  }
}

template <int ACT, bool C_BF16>
static void launch(const GemmBf16Args& a, bool vec, dim3 grid, hipStream_t s) {
  if (vec) hipLaunchKernelGGL((gemm_bf16_kernel<ACT, C_BF16, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((gemm_bf16_kernel<ACT, C_BF16, false>), grid, dim3(256), 0, s, a);
}

int gemm_bf16(const GemmBf16Args& a, hipStream_t s) {
  SG_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.batch > 0, "gemm_bf16: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  SG_REQUIRE(a.K % BK == 0, "gemm_bf16: K=%d must be a multiple of %d (pad the operands)", a.K, BK);
  SG_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.strideA % 8 == 0 && a.strideW % 8 == 0,
             "gemm_bf16: operand strides must be multiples of 8 elements (16-byte chunks)");
  SG_REQUIRE((((uintptr_t)a.A) & 15) == 0 && (((uintptr_t)a.W) & 15) == 0, "gemm_bf16: operands must be 16-byte aligned");
  const int csz = a.c_is_bf16 ? 2 : 4;
  bool vec = (a.N % 4 == 0) && (a.ldc % 4 == 0) && (a.strideC % 4 == 0) && ((((uintptr_t)a.C) & 15) == 0);
  if (a.bias) vec = vec && ((((uintptr_t)a.bias) & 15) == 0);
  if (a.residual) vec = vec && (a.ldr % 4 == 0) && ((((uintptr_t)a.residual) & 15) == 0);
  (void)csz;
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
