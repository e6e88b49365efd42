// Shared host/device helpers for libsegearth_hip (gfx950 only; no CUDA dual path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include "../../include/segearth_hip.h"

namespace sg {

typedef uint16_t bf16_t;                                            // raw bf16 bits in HBM
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;          // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int WAVE = 64;

// ---- error plumbing (no exceptions cross the ABI) -------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define SG_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t _e = (call);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return ::sg::fail(SG_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
  } while (0)

#define SG_REQUIRE(cond, ...)                                                                \
  do {                                                                                       \
    if (!(cond)) return ::sg::fail(SG_ERR_INVALID, __VA_ARGS__);                             \
  } while (0)

#define SG_LAUNCH_CHECK()                                                                    \
  do {                                                                                       \
    hipError_t _e = hipGetLastError();                                                       \
    if (_e != hipSuccess)                                                                    \
      return ::sg::fail(SG_ERR_HIP, "%s:%d launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
  } while (0)

#define SG_TRY(expr)                                                                         \
  do {                                                                                       \
    int _rc = (expr);                                                                        \
    if (_rc != SG_OK) return _rc;                                                            \
  } while (0)

// ---- per-device launch bookkeeping: no per-process "done once" flags (a second device / a second host thread must work) ----
// ensure_dynamic_lds: opt a kernel in to `bytes` of dynamic LDS on the CURRENT device, once per (device, kernel), thread-safe.
// device_cu_count: compute units of the current device (cached per device).  DeviceGuard: make a context's device current for
// the duration of an entry point (a context created on cuda:1 must launch on cuda:1 whatever the caller's current device is).
int ensure_dynamic_lds(const void* kernel, size_t bytes);
int device_cu_count();
struct DeviceGuard {
  int prev = -1, dev;
  explicit DeviceGuard(int d) : dev(d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); }
  ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete; DeviceGuard& operator=(const DeviceGuard&) = delete;
};

static inline hipStream_t as_stream(sg_stream s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ---- bf16 <-> f32 -------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {                    // round-to-nearest-even, NaN kept
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

// ---- f16 (SG_PREC_F16): a distinct storage type so the row kernels can be instantiated for it ----------------
// The reference's GPU arithmetic is fp16 + fp32 LayerNorm (segmentor.py:467, open_clip/model.py:142); gfx950 runs the f16 MFMA
// forms at the bf16 rate.  Stores SATURATE at +-65504 instead of producing inf (QKV / fc outputs of a trained tower can be large).
struct f16_t { uint16_t bits; };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
constexpr float F16_MAX = 65504.0f;
__device__ __forceinline__ float h2f(f16_t v) { _Float16 h; __builtin_memcpy(&h, &v.bits, 2); return (float)h; }
__device__ __forceinline__ f16_t f2h(float f) {
  const _Float16 h = (_Float16)__builtin_amdgcn_fmed3f(f, -F16_MAX, F16_MAX);   // round-to-nearest-even; NaN stays NaN
  f16_t o; __builtin_memcpy(&o.bits, &h, 2); return o;
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
// two f32 -> packed f16 (one v_cvt_pk_f16_f32, round-to-nearest-even) behind two saturating v_med3_f32
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
  const f32x2_t v = {__builtin_amdgcn_fmed3f(lo, -F16_MAX, F16_MAX), __builtin_amdgcn_fmed3f(hi, -F16_MAX, F16_MAX)};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
// Sums over aligned groups of 8 / 16 lanes, result in every lane of the group, on DPP (data-parallel primitives: the add reads its second
// operand from another lane, one VALU instruction) instead of __shfl_xor, which hipcc lowers to ds_bpermute_b32 -- an LDS-queue round trip
// per step.  Steps: quad xor 1, quad xor 2, half-row mirror (i <-> 7 - i: pairs the two quads), row mirror (i <-> 15 - i: pairs the two
// halves).  Same pairs in the first two steps and commutative adds afterwards: bit-identical to the xor butterfly.
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum8_dpp(float v) { v += dpp_f32<0xB1>(v); v += dpp_f32<0x4E>(v); v += dpp_f32<0x141>(v); return v; }
__device__ __forceinline__ float sum16_dpp(float v) { v = sum8_dpp(v); v += dpp_f32<0x140>(v); return v; }
// half-precision kind of a buffer / kernel: 0 = f32, 1 = bf16, 2 = f16, 3 = two-plane f16 (every `int ..._bf16` flag of the internal ops takes these values)
enum HalfKind { HK_F32 = 0, HK_BF16 = 1, HK_F16 = 2, HK_F16X2 = 3 };

// ---- two-plane f16 (SG_PREC_F16X2): x = hi + lo with hi = f16(x), lo = f16(x - hi) -- 22 significant bits on the f16 matrix pipe --------
// A product A.W^T is issued as A_hi.W_hi + A_hi.W_lo + A_lo.W_hi into ONE f32 accumulator (the lo.lo term is below f32 resolution): three
// MFMAs per K step, measured at or below the error of an f32 fmaf chain (tools/h2_probe.hip: K = 1024, 4.0e-5 against 6.1e-5 on O(1) data).
// lo is NOT rescaled: gfx950 converts to f16 subnormals and its f16 MFMA honours subnormal inputs on both ports (same probe), so the
// representation error of an element is max(2^-22 |x|, 2^-25) -- f32-grade for the O(0.01 .. 100) data of this path.  |x| above 2 * 65504
// is not representable (both planes saturate); parity mode (SG_PREC_F32) has no such bound.
// Storage: groups of 8 consecutive elements as [8 x hi f16][8 x lo f16] = 32 bytes, i.e. exactly the f32 footprint; element offsets that
// are multiples of 8 address like 4-byte elements, so `h2_t` is the pointer-arithmetic type (never dereferenced as a value).  One 32-byte
// load hands a lane both planes of its 8 K values: the hi and the lo fragment of a v_mfma_f32_16x16x32_f16 / 32x32x16 operand.
struct h2_t { uint32_t raw; };
// two f32 -> (packed hi pair, packed lo pair); hi saturates at +-65504, lo carries what is left (saturating too)
__device__ __forceinline__ void split_h2(float a, float b, uint32_t& hi, uint32_t& lo) {
  const f32x2_t v = {__builtin_amdgcn_fmed3f(a, -F16_MAX, F16_MAX), __builtin_amdgcn_fmed3f(b, -F16_MAX, F16_MAX)};
  const f16x2_t h = __builtin_convertvector(v, f16x2_t);
  const f32x2_t back = __builtin_convertvector(h, f32x2_t);
  const f32x2_t r = {__builtin_amdgcn_fmed3f(a - back[0], -F16_MAX, F16_MAX), __builtin_amdgcn_fmed3f(b - back[1], -F16_MAX, F16_MAX)};
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2_t));
}
// the same without the saturating clamps, for values known to lie well inside the f16 range (softmax probabilities <= 2^8): 6 instead of 10 VALU per pair
__device__ __forceinline__ void split_h2_bounded(float a, float b, uint32_t& hi, uint32_t& lo) {
  const f32x2_t v = {a, b};
  const f16x2_t h = __builtin_convertvector(v, f16x2_t);
  const f32x2_t back = __builtin_convertvector(h, f32x2_t);
  const f32x2_t r = {a - back[0], b - back[1]};
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2_t));
}
// 8 consecutive elements (one storage group) -> the 32 bytes of the group
__device__ __forceinline__ void split_h2x8(const float (&v)[8], uint4& hi, uint4& lo) {
  split_h2(v[0], v[1], hi.x, lo.x); split_h2(v[2], v[3], hi.y, lo.y); split_h2(v[4], v[5], hi.z, lo.z); split_h2(v[6], v[7], hi.w, lo.w);
}
__device__ __forceinline__ void store_h2x8(h2_t* group, const float (&v)[8]) {      // `group` = address of element 8g of a row
  uint4 hi, lo; split_h2x8(v, hi, lo);
  reinterpret_cast<uint4*>(group)[0] = hi; reinterpret_cast<uint4*>(group)[1] = lo;
}
static inline int hk_of_precision(int p) { return p == SG_PREC_F32 ? HK_F32 : p == SG_PREC_F16 ? HK_F16 : p == SG_PREC_F16X2 ? HK_F16X2 : HK_BF16; }
static inline size_t hk_esz(int hk) { return (hk == HK_BF16 || hk == HK_F16) ? 2 : 4; }    // bytes per element of an operand buffer
template <bool F16> __device__ __forceinline__ uint32_t pack_half2(float lo, float hi) { return F16 ? pack_h2(lo, hi) : pack_bf2(lo, hi); }
__device__ __forceinline__ uint32_t pack_half2(int kind, float lo, float hi) { return kind == HK_F16 ? pack_h2(lo, hi) : pack_bf2(lo, hi); }
// MFMA on 2-byte operands held as bf16x8 bit patterns: the f16 forms take the same cycles as the bf16 forms (MI355X_MICROARCH.md)
template <bool F16> __device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf2f(v); }
template <> __device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return h2f(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f2bf(v); }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return f2h(v); }
// Element i of a row (row kernels that walk a row element by element).  For h2_t the row base must sit on a storage-group boundary
// (a multiple of 8 elements); the element's planes are 16 bytes apart inside its 32-byte group.
template <typename T> __device__ __forceinline__ float ld_elem(const T* row, int64_t i) { return to_f32<T>(row[i]); }
template <> __device__ __forceinline__ float ld_elem<h2_t>(const h2_t* row, int64_t i) {
  const uint16_t* p = reinterpret_cast<const uint16_t*>(row) + ((i >> 3) << 4) + (i & 7);
  return h2f(f16_t{p[0]}) + h2f(f16_t{p[8]});
}
template <typename T> __device__ __forceinline__ void st_elem(T* row, int64_t i, float v) { row[i] = from_f32<T>(v); }
template <> __device__ __forceinline__ void st_elem<h2_t>(h2_t* row, int64_t i, float v) {
  uint16_t* p = reinterpret_cast<uint16_t*>(row) + ((i >> 3) << 4) + (i & 7);
  const f16_t h = f2h(v);
  p[0] = h.bits; p[8] = f2h(v - h2f(h)).bits;
}

// ---- wave-level reductions (64 lanes) ----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// The same reductions without the LDS queue: DPP adds inside each row of 16 lanes, then the four row results through v_readlane_b32
// (every lane gets the same scalar).  For kernels whose whole wave is active; summation order differs from the xor butterfly's.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  const int r = __builtin_bit_cast(int, sum16_dpp(v));
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 48)));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v)); v = fmaxf(v, dpp_f32<0x4E>(v)); v = fmaxf(v, dpp_f32<0x141>(v)); v = fmaxf(v, dpp_f32<0x140>(v));
  const int r = __builtin_bit_cast(int, v);
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 48))));
}

// activations of the MLP (reference open_clip/transformer.py:35-38 / nn.GELU)
// throughput-mode QuickGELU: x * rcp(1 + 2^(-1.702 log2e x)) -- v_exp_f32 + v_rcp_f32 (1 ulp each), inputs end up in bf16 anyway
__device__ __forceinline__ float quick_gelu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * x)); }
__device__ __forceinline__ float quick_gelu_exact(float x) { return x * (1.0f / (1.0f + expf(-1.702f * x))); }
// the two-plane mode's QuickGELU: the reference's own first rounding (t = -1.702 x), then e^t on the hardware exponential with the log2(e)
// product carried in two pieces (argument error << 1 ulp) and the hardware reciprocal (1 ulp) in place of expf + IEEE division: 7 issue slots
// instead of ~25, within 2 ulp of quick_gelu_exact (tests/test_gpu_ops.py::test_linear f16x2 cases hold the GEMM + activation to 2e-5 of f64)
__device__ __forceinline__ float quick_gelu_split(float x) {
  const float t = -1.702f * x;
  const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(t, 1.44269502162933349609f, t * 1.92596299112661746e-8f));
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float erf_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// The 2-byte / fp8 GEMM epilogues (whose results are rounded to 8-11 mantissa bits anyway): erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7 absolute) on the hardware exp2 / rcp -- about a third of erff's instructions.  Parity mode keeps erff.
__device__ __forceinline__ float erf_gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);      // erf(|x| / sqrt 2)
  return 0.5f * x + 0.5f * fabsf(x) * e;                                                   // 0.5 x (1 + sign(x) erf(|x| / sqrt 2))
}

enum Act { ACT_NONE = 0, ACT_QUICK_GELU = 1, ACT_GELU = 2 };

// ---- optional live kernel timing (bench.py roofline): HIP events around the launches of one kernel family ----
enum ProfCat { PROF_GEMM_BF16 = 0, PROF_ATTENTION = 1, PROF_GEMM_F32 = 2, PROF_GEMM_PERSIST = 3, PROF_GEMM_FP8 = 4, PROF_GEMM_PERSIST_LN_CONSUMER = 5,
               PROF_GEMM_PERSIST_LN_PRODUCER = 6, PROF_GEMM_H2 = 7, PROF_NCAT = 8 };   // 7: the two-plane f16 GEMM (work = ALGORITHMIC 2 M N K; the kernel issues 3x that on the matrix pipe)   // 5 / 6: the persistent kernel's folded-LayerNorm instantiations (sg_profile_read(3) includes them)   // 3: the persistent bf16 kernel alone (the dominant kernel bench.py prices)
bool prof_on();
void prof_begin(int cat, double work, hipStream_t s);   // work = algorithmic FLOPs of the launch
void prof_end(int cat, hipStream_t s);

// ---- internal op entry points (defined across the .hip files) -------------------------------------
// bf16 GEMM: C[M,N] = act(A[M,K] . W[N,K]^T + bias) (+ residual).  A, W bf16 (K contiguous, K % 64 == 0).
struct GemmBf16Args {
  const bf16_t* A; int64_t lda; int64_t strideA;      // batch stride (elements)
  const bf16_t* W; int64_t ldw; int64_t strideW;
  const float* bias;                                  // [N] or null
  const float* residual; int64_t ldr;                 // f32 [M,N] or null (batch stride = strideC)
  int res_half;                                       // the residual is 2-byte in the operands' type (ldr in elements): 2-byte outputs of the small-tile kernels only
  void* C; int64_t ldc; int64_t strideC; int c_is_bf16;   // C: 0 = f32, 1 = the operands' 2-byte type (bf16, or f16 when `f16` is set)
  int f16;                                            // operands (and a 2-byte C) are IEEE f16 instead of bf16: SG_PREC_F16
  int h2;                                             // SG_PREC_F16X2: A, W (and C when c_is_bf16) are two-plane f16 (h2_t: 4 bytes per element, lda / ldw / ldc /
                                                      // strides in ELEMENTS, multiples of 8); K % 32 == 0; f32-grade activations; folded-LayerNorm forms as below (copy16 two-plane); no row-dot / fp8 forms
  int M, N, K, batch, act;
  float alpha;                                        // applied to the accumulator before bias
  // fp8 mode (fp8 != 0): A and W hold OCP e4m3 bytes ([M,K] / [N,K], K % 128 == 0); lda / ldw / K stay in ELEMENTS (= bytes);
  // the accumulator is de-quantised with row_scale[m] * col_scale[n] (per-token / per-output-channel absmax scales).
  int fp8; const float* row_scale; const float* col_scale;
  // fp8 mode, MX block scaling (OCP MXFP8: one E8M0 power-of-two scale per 32 consecutive K elements, consumed by the scale operands of
  // v_mfma_scale_f32_16x16x128_f8f6f4 -- the scale of K block b of row r is read from lane r + 16 b, tools/mx_probe.hip):
  //   a_mx      : A's block scales, laid out [K/128][M][4] bytes (the four blocks of one K step of a row are one dword); row_scale is then unused
  //   c_mx      : write C as e4m3 bytes [M, N] with block scales c_mx_scale in the same [N/128][M][4] layout (C / c_is_bf16 ignored):
  //               the next linear's MX operand straight out of this epilogue, no separate quantisation pass
  const uint8_t* a_mx; uint8_t* c_mx; uint8_t* c_mx_scale;
  // LayerNorm folded into the two GEMMs either side of it (persistent kernel only, batch 1; capi.hip std_block).  LN(x).W^T + b =
  // rstd (x.W'^T - mean c) + b' with W' = gamma o W, c[n] = sum_k W'[n][k], b' = b + W.beta, so the normalisation never runs as a pass:
  //   producer (f32 C, with or without residual): copy16 / ld16 = a 2-byte copy of the finished rows (the next GEMM's A operand),
  //     row_stats [N/64][M][2] = (sum, centred sum of squares) of every 64-column slice of a finished row (slice-major: full-line stores)
  //   consumer: ln_stats [M][2] = (mean, rstd) per row, ln_c = c; bias = b'
  void* copy16; int64_t ld16; float* row_stats;
  const float* ln_stats; const float* ln_c;
  // persistent kernel only: row-dot epilogue.  With v = alpha * acc + bias and r = residual, NOTHING is stored to C; instead
  // rowdot[m * rowdot_ld + n / 64] = sum over the 64 columns [n, n + 64) of v * (2 r + v)  (= |r + v|^2 - |r|^2 of that column slice):
  // the JBU tail needs only the norm of x + 0.1 * conv1x1(x), never the C x S^2 map itself.  f32 residual required.
  float* rowdot; int64_t rowdot_ld;
  int rowdot_res_bf16;
  // persistent kernel only: W-panel-resident tile order.  ngroup > 0: every XCD owns a block of M tiles and walks them N-group by
  // N-group (ngroup N tiles at a time, chosen so that their W panels fit the XCD's 4 MiB L2 next to the streaming A panels):
  // the W panels are fetched once per XCD instead of once per round of workgroups.  0 = raster order in XCD chunks.
  int ngroup;                                // row-dot mode: `residual` points to bf16 (same ldr, in elements) instead of f32
};
int gemm_bf16(const GemmBf16Args& a, hipStream_t s);
int get_gemm_config();
bool gemm_bf16_ln_fold_ok(int M, int N, int K);   // the shapes the persistent kernel (home of the folded-LayerNorm epilogues) can run
bool gemm_bf16_prefers_persistent(int M, int N);  // ... and whether the automatic dispatch would pick it (enough 256 x 256 tiles to fill the chip)
void set_gemm_config(int c);   // tuning hook (per calling thread): -1 auto, 0 = 128x128x2-stage baseline, 1.. = ring variants

// f32 GEMM (f32 MFMA, exact fmaf chains), fully general strides: A(m,k) at A[m*lda + k];
// B(k,n) at B[k*sbk + n*sbn]; two-level batch: z -> (z / inner, z % inner).
struct GemmF32Args {
  const float* A; int64_t lda; int64_t sAo, sAi;
  const float* B; int64_t sbk, sbn; int64_t sBo, sBi;
  const float* bias; const float* residual; int64_t ldr;
  float* C; int64_t ldc; int64_t sCo, sCi;
  int M, N, K, batch, inner, act;
  float alpha;
};
int gemm_f32(const GemmF32Args& a, hipStream_t s);

}  // namespace sg
