// Probe written while chasing the run-to-run differences of jbu_conv_lowres_kernel (round 3, DESIGN.md section 4 'JBU reproducibility').
// Observation there: with the kernel's Keff arithmetic SLP-packed into v_pk_fma_f32 rare pixels changed from run to run -- always lanes
// 48..63, always the HIGH half of the packed results (the last quarter of the last pass of the instruction), and only in launches with
// more workgroups than the chip holds, i.e. once co-resident workgroups ran different phases (one in its MFMA loop, the other in this
// arithmetic).  In that code the register allocator reuses a source of a packed op as the destination of the next ds_read (34 such
// pairs, distance 1..8 instructions; legal for ordinary VALU instructions, whose operands are read at issue).
// Question asked here: does that write-after-read pair alone corrupt a v_pk_fma_f32 when the SIMD is busy with another wave's MFMAs?
//   waves 0..3 of every workgroup: a dependent MFMA chain (one wave per SIMD), on / off
//   waves 4..7: v_pk_fma_f32 d, a, b, c ; ds_read_b64 b <- poison ; wait ; check d == fma(a, b_old, c), per lane and half
// RESULT on MI355X (r03): 0 wrong results out of 5.2e9 per configuration, packed or scalar, MFMA waves on or off -- the two-instruction
// pair does NOT reproduce the defect; its root cause is unconfirmed.  What is established: the scalar build of jbu.hip
// (-fno-slp-vectorize, build.py) is bit-reproducible and tests/test_gpu_repro.py holds every path of the library to that.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_war_probe.hip -o tools/pk_war_probe && tools/pk_war_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int PACKED>
__global__ __launch_bounds__(512) void probe(int iters, int with_mfma, unsigned* bad_lo, unsigned* bad_hi, float* sink) {
  __shared__ float poison[512 * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  poison[2 * tid] = 1.0e30f; poison[2 * tid + 1] = -1.0e30f;
  __syncthreads();
  if (wave < 4) {
    if (!with_mfma) return;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + lane); b[j] = (short)(0x3c00 + j); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u], 0, 0, 0);
    }
    sink[blockIdx.x * 512 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    return;
  }
  const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(&poison[2 * tid]);
  unsigned nlo = 0, nhi = 0;
  for (int it = 0; it < iters; ++it) {
    f32x2 a = {1.0f + lane * 0.25f + it, 2.0f + lane * 0.5f}, b = {3.0f + (it & 7), 0.5f + lane}, c = {0.125f * it, -7.0f};
    f32x2 d;
    if (PACKED) {
      asm volatile(
          "v_pk_fma_f32 %0, %2, %1, %3\n"
          "ds_read_b64 %1, %4\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(d), "+v"(b)
          : "v"(a), "v"(c), "v"(lds_addr)
          : "memory");
    } else {
      float d0, d1, b0 = b[0], b1 = b[1];
      asm volatile(
          "v_fma_f32 %0, %4, %2, %6\n"
          "v_fma_f32 %1, %5, %3, %7\n"
          "ds_read_b32 %3, %8 offset:4\n"
          "ds_read_b32 %2, %8\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(d0), "=&v"(d1), "+v"(b0), "+v"(b1)
          : "v"(a[0]), "v"(a[1]), "v"(c[0]), "v"(c[1]), "v"(lds_addr)
          : "memory");
      d[0] = d0; d[1] = d1; b[0] = b0; b[1] = b1;
    }
    const float e0 = __builtin_fmaf(a[0], 3.0f + (it & 7), c[0]), e1 = __builtin_fmaf(a[1], 0.5f + lane, c[1]);
    nlo += d[0] != e0; nhi += d[1] != e1;
    if (b[0] != 1.0e30f || b[1] != -1.0e30f) nlo += 1u << 20;              // the load itself must still deliver the poison
  }
  atomicAdd(&bad_lo[lane], nlo);
  atomicAdd(&bad_hi[lane], nhi);
}

int main() {
  unsigned *blo, *bhi; float* sink;
  hipMalloc(&blo, 256); hipMalloc(&bhi, 256); hipMalloc(&sink, 1024 * 512 * 4);
  for (int packed = 1; packed >= 0; --packed)
    for (int with_mfma = 0; with_mfma < 2; ++with_mfma) {
      hipMemset(blo, 0, 256); hipMemset(bhi, 0, 256);
      if (packed) probe<1><<<1024, 512>>>(20000, with_mfma, blo, bhi, sink);
      else probe<0><<<1024, 512>>>(20000, with_mfma, blo, bhi, sink);
      if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
      unsigned lo[64], hi[64];
      hipMemcpy(lo, blo, 256, hipMemcpyDeviceToHost); hipMemcpy(hi, bhi, 256, hipMemcpyDeviceToHost);
      unsigned long long q[4][2] = {};
      for (int l = 0; l < 64; ++l) { q[l >> 4][0] += lo[l]; q[l >> 4][1] += hi[l]; }
      printf("%s, MFMA waves %s: wrong results per 16-lane quarter (low half / high half):", packed ? "v_pk_fma_f32" : "2 x v_fma_f32 ", with_mfma ? "on " : "off");
      for (int k = 0; k < 4; ++k) printf("  [%d-%d] %llu / %llu", 16 * k, 16 * k + 15, q[k][0], q[k][1]);
      printf("   of %llu per quarter\n", 1024ull * 4 * 16 * 20000);
    }
  return 0;
}
