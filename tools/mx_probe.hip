// Probe of v_mfma_scale_f32_16x16x128_f8f6f4's scale operands (no ISA manual at hand): which lane's scale byte multiplies which
// products?  A = B = all e4m3 ones (0x38), so D[i][j] = sum over the four 32-element K blocks g of 32 * 2^(sa(i,g)-127) * 2^(sb(j,g)-127).
// Run 1: lane (row r = l&15, block g = l>>4) passes scale byte 127 + r + 4*g in byte 0 (opsel 0) for A, 127 for B.
//   hypothesis "a lane's scale applies to its own 32 K elements": D[i][j] = 32 * sum_g 2^(i + 4 g).
// Run 2: the same for B.   Run 3: byte 1 with opsel 1.
//   hipcc --offload-arch=gfx950 -O2 tools/mx_probe.hip -o tools/mx_probe && tools/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ void probe(float* out) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  i32x8 ones;
  for (int i = 0; i < 8; ++i) ones[i] = 0x38383838;            // e4m3 1.0 = 0x38
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int mine = 127 + r + 4 * g;
  if (MODE == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, ones, acc, 0, 0, 0, mine, 0, 0x7f7f7f7f);
  if (MODE == 1) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, ones, acc, 0, 0, 0, 0x7f7f7f7f, 0, mine);
  if (MODE == 2) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, ones, acc, 0, 0, 1, (mine << 8) | 0x7f, 0, 0x7f7f7f7f);
  // C/D layout of the 16x16 shapes: col = lane & 15, row = (lane >> 4) * 4 + reg
  for (int e = 0; e < 4; ++e) out[((l >> 4) * 4 + e) * 16 + (l & 15)] = acc[e];
}

// MEASURED (MI355X): run 4 prints block 0,0,1,1 for dwords 0-3 and 2,2,3,3 for dwords 4-7 of lane blocks g0 = 0..3, i.e. the operand's K order
// is  K = 64 * (dword / 4) + 16 * g + 4 * (dword % 4) + byte,  scale block b = K / 32 is read from lane r + 16 b (byte `opsel` of its scale
// register, the other three bytes ignored -- run 5).  A lane therefore does NOT scale "its own" 32 bytes: they straddle blocks g/2 and 2 + g/2.
// Run 4: WHICH K elements does a lane's scale cover?  A is zero except dword q0 of the lanes of block g0 (four e4m3 ones), B all ones, A's
// scale byte is 127 + g in lane block g: D[i][j] = 4 * 2^(block whose scale was applied).  "its own registers" <=> the answer is g0 for every q0.
// Run 5: the same with junk in the three unused bytes of the scale dword (opsel 0 must ignore them).
__global__ void probe_k(float* out, int g0, int q0, int junk) {
  const int l = threadIdx.x, g = l >> 4;
  i32x8 a, ones;
  for (int i = 0; i < 8; ++i) { ones[i] = 0x38383838; a[i] = (g == g0 && i == q0) ? 0x38383838 : 0; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int sc = 127 + g;
  if (junk) sc |= 0x9c8b7a00;
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, ones, acc, 0, 0, 0, sc, 0, 0x7f7f7f7f);
  for (int e = 0; e < 4; ++e) out[((l >> 4) * 4 + e) * 16 + (l & 15)] = acc[e];
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 4); (void)hipMemset(d, 0, 1024);
  float h[256];
  for (int junk = 0; junk < 2; ++junk) {
    printf("block whose scale is applied to dword q0 of lane block g0 (junk upper bytes: %d):\n", junk);
    for (int g0 = 0; g0 < 4; ++g0) {
      printf("  g0=%d:", g0);
      for (int q0 = 0; q0 < 8; ++q0) {
        probe_k<<<1, 64>>>(d, g0, q0, junk);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf(" %g/%g/%g", h[0], h[1], h[16 * 5 + 3]);
      }
      printf("\n");
    }
  }
  for (int mode = 0; mode < 3; ++mode) {
    if (mode == 0) probe<0><<<1, 64>>>(d); else if (mode == 1) probe<1><<<1, 64>>>(d); else probe<2><<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (first operand = builtin arg 0):\n  D[row][0]: ", mode);
    for (int i = 0; i < 16; ++i) printf("%g ", h[i * 16]);
    printf("\n  D[0][col]: ");
    for (int j = 0; j < 16; ++j) printf("%g ", h[j]);
    double want0 = 0; for (int g = 0; g < 4; ++g) want0 += 32.0 * std::pow(2.0, 4 * g);
    printf("\n  hypothesis value for index 0: %g, for index 1: %g\n", want0, 2 * want0);
  }
  return 0;
}
