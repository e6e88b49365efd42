// What does a CU's store path sustain?  The persistent GEMM's epilogue writes 128 KiB (bf16) / 256 KiB (f32) per output tile with
// 16-byte-per-lane full-line stores and was measured at ~13.6 B/clk/CU while all 256 workgroups are in their epilogues together
// (tools/ablate_persist.sh: the epilogue is 15-42 % of a launch).  Is that the CU's own limit or the chip's HBM write bandwidth?
//   grid = 1, 8, 64, 256 workgroups of 8 waves; each wave writes STRIPS x 1 KiB (64 lanes x 16 B, row-contiguous 128-B lines as the
//   epilogue does), plain / nontemporal, 16 B or 8 B per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o tools/store_probe && tools/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(u32x4* out, int strips, int reps, int64_t wg_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 v = {(uint32_t)lane, (uint32_t)wave, blockIdx.x, 7u};
  // a wave's strip = 8 rows x 128 B of a [rows][64 bf16] block (the epilogue's 2-byte output shape): lane -> row lane / 8, 16-byte piece lane % 8
  for (int r = 0; r < reps; ++r) {
    u32x4* base = out + (int64_t)blockIdx.x * wg_stride + ((int64_t)r * 8 + wave) * strips * 64;
    for (int s = 0; s < strips; ++s) {
      u32x4* p = base + s * 64 + lane;
      v.x += s;
      if (MODE == 0) *p = v;
      else if (MODE == 1) __builtin_nontemporal_store(v, p);
      else if (MODE == 2) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
      // the same 1 KiB per instruction laid out as a register-resident MFMA fragment would store it (no LDS transposition):
      // a strip PAIR = 16 rows x 128 B; lane -> row lane & 15, 16-lane group q = lane >> 4
      else if (MODE == 3) {                                  // 16 rows x 64 contiguous bytes per instruction: piece q of half (s & 1)
        u32x4* pp = base + (s >> 1) * 128 + (lane & 15) * 8 + (s & 1) * 4 + (lane >> 4);
        *pp = v;
      } else if (MODE == 4) {                                // 16 rows x four 16-byte pieces at stride 32 B per instruction
        u32x4* pp = base + (s >> 1) * 128 + (lane & 15) * 8 + (lane >> 4) * 2 + (s & 1);
        *pp = v;
      }
    }
  }
}

int main() {
  const int strips = 16, reps = 64;                        // 16 KiB per wave per rep = 128 KiB per workgroup per rep (one bf16 output tile)
  const int64_t wg_stride = (int64_t)reps * 8 * strips * 64;   // u32x4 units: every workgroup its own region
  const int max_grid = 256;
  u32x4* buf;
  if (hipMalloc(&buf, (size_t)max_grid * wg_stride * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[5] = {"plain", "nontemporal", "sc0 sc1", "16rows x 64B", "16rows x 4x16B/32"};
  for (int mode = 0; mode < 5; ++mode)
    for (int grid : {1, 8, 64, 256}) {
      float best = 1e30f;
      for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        if (mode == 0) store_kernel<0><<<grid, 512>>>(buf, strips, reps, wg_stride);
        if (mode == 1) store_kernel<1><<<grid, 512>>>(buf, strips, reps, wg_stride);
        if (mode == 2) store_kernel<2><<<grid, 512>>>(buf, strips, reps, wg_stride);
        if (mode == 3) store_kernel<3><<<grid, 512>>>(buf, strips, reps, wg_stride);
        if (mode == 4) store_kernel<4><<<grid, 512>>>(buf, strips, reps, wg_stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double bytes = (double)grid * reps * 8 * strips * 1024;
      printf("%-18s grid %3d: %8.1f us  %7.1f GB/s total  %6.1f GB/s per CU  (%.1f B/clk/CU at 2.0 GHz)\n", names[mode], grid, best * 1e3, bytes / best / 1e6,
             bytes / best / 1e6 / grid, bytes / best / 1e6 / grid / 2.0);
    }
  return 0;
}
