// Micro-benchmark: HBM store rate of a GEMM-epilogue-shaped write pattern vs a linear stream (tuning aid, not product code).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0: block b writes tile (b / tn, b % tn) of a row-major [M][N] bf16 matrix, 256x256 per block, each wave-instruction = 8 rows x 128 B
// mode 1: same bytes, but the matrix is stored tile-blocked: block b writes 128 KiB contiguous
// mode 2: grid-stride linear stream
__global__ __launch_bounds__(512) void wk(uint4* out, int M, int N, int mode, int tiles_n) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  uint4 v = make_uint4(tid, blockIdx.x, 1, 2);
  if (mode == 2) {
    const size_t total = (size_t)M * N * 2 / 16;
    for (size_t i = (size_t)blockIdx.x * 512 + tid; i < total; i += (size_t)gridDim.x * 512) out[i] = v;
    return;
  }
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int g = wave >> 2, wi = wave & 3;
  for (int strip = 0; strip < 8; ++strip)
    for (int pass = 0; pass < 2; ++pass) {
      const int r = 128 * g + strip * 16 + pass * 8 + (lane >> 3);     // row in tile
      const int c = 64 * wi + (lane & 7) * 8;                          // col in tile (bf16 elements)
      size_t off;
      if (mode == 0) off = ((size_t)(tm * 256 + r) * N + tn * 256 + c) * 2;
      else off = ((size_t)blockIdx.x * 65536 + (size_t)r * 256 + c) * 2;
      if (tm * 256 + r < M) out[off / 16] = v;
    }
}

int main() {
  const int M = 43840, N = 3072;
  const int tm = (M + 255) / 256, tn = N / 256;
  uint4* buf; CK(hipMalloc(&buf, (size_t)tm * 256 * N * 2 + (1 << 20)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(wk, dim3(mode == 2 ? 2048 : tm * tn), dim3(512), 0, 0, buf, M, N, mode, tn);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("mode %d: %.1f us per pass, %.2f TB/s\n", mode, ms * 100, (double)M * N * 2 / (ms / 10 * 1e-3) / 1e12);
    }
  }
  return 0;
}
