// Tile extraction + SegDataPreProcessor normalisation + zero padding to a patch multiple + im2col,
// in one pass over the scene (reference segmentor.py:64-67 mean/std, :418-431 crop + F.pad,
// open_clip/transformer.py:560 conv1 with kernel = stride = P, no bias).
// Output: patch matrix [n_tiles * gh * gw, Kpad], column k = c*P*P + py*P + px (the flatten order of
// conv1.weight[D,3,P,P]); columns >= 3*P*P are zero so K is a multiple of the GEMM's K tile.
#include "rowops.h"

namespace sg {

__constant__ float c_mean[3] = {122.771f, 116.746f, 104.094f};
__constant__ float c_std[3] = {68.501f, 66.632f, 70.323f};

// PC = compile-time patch size (0 = runtime): the three index divisions per element become multiply-shifts for the shipped 14 / 16 / 32
template <typename OutT, int PC>
__global__ __launch_bounds__(256) void patchify_kernel(sg_tile_batch t, int P_rt, OutT* __restrict__ out, int Kpad_rt) {
  // one workgroup per (tile, patch row); threads sweep (patch col, k) with k fastest
  const int P = PC ? PC : P_rt;
  const int Kpad = PC ? (3 * PC * PC + 63) / 64 * 64 : Kpad_rt;
  const int tile = blockIdx.y, py_idx = blockIdx.x;
  const int y1 = t.windows[tile * 4 + 0], x1 = t.windows[tile * 4 + 2];
  const int P2 = P * P, K = 3 * P2;
  const int64_t img_off = t.scene_index ? (int64_t)t.scene_index[tile] * t.scene_stride : 0;
  const int64_t row0 = ((int64_t)tile * t.grid_h + py_idx) * t.grid_w;
  const int total = t.grid_w * Kpad;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int pxi = i / Kpad, k = i % Kpad;
    float v = 0.f;
    if (k < K) {
      const int c = k / P2, rem = k % P2, dy = rem / P, dx = rem % P;
      const int ty = py_idx * P + dy - t.pad_t, tx = pxi * P + dx - t.pad_l;       // coordinates inside the window
      if (ty >= 0 && ty < t.tile_h && tx >= 0 && tx < t.tile_w) {
        const int sy = y1 + ty, sx = x1 + tx;
        if (t.format == SG_IMG_F32_NCHW) {
          v = reinterpret_cast<const float*>(t.scene)[img_off + ((int64_t)c * t.scene_h + sy) * t.scene_w + sx];
        } else {
          const uint8_t u = reinterpret_cast<const uint8_t*>(t.scene)[img_off + ((int64_t)sy * t.scene_w + sx) * 3 + c];
          v = ((float)u - c_mean[c]) / c_std[c];
        }
      }
    }
    st_elem<OutT>(out + (row0 + pxi) * Kpad, k, v);
  }
}

int patchify(const sg_tile_batch& t, int P, void* out, int Kpad, int out_bf16, hipStream_t s) {
  SG_REQUIRE(t.n_tiles > 0 && t.grid_h > 0 && t.grid_w > 0, "patchify: empty batch");
  SG_REQUIRE(Kpad >= 3 * P * P, "patchify: Kpad too small");
  SG_REQUIRE(t.grid_h * P >= t.tile_h + t.pad_t && t.grid_w * P >= t.tile_w + t.pad_l, "patchify: grid does not cover the padded tile");
  SG_REQUIRE(t.n_tiles < 65536, "patchify: too many tiles in one launch");
  dim3 grid((unsigned)t.grid_h, (unsigned)t.n_tiles);
  const bool std_pad = Kpad == (3 * P * P + 63) / 64 * 64;
  if (out_bf16 == HK_F16X2) {                            // two-plane f16 (Kpad % 8 == 0: row starts sit on storage-group boundaries)
    SG_REQUIRE(Kpad % 8 == 0, "patchify: two-plane f16 rows are multiples of 8 elements");
    if (std_pad && P == 14) hipLaunchKernelGGL((patchify_kernel<h2_t, 14>), grid, dim3(256), 0, s, t, P, (h2_t*)out, Kpad);
    else if (std_pad && P == 16) hipLaunchKernelGGL((patchify_kernel<h2_t, 16>), grid, dim3(256), 0, s, t, P, (h2_t*)out, Kpad);
    else hipLaunchKernelGGL((patchify_kernel<h2_t, 0>), grid, dim3(256), 0, s, t, P, (h2_t*)out, Kpad);
  }
  else if (out_bf16 == HK_F16) {
    if (std_pad && P == 14) hipLaunchKernelGGL((patchify_kernel<f16_t, 14>), grid, dim3(256), 0, s, t, P, (f16_t*)out, Kpad);
    else if (std_pad && P == 16) hipLaunchKernelGGL((patchify_kernel<f16_t, 16>), grid, dim3(256), 0, s, t, P, (f16_t*)out, Kpad);
    else hipLaunchKernelGGL((patchify_kernel<f16_t, 0>), grid, dim3(256), 0, s, t, P, (f16_t*)out, Kpad);
  }
  else if (out_bf16 && std_pad && P == 14) hipLaunchKernelGGL((patchify_kernel<bf16_t, 14>), grid, dim3(256), 0, s, t, P, (bf16_t*)out, Kpad);
  else if (out_bf16 && std_pad && P == 16) hipLaunchKernelGGL((patchify_kernel<bf16_t, 16>), grid, dim3(256), 0, s, t, P, (bf16_t*)out, Kpad);
  else if (out_bf16 && std_pad && P == 32) hipLaunchKernelGGL((patchify_kernel<bf16_t, 32>), grid, dim3(256), 0, s, t, P, (bf16_t*)out, Kpad);
  else if (out_bf16) hipLaunchKernelGGL((patchify_kernel<bf16_t, 0>), grid, dim3(256), 0, s, t, P, (bf16_t*)out, Kpad);
  else hipLaunchKernelGGL((patchify_kernel<float, 0>), grid, dim3(256), 0, s, t, P, (float*)out, Kpad);
  SG_LAUNCH_CHECK();
  return SG_OK;
}

}  // namespace sg
