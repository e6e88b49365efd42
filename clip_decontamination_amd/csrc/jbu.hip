// SimFeatUp joint bilateral upsampler (reference simfeatup_dev/upsamplers.py:202-325).
#include "rowops.h"

namespace sg {

// Stand-alone adaptive convolution in FeatUp's NCHW calling convention
// (featup.adaptive_conv_cuda.AdaptiveConv.apply as used at upsamplers.py:274; semantics restated from
// adaptive_conv_py_simple, upsamplers.py:14-25):  out[b,c,y,x] = sum_{i,j<d} in[b,c,y+i,x+j] * filt[b,y,x,i,j]
__global__ __launch_bounds__(256) void adaptive_conv_nchw_kernel(const float* __restrict__ in, const float* __restrict__ filt,
                                                                 int C, int h, int w, int d, float* __restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z / C, c = blockIdx.z % C;
  if (x >= w || y >= h) return;
  const int wp = w + d - 1, hp = h + d - 1;
  const float* ip = in + ((int64_t)(b * C + c) * hp + y) * wp + x;
  const float* fp = filt + (((int64_t)b * h + y) * w + x) * d * d;
  float acc = 0.f;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) acc += ip[i * wp + j] * fp[i * d + j];
  out[((int64_t)(b * C + c) * h + y) * w + x] = acc;
}

}  // namespace sg

using namespace sg;

extern "C" int sg_adaptive_conv(const float* input, const float* filters, int B, int C, int h, int w, int d, float* out, sg_stream s) {
  SG_REQUIRE(input && filters && out, "sg_adaptive_conv: null pointer");
  SG_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0 && d > 0 && (int64_t)B * C < 65536 && cdiv(h, 4) < 65536, "sg_adaptive_conv: bad shape");
  hipLaunchKernelGGL(adaptive_conv_nchw_kernel, dim3((unsigned)cdiv(w, 64), (unsigned)cdiv(h, 4), (unsigned)(B * C)), dim3(256), 0,
                     as_stream(s), input, filters, C, h, w, d, out);
  SG_LAUNCH_CHECK();
  return SG_OK;
}
