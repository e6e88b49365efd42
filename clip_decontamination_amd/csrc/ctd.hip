// Cluster-Then-Debias on the device (reference CTD.py as driven by segmentor.py:339-365).
//
// The reference moves the patch tokens of every tile to the CPU, runs scikit-learn DBSCAN (brute-force radius query in
// float64 on float32 points, eps 1.1, min_samples 11) and adds a per-cluster multiple of the CLS token back on the GPU.
// Here the whole step stays in HBM, all tiles of a launch in parallel:
//   ctd_points      p = unit(x / (|x| + 1.1))                              CTD.py:62-63,212-219,104
//   gemm_f32        G = p p^T per tile (f32 MFMA)                          squared distance = |p_i|^2 + |p_j|^2 - 2 G_ij
//   ctd_adjacency   eps-neighbourhood bit matrix + core flags; pairs within 1e-4 of eps^2 are re-evaluated with the float64
//                   Gram expansion scikit-learn itself uses, so the decision matches it away from exact ties
//   ctd_components  one workgroup per tile: min-label propagation + pointer jumping over core points in LDS, clusters numbered
//                   by their smallest core index (= scikit-learn's discovery order), border points join the lowest-numbered
//                   neighbouring cluster, everything else is noise (-1)
//   ctd_proto_sim   per cluster: mean token (index order, deterministic), cos-sim with the CLS token (eps 1.1)   CTD.py:344-358
//   ctd_apply       x += sim_k * (factor * cls)                                                                  CTD.py:360-361
#include "rowops.h"

namespace sg {

constexpr int CTD_MAX_POINTS = 8192;        // segmentor.py:346 'max_points': larger grids skip the step in the reference
constexpr int CTD_MAX_C = 2048;

__global__ __launch_bounds__(256) void ctd_points_kernel(const float* __restrict__ x, int64_t rows, int C, float* __restrict__ p,
                                                         float* __restrict__ nrm2) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  float* pr = p + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c] * xr[c];
  const float d1 = sqrtf(wave_sum(s)) + 1.1f;
  float s2 = 0.f;
  for (int c = lane; c < C; c += 64) { const float u = xr[c] / d1; pr[c] = u; s2 += u * u; }
  const float d2 = sqrtf(wave_sum(s2)) + 1e-8f;
  float s3 = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = pr[c] / d2; pr[c] = v; s3 += v * v; }   // each lane re-reads its own stores
  s3 = wave_sum(s3);
  if (lane == 0) nrm2[row] = s3;
}

__global__ __launch_bounds__(256) void ctd_adjacency_kernel(const float* __restrict__ G, const float* __restrict__ p,
                                                            const float* __restrict__ nrm2, int n, int C, double eps2, int min_samples,
                                                            int W, unsigned long long* __restrict__ adj, uint8_t* __restrict__ core) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const float* Gi = G + ((int64_t)b * n + i) * n;
  const float* pi = p + ((int64_t)b * n + i) * C;
  const float ni = nrm2[(int64_t)b * n + i];
  const float eps2f = (float)eps2;
  int count = 0;
  for (int w = 0; w < W; ++w) {
    const int j = w * 64 + lane;
    bool in = false;
    if (j < n) {
      const float d2 = ni + nrm2[(int64_t)b * n + j] - 2.0f * Gi[j];
      if (fabsf(d2 - eps2f) < 1e-4f) {                      // near the radius: the float64 Gram expansion scikit-learn evaluates
        const float* pj = p + ((int64_t)b * n + j) * C;
        double dot = 0.0, a2 = 0.0, b2 = 0.0;
        for (int c = 0; c < C; ++c) { const double a = pi[c], bb = pj[c]; dot += a * bb; a2 += a * a; b2 += bb * bb; }
        in = (a2 + b2 - 2.0 * dot) <= eps2;
      } else {
        in = d2 <= eps2f;
      }
    }
    const unsigned long long m = __ballot(in);
    if (lane == 0) adj[((int64_t)b * n + i) * W + w] = m;
    count += __popcll(m);
  }
  if (lane == 0) core[(int64_t)b * n + i] = count >= min_samples ? 1 : 0;
}

// one workgroup per tile; LDS: lab[n] | cid[n] | cmask[W] | scan[1024]
__global__ __launch_bounds__(1024) void ctd_components_kernel(const unsigned long long* __restrict__ adj, const uint8_t* __restrict__ core,
                                                              int n, int W, int32_t* __restrict__ labels, int32_t* __restrict__ n_clusters) {
  extern __shared__ __attribute__((aligned(16))) char ctd_sm[];
  int* lab = reinterpret_cast<int*>(ctd_sm);
  int* cid = lab + n;
  unsigned long long* cmask = reinterpret_cast<unsigned long long*>(cid + n + ((2 * n) & 1));   // 8-byte aligned
  int* scan = reinterpret_cast<int*>(cmask + W);
  __shared__ int changed;
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint8_t* cb = core + (int64_t)b * n;
  const unsigned long long* ab = adj + (int64_t)b * n * W;
  for (int w = tid; w < W; w += 1024) {
    unsigned long long m = 0;
    for (int k = 0; k < 64; ++k) { const int j = w * 64 + k; if (j < n && cb[j]) m |= 1ull << k; }
    cmask[w] = m;
  }
  for (int i = tid; i < n; i += 1024) lab[i] = cb[i] ? i : 0x7fffffff;
  __syncthreads();
  for (int iter = 0; iter < n; ++iter) {                    // every wave leaves together: `changed` is read after a barrier
    if (tid == 0) changed = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
      if (!cb[i]) continue;
      int m = lab[i];
      const unsigned long long* row = ab + (int64_t)i * W;
      for (int w = 0; w < W; ++w) {
        unsigned long long bits = row[w] & cmask[w];
        while (bits) {
          const int j = w * 64 + __builtin_ctzll(bits);
          bits &= bits - 1;
          const int lj = lab[j];
          m = lj < m ? lj : m;
        }
      }
      const int mm = lab[m];                                // pointer jump (m is a core index)
      m = mm < m ? mm : m;
      if (m < lab[i]) { lab[i] = m; changed = 1; }
    }
    __syncthreads();
    const int again = changed;
    __syncthreads();
    if (!again) break;
  }
  // clusters numbered by their smallest core index: exclusive count of roots
  const int per = (n + 1023) / 1024;
  int local = 0;
  for (int k = 0; k < per; ++k) { const int i = tid * per + k; if (i < n && cb[i] && lab[i] == i) ++local; }
  scan[tid] = local;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < 1024; ++t) { const int v = scan[t]; scan[t] = run; run += v; }
    n_clusters[b] = run;
  }
  __syncthreads();
  int next = scan[tid];
  for (int k = 0; k < per; ++k) { const int i = tid * per + k; if (i < n) cid[i] = (cb[i] && lab[i] == i) ? next++ : -1; }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    int out;
    if (cb[i]) out = cid[lab[i]];
    else {
      int m = 0x7fffffff;
      const unsigned long long* row = ab + (int64_t)i * W;
      for (int w = 0; w < W; ++w) {
        unsigned long long bits = row[w] & cmask[w];
        while (bits) {
          const int j = w * 64 + __builtin_ctzll(bits);
          bits &= bits - 1;
          const int c = cid[lab[j]];
          m = c < m ? c : m;
        }
      }
      out = m == 0x7fffffff ? -1 : m;
    }
    labels[(int64_t)b * n + i] = out;
  }
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// grid (n cluster slots, B); block 256.  sims[b][k] = clamp(cos_eps(mean of cluster k, cls_b), -1, 1)
__global__ __launch_bounds__(256) void ctd_proto_sim_kernel(const float* __restrict__ x, const float* __restrict__ cls,
                                                            const int32_t* __restrict__ labels, const int32_t* __restrict__ n_clusters, int n,
                                                            int C, float* __restrict__ sims) {
  __shared__ float red[4];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  if (k >= n_clusters[b]) return;
  const int32_t* lb = labels + (int64_t)b * n;
  const float* xb = x + (int64_t)b * n * C;
  const float* cv = cls + (int64_t)b * C;
  float pr[CTD_MAX_C / 256];
  float count = 0.f;
  for (int i = 0; i < n; ++i) count += lb[i] == k ? 1.f : 0.f;
  const float cnt = fmaxf(count, 1.0f);
  float pp = 0.f, cc = 0.f;
#pragma unroll
  for (int t = 0; t < CTD_MAX_C / 256; ++t) {
    const int c = t * 256 + tid;
    float acc = 0.f;
    if (c < C) {
      for (int i = 0; i < n; ++i) if (lb[i] == k) acc += xb[(int64_t)i * C + c];          // index order, as index_add_ on the CPU
      acc /= cnt;
      pp += acc * acc;
      cc += cv[c] * cv[c];
    }
    pr[t] = acc;
  }
  const float np_ = sqrtf(block_sum_256(pp, red)) + 1.1f;
  const float nc_ = sqrtf(block_sum_256(cc, red)) + 1.1f;
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < CTD_MAX_C / 256; ++t) {
    const int c = t * 256 + tid;
    if (c < C) dot += (pr[t] / np_) * (cv[c] / nc_);
  }
  dot = block_sum_256(dot, red);
  if (tid == 0) sims[(int64_t)b * n + k] = fminf(fmaxf(dot, -1.0f), 1.0f);
}

__global__ void ctd_apply_kernel(float* __restrict__ x, const float* __restrict__ cls, const int32_t* __restrict__ labels,
                                 const float* __restrict__ sims, int n, int C, float factor) {
  const int b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * C) return;
  const int tok = (int)(i / C), c = (int)(i % C);
  const int l = labels[(int64_t)b * n + tok];
  if (l < 0) return;
  x[(int64_t)b * n * C + i] += sims[(int64_t)b * n + l] * (factor * cls[(int64_t)b * C + c]);
}

// cls_hat = cls / ||cls|| (segmentor.py:310), one wave per row
__global__ __launch_bounds__(64) void ctd_unit_cls_kernel(const float* __restrict__ cls, int C, float* __restrict__ out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) ss += cls[(int64_t)b * C + c] * cls[(int64_t)b * C + c];
  const float nrm = sqrtf(wave_sum(ss));
  for (int c = lane; c < C; c += 64) out[(int64_t)b * C + c] = cls[(int64_t)b * C + c] / nrm;
}

__global__ void ctd_fill_kernel(int32_t* p, int64_t count, int32_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) p[i] = v;
}

struct CtdPlan { float* p; float* G; float* nrm2; float* clsu; unsigned long long* adj; uint8_t* core; int32_t* labels; int32_t* ncl; float* sims; size_t bytes; };
static CtdPlan ctd_plan(void* base, int B, int n, int C) {
  CtdPlan q{};
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = base ? static_cast<char*>(base) + off : nullptr; off += align_up(bytes, 256); return r; };
  const int W = (n + 63) / 64;
  q.p = static_cast<float*>(take((size_t)B * n * C * 4));
  q.G = static_cast<float*>(take((size_t)B * n * n * 4));
  q.nrm2 = static_cast<float*>(take((size_t)B * n * 4));
  q.clsu = static_cast<float*>(take((size_t)B * C * 4));
  q.adj = static_cast<unsigned long long*>(take((size_t)B * n * W * 8));
  q.core = static_cast<uint8_t*>(take((size_t)B * n));
  q.labels = static_cast<int32_t*>(take((size_t)B * n * 4));
  q.ncl = static_cast<int32_t*>(take((size_t)B * 4));
  q.sims = static_cast<float*>(take((size_t)B * n * 4));
  q.bytes = off;
  return q;
}

}  // namespace sg

using namespace sg;

extern "C" size_t sg_ctd_scratch_bytes(int B, int n, int C) {
  if (B <= 0 || n <= 0 || C <= 0 || n > CTD_MAX_POINTS) return 256;
  return ctd_plan(nullptr, B, n, C).bytes + 256;
}

extern "C" int sg_ctd_debias(float* tokens, const float* cls, int B, int n, int C, double eps, int min_samples, float factor,
                             int normalize_cls, int32_t* labels_out, void* scratch, size_t scratch_bytes, sg_stream st) {
  SG_REQUIRE(tokens && cls && scratch && B > 0 && n > 0 && C > 0 && eps > 0 && min_samples > 0, "sg_ctd_debias: bad arguments");
  SG_REQUIRE(C <= CTD_MAX_C && C % 4 == 0, "sg_ctd_debias: C = %d must be a multiple of 4 and <= %d", C, CTD_MAX_C);
  hipStream_t s = as_stream(st);
  if (n > CTD_MAX_POINTS) {                                   // the reference skips the clustering (labels None, CTD.py:184-189)
    if (labels_out) {
      hipLaunchKernelGGL(ctd_fill_kernel, dim3((unsigned)cdiv((int64_t)B * n, 256)), dim3(256), 0, s, labels_out, (int64_t)B * n, -1);
      SG_LAUNCH_CHECK();
    }
    return SG_OK;
  }
  const CtdPlan q = ctd_plan(scratch, B, n, C);
  SG_REQUIRE(q.bytes <= scratch_bytes, "sg_ctd_debias: scratch %zu < required %zu", scratch_bytes, q.bytes);
  const int W = (n + 63) / 64;
  if (normalize_cls) {
    hipLaunchKernelGGL(ctd_unit_cls_kernel, dim3(B), dim3(64), 0, s, cls, C, q.clsu);
    SG_LAUNCH_CHECK();
    cls = q.clsu;
  }
  hipLaunchKernelGGL(ctd_points_kernel, dim3((unsigned)cdiv((int64_t)B * n, 4)), dim3(256), 0, s, tokens, (int64_t)B * n, C, q.p, q.nrm2);
  SG_LAUNCH_CHECK();
  GemmF32Args g{};
  g.A = q.p; g.lda = C; g.sAo = (int64_t)n * C; g.B = q.p; g.sbk = 1; g.sbn = C; g.sBo = (int64_t)n * C;
  g.C = q.G; g.ldc = n; g.sCo = (int64_t)n * n; g.M = n; g.N = n; g.K = C; g.batch = B; g.inner = 1; g.act = 0; g.alpha = 1.f;
  SG_TRY(gemm_f32(g, s));
  hipLaunchKernelGGL(ctd_adjacency_kernel, dim3((unsigned)cdiv(n, 4), (unsigned)B), dim3(256), 0, s, q.G, q.p, q.nrm2, n, C, eps * eps,
                     min_samples, W, q.adj, q.core);
  SG_LAUNCH_CHECK();
  const size_t lds = (size_t)(2 * n + ((2 * n) & 1)) * 4 + (size_t)W * 8 + 1024 * 4;
  SG_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(ctd_components_kernel), 96 * 1024));
  hipLaunchKernelGGL(ctd_components_kernel, dim3(B), dim3(1024), lds, s, q.adj, q.core, n, W, q.labels, q.ncl);
  SG_LAUNCH_CHECK();
  hipLaunchKernelGGL(ctd_proto_sim_kernel, dim3(n, B), dim3(256), 0, s, tokens, cls, q.labels, q.ncl, n, C, q.sims);
  SG_LAUNCH_CHECK();
  hipLaunchKernelGGL(ctd_apply_kernel, dim3((unsigned)cdiv((int64_t)n * C, 256), (unsigned)B), dim3(256), 0, s, tokens, cls, q.labels, q.sims, n, C,
                     factor);
  SG_LAUNCH_CHECK();
  if (labels_out) SG_HIP(hipMemcpyAsync(labels_out, q.labels, (size_t)B * n * 4, hipMemcpyDeviceToDevice, s));
  return SG_OK;
}
