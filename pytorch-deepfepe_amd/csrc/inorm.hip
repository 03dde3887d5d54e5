// inorm_lrelu — InstanceNorm1d(affine) + LeakyReLU over rows of N contiguous floats, forward and backward.
//
// "Next" row f-1 of SURVEY.md §8: the per-correspondence weight estimator (deepFEPE/models/ErrorEstimators.py:47-64) is
// a chain of Conv1d(k=1) -> InstanceNorm1d(affine) -> LeakyReLU.  The 1x1 convolutions are plain GEMMs (left to
// rocBLAS/hipBLASLt through torch.mm on a channel-major [C, B*N] layout); this kernel is the part in between, which in
// stock PyTorch costs ~5 HBM passes per layer (batch-norm statistics, normalise, affine, activation).  Here each
// activation is read once and written once.
//
// Layout: Y[(c * R + r) * N + n], c < C channels, r < R rows per channel (= pairs), N points; a row is one
// (pair, channel) instance.  HBM-bound by construction: 8 B per element forward, 12 B per element backward.
// Mapping: 16 lanes per row (4 rows per wavefront, rows are contiguous so a wavefront touches 4N contiguous floats),
// float4 accesses, row statistics by DPP reductions inside the 16-lane row (no LDS, no cross-lane traffic otherwise).
#include "dfepe_common.h"

namespace {

constexpr int kGroup = 16;   // lanes per row
constexpr int kMaxVec = 8;   // float4 per lane kept in registers: N <= 16 * 8 * 4 = 512

__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16-lane DPP row, result in every lane of the row
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  v += dpp_f32<0x140>(v);
  return v;
}

template <int NV>
__global__ void __launch_bounds__(256)
inorm_lrelu_fwd_kernel(const float* __restrict__ Y, const float* __restrict__ gamma, const float* __restrict__ beta, int R,
                       long rows, int N, float eps, float slope, float* __restrict__ A, float* __restrict__ stats) {
  const int sub = threadIdx.x & (kGroup - 1);
  const long row = (long)blockIdx.x * (blockDim.x / kGroup) + (threadIdx.x / kGroup);
  const bool live = row < rows;
  const long rr = live ? row : rows - 1;  // keep the lanes of dead rows in the DPP reductions with valid addresses
  const int nvec = N >> 2;
  const float4* src = reinterpret_cast<const float4*>(Y + rr * N);
  float4 v[NV];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int idx = sub + kGroup * k;
    v[k] = (idx < nvec) ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  const float mean = row16_sum(s) / (float)N;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int idx = sub + kGroup * k;
    if (idx < nvec) {
      const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float var = row16_sum(q) / (float)N;  // biased variance, like F.instance_norm
  const float rstd = 1.0f / sqrtf(var + eps);
  const int ch = (int)(rr / R);
  const float g = gamma[ch] * rstd, bta = beta[ch];
  float4* dst = reinterpret_cast<float4*>(A + rr * N);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int idx = sub + kGroup * k;
    if (live && idx < nvec) {
      float4 o;
      o.x = fmaf(v[k].x - mean, g, bta); o.y = fmaf(v[k].y - mean, g, bta);
      o.z = fmaf(v[k].z - mean, g, bta); o.w = fmaf(v[k].w - mean, g, bta);
      o.x = (o.x > 0.f) ? o.x : o.x * slope; o.y = (o.y > 0.f) ? o.y : o.y * slope;
      o.z = (o.z > 0.f) ? o.z : o.z * slope; o.w = (o.w > 0.f) ? o.w : o.w * slope;
      dst[idx] = o;
    }
  }
  if (live && sub == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

template <int NV>
__global__ void __launch_bounds__(256)
inorm_lrelu_bwd_kernel(const float* __restrict__ Y, const float* __restrict__ gA, const float* __restrict__ gamma,
                       const float* __restrict__ beta, const float* __restrict__ stats, int R, long rows, int N, float slope,
                       float* __restrict__ gY, float* __restrict__ row_ggamma, float* __restrict__ row_gbeta) {
  const int sub = threadIdx.x & (kGroup - 1);
  const long row = (long)blockIdx.x * (blockDim.x / kGroup) + (threadIdx.x / kGroup);
  const bool live = row < rows;
  const long rr = live ? row : rows - 1;
  const int nvec = N >> 2;
  const float mean = stats[2 * rr], rstd = stats[2 * rr + 1];
  const int ch = (int)(rr / R);
  const float gm = gamma[ch], bta = beta[ch];
  const float4* ys = reinterpret_cast<const float4*>(Y + rr * N);
  const float4* gs = reinterpret_cast<const float4*>(gA + rr * N);
  float4 xh[NV], gz[NV];
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int idx = sub + kGroup * k;
    if (idx < nvec) {
      const float4 y = ys[idx], g = gs[idx];
      xh[k].x = (y.x - mean) * rstd; xh[k].y = (y.y - mean) * rstd; xh[k].z = (y.z - mean) * rstd; xh[k].w = (y.w - mean) * rstd;
      gz[k].x = (fmaf(xh[k].x, gm, bta) > 0.f) ? g.x : g.x * slope;
      gz[k].y = (fmaf(xh[k].y, gm, bta) > 0.f) ? g.y : g.y * slope;
      gz[k].z = (fmaf(xh[k].z, gm, bta) > 0.f) ? g.z : g.z * slope;
      gz[k].w = (fmaf(xh[k].w, gm, bta) > 0.f) ? g.w : g.w * slope;
      s1 += (gz[k].x + gz[k].y) + (gz[k].z + gz[k].w);
      s2 += (gz[k].x * xh[k].x + gz[k].y * xh[k].y) + (gz[k].z * xh[k].z + gz[k].w * xh[k].w);
    } else {
      xh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      gz[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  s1 = row16_sum(s1);
  s2 = row16_sum(s2);
  const float m1 = s1 / (float)N, m2 = s2 / (float)N, sc = gm * rstd;
  float4* dst = reinterpret_cast<float4*>(gY + rr * N);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int idx = sub + kGroup * k;
    if (live && idx < nvec) {
      float4 o;
      o.x = sc * (gz[k].x - m1 - xh[k].x * m2); o.y = sc * (gz[k].y - m1 - xh[k].y * m2);
      o.z = sc * (gz[k].z - m1 - xh[k].z * m2); o.w = sc * (gz[k].w - m1 - xh[k].w * m2);
      dst[idx] = o;
    }
  }
  if (live && sub == 0) {
    row_gbeta[row] = s1;    // d/d(beta[ch])  contribution of this row (summed over the R rows of a channel by the caller)
    row_ggamma[row] = s2;   // d/d(gamma[ch])
  }
}

int pick_nv(int N) {
  if (N <= 0 || (N & 3)) return 0;
  const int need = ((N >> 2) + kGroup - 1) / kGroup;
  if (need <= 2) return 2;
  if (need <= 4) return 4;
  if (need <= kMaxVec) return 8;
  return 0;
}

}  // namespace

extern "C" int dfepe_inorm_lrelu_fwd(const float* Y, const float* gamma, const float* beta, int C, int R, int N, float eps,
                                     float slope, float* A, float* stats, void* stream) {
  if (C <= 0 || R < 0 || N <= 0) return DFEPE_ERR_INVALID_ARG;
  if (R == 0) return DFEPE_OK;
  if (!Y || !gamma || !beta || !A || !stats) return DFEPE_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(A)) & 15u) return DFEPE_ERR_INVALID_ARG;
  const int nv = pick_nv(N);
  if (nv == 0) return DFEPE_ERR_UNSUPPORTED;  // N must be a multiple of 4 and <= 512
  const long rows = (long)C * R;
  const dim3 block(256), grid((unsigned)((rows + 15) / 16));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nv == 2) hipLaunchKernelGGL(inorm_lrelu_fwd_kernel<2>, grid, block, 0, st, Y, gamma, beta, R, rows, N, eps, slope, A, stats);
  else if (nv == 4) hipLaunchKernelGGL(inorm_lrelu_fwd_kernel<4>, grid, block, 0, st, Y, gamma, beta, R, rows, N, eps, slope, A, stats);
  else hipLaunchKernelGGL(inorm_lrelu_fwd_kernel<8>, grid, block, 0, st, Y, gamma, beta, R, rows, N, eps, slope, A, stats);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_inorm_lrelu_bwd(const float* Y, const float* gA, const float* gamma, const float* beta, const float* stats,
                                     int C, int R, int N, float slope, float* gY, float* row_ggamma, float* row_gbeta,
                                     void* stream) {
  if (C <= 0 || R < 0 || N <= 0) return DFEPE_ERR_INVALID_ARG;
  if (R == 0) return DFEPE_OK;
  if (!Y || !gA || !gamma || !beta || !stats || !gY || !row_ggamma || !row_gbeta) return DFEPE_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(gA) | reinterpret_cast<uintptr_t>(gY)) & 15u)
    return DFEPE_ERR_INVALID_ARG;
  const int nv = pick_nv(N);
  if (nv == 0) return DFEPE_ERR_UNSUPPORTED;
  const long rows = (long)C * R;
  const dim3 block(256), grid((unsigned)((rows + 15) / 16));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nv == 2) hipLaunchKernelGGL(inorm_lrelu_bwd_kernel<2>, grid, block, 0, st, Y, gA, gamma, beta, stats, R, rows, N, slope, gY, row_ggamma, row_gbeta);
  else if (nv == 4) hipLaunchKernelGGL(inorm_lrelu_bwd_kernel<4>, grid, block, 0, st, Y, gA, gamma, beta, stats, R, rows, N, slope, gY, row_ggamma, row_gbeta);
  else hipLaunchKernelGGL(inorm_lrelu_bwd_kernel<8>, grid, block, 0, st, Y, gA, gamma, beta, stats, R, rows, N, slope, gY, row_ggamma, row_gbeta);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
