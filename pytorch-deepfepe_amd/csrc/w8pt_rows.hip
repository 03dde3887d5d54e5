// dfepe_w8pt_rows_fwd -- the 8-point closing steps for an EXPLICIT design matrix: smallest right singular vector of rows [N,9],
// rank-2 (or (1,1,0)) step, de-normalisation.  Serves the dense-W form of the textbook solvers, where the reference
// left-multiplies the design matrix by W [N,N] before the SVD (deepFEPE/dsac_tools/utils_F.py:129-130,245-246): the rows of
// W XX are no longer Kronecker products of two points, so the 36-sum moment form of the fit kernels does not apply and the full
// symmetric 9x9 normal matrix (45 sums, fp64) is accumulated from the rows as given.  Every reference caller passes a diagonal W
// (a per-correspondence weight, served by dfepe_w8pt_fwd); this entry point closes the remaining surface, it is not a hot path.
// One 16-lane row per problem like the fit kernels, the same fp64 tridiagonal eigen-route (w8pt16_body.h: eig9_select).
#include "dfepe_common.h"
#include "w8pt16_body.h"

namespace {

constexpr int kProblemsPerBlock = 16;

__global__ void __launch_bounds__(256)
w8pt_rows_kernel(const float* __restrict__ rows, int B, int N, unsigned variant, const float* __restrict__ T1, const float* __restrict__ T2,
                 float* __restrict__ F_out) {
  const int prob = (int)blockIdx.x * kProblemsPerBlock + (int)(threadIdx.x >> 4);
  if (prob >= B) return;
  const int l = rg_lane();
  const float* src = rows + (size_t)prob * N * 9;
  double acc[45];
#pragma unroll
  for (int e = 0; e < 45; ++e) acc[e] = 0.0;
  for (int i = l; i < N; i += 16) {
    double r[9];
    bool fin = true;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const float v = src[(size_t)i * 9 + c];
      fin = fin && (fabsf(v) < 1e18f);
      r[c] = (double)v;
    }
    if (!fin) continue;  // a non-finite row is dropped, like a non-finite correspondence in the fit kernels
    int e = 0;
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
      for (int v = u; v < 9; ++v) {
        acc[e] = fma(r[u], r[v], acc[e]);
        ++e;
      }
  }
  // M[u][v] in every lane (45 row sums); lane i < 9 then keeps row i
  double Ar[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) Ar[j] = 0.0;
  double tr = 0.0;
  {
    int e = 0;
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
      for (int v = u; v < 9; ++v) {
        const double m = rg_sum(acc[e++]);
        if (u == v) tr += m;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
          if (j == v) Ar[j] = (l == u) ? m : Ar[j];
          if (j == u && u != v) Ar[j] = (l == v) ? m : Ar[j];
        }
      }
  }
  const double inv_tr = (tr > 0.0) ? rcp_nr<2>(tr) : 1.0;
#pragma unroll
  for (int j = 0; j < 9; ++j) Ar[j] *= inv_tr;
  // torch.svd(XX, some=True)[2][:, -1]: for N < 9 the smallest of the N non-null directions (as in the fit kernels)
  const int kth = (N >= 9) ? 0 : 9 - N;
  double f[9], z[9], td[9], te[8], hv[7], hb[7], lam;
  int twist;
  eig9_select(Ar, l, kth, f, z, twist, lam, td, te, hv, hb);
  double fn2 = 0.0, big = f[0];
#pragma unroll
  for (int c = 0; c < 9; ++c) fn2 = fma(f[c], f[c], fn2);
#pragma unroll
  for (int c = 1; c < 9; ++c)
    if (fabs(f[c]) > fabs(big)) big = f[c];
  const double fscale = ((big < 0.0) ? -1.0 : 1.0) * rsqrt_nr<2>(fn2);  // the library's sign gauge: largest component positive
#pragma unroll
  for (int c = 0; c < 9; ++c) f[c] *= fscale;
  double Fp[9];
  if (variant & DFEPE_W8PT_FORCE_110) {  // U diag(1,1,0) V^T (utils_F.py:148-149)
    float Ff[9], U3[9], S3[3], V3[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) Ff[c] = (float)f[c];
    svd3_fast(Ff, U3, S3, V3);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Fp[3 * r + c] = (double)U3[3 * r] * (double)V3[3 * c] + (double)U3[3 * r + 1] * (double)V3[3 * c + 1];
  } else {  // S3 -> 0 (utils_F.py:266-270)
    double u3[3], v3[3], s3;
    smallest_singular_triplet3(f, u3, v3, s3);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Fp[3 * r + c] = fma(-s3 * u3[r], v3[c], f[3 * r + c]);
  }
  double out[9];
  if (T1 != nullptr) {  // T2^T F' T1 with general 3x3 transforms (the caller's _normalize_XY)
    double t1[9], t2[9], tmp[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) { t1[c] = (double)T1[(size_t)prob * 9 + c]; t2[c] = (double)T2[(size_t)prob * 9 + c]; }
    mat3_mul_tn(t2, Fp, tmp);
    mat3_mul(tmp, t1, out);
  } else {
#pragma unroll
    for (int c = 0; c < 9; ++c) out[c] = Fp[c];
  }
  float mine = (float)out[0];
#pragma unroll
  for (int c = 1; c < 9; ++c) mine = (l == c) ? (float)out[c] : mine;
  if (l < 9) F_out[(size_t)prob * 9 + l] = mine;
}

}  // namespace

extern "C" int dfepe_w8pt_rows_fwd(const float* rows, int B, int N, unsigned flags, const float* T1, const float* T2, float* F_out,
                                   void* stream) {
  if (B < 0 || N <= 0) return DFEPE_ERR_INVALID_ARG;
  if (flags & ~DFEPE_W8PT_FORCE_110) return DFEPE_ERR_INVALID_ARG;
  if (B == 0) return DFEPE_OK;
  if (!rows || !F_out || ((T1 == nullptr) != (T2 == nullptr))) return DFEPE_ERR_INVALID_ARG;
  const dim3 grid((B + kProblemsPerBlock - 1) / kProblemsPerBlock), block(256);
  hipLaunchKernelGGL(w8pt_rows_kernel, grid, block, 0, static_cast<hipStream_t>(stream), rows, B, N, flags, T1, T2, F_out);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
