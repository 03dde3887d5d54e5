// Stand-alone geometry entry points of the dsac_tools API: epipolar residual / metrics, small pose helpers,
// cheirality-checked pose selection.  These sit on either side of the solver.
#include "dfepe_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// epipolar residual (utils_F.py:400-413), one wavefront per pair
// ------------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ void __launch_bounds__(256)
epi_residual_kernel(const float* __restrict__ pts1, const float* __restrict__ pts2, const float* __restrict__ F, int B,
                    int N, float clamp_at, float* __restrict__ out, const float* __restrict__ g_out,
                    float* __restrict__ g_F) {
  const int lane = threadIdx.x & 63;
  const size_t pair = (size_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform by construction
  if (pair >= (size_t)B) return;
  float o[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) o[c] = F[pair * 9 + c];
  double go[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) go[c] = 0.0;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
  for (int i = lane; i < N; i += WAVE) {
    const Pt p = global_point<false>(pts1, pts2, pair, i, N, 0.f, 0.f);
    if (!BWD) {
      const float l1x = fmaf(p.x2, o[0], fmaf(p.y2, o[3], p.z2 * o[6]));
      const float l1y = fmaf(p.x2, o[1], fmaf(p.y2, o[4], p.z2 * o[7]));
      const float l1z = fmaf(p.x2, o[2], fmaf(p.y2, o[5], p.z2 * o[8]));
      const float l2x = fmaf(p.x1, o[0], fmaf(p.y1, o[1], p.z1 * o[2]));
      const float l2y = fmaf(p.x1, o[3], fmaf(p.y1, o[4], p.z1 * o[5]));
      const float dd = fmaf(p.x1, l1x, fmaf(p.y1, l1y, p.z1 * l1z));
      const float n1 = sqrtf(fmaf(l1x, l1x, l1y * l1y)) + 1e-6f;
      const float n2 = sqrtf(fmaf(l2x, l2x, l2y * l2y)) + 1e-6f;
      out[pair * N + i] = fminf(fabsf(dd) * (1.0f / n1 + 1.0f / n2), clamp_at);
    } else {
      const double x1[3] = {p.x1, p.y1, p.z1}, x2[3] = {p.x2, p.y2, p.z2};
      double l1[3], l2[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) l1[c] = x2[0] * o[c] + x2[1] * o[3 + c] + x2[2] * o[6 + c];
#pragma unroll
      for (int r = 0; r < 3; ++r) l2[r] = o[3 * r] * x1[0] + o[3 * r + 1] * x1[1] + o[3 * r + 2] * x1[2];
      const double dd = x1[0] * l1[0] + x1[1] * l1[1] + x1[2] * l1[2];
      const double n1 = sqrt(l1[0] * l1[0] + l1[1] * l1[1]), n2 = sqrt(l2[0] * l2[0] + l2[1] * l2[1]);
      const double i1 = 1.0 / (n1 + 1e-6), i2 = 1.0 / (n2 + 1e-6);
      const double S = i1 + i2, ad = fabs(dd);
      const double g = (ad * S <= (double)clamp_at) ? (double)g_out[pair * N + i] : 0.0;
      const double sg = (dd > 0.0) ? 1.0 : ((dd < 0.0) ? -1.0 : 0.0);
      const double k1 = (n1 > 0.0) ? ad * i1 * i1 / n1 : 0.0;
      const double k2 = (n2 > 0.0) ? ad * i2 * i2 / n2 : 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          double t = sg * S * x2[r] * x1[c];
          if (c < 2) t -= k1 * l1[c] * x2[r];
          if (r < 2) t -= k2 * l2[r] * x1[c];
          go[3 * r + c] += g * t;
        }
    }
  }
  if (BWD) {
#pragma unroll
    for (int c = 0; c < 9; ++c) go[c] = wave_sum(go[c]);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 9; ++c) g_F[pair * 9 + c] = (float)go[c];
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// epipolar metrics on 2-D points (utils_F.py:291-361), one lane per correspondence
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
epi_metrics_kernel(int kind, const float* __restrict__ F, const float* __restrict__ X, const float* __restrict__ Y, int B,
                   int N, float clamp_at, float eps, float* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * N) return;
  const size_t b = idx / N;
  double f[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) f[c] = (double)F[b * 9 + c];
  const bool homo = (kind & DFEPE_EPI_HOMOGENEOUS) != 0;  // [B,N,3] homogeneous points used as they are (if_homo=True)
  kind &= 7;
  const int sd = homo ? 3 : 2;
  const double x[3] = {X[idx * sd], X[idx * sd + 1], homo ? (double)X[idx * sd + 2] : 1.0};
  const double y[3] = {Y[idx * sd], Y[idx * sd + 1], homo ? (double)Y[idx * sd + 2] : 1.0};
  double fx[3], fty[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) fx[r] = f[3 * r] * x[0] + f[3 * r + 1] * x[1] + f[3 * r + 2] * x[2];       // F x
#pragma unroll
  for (int c = 0; c < 3; ++c) fty[c] = f[c] * y[0] + f[3 + c] * y[1] + f[6 + c] * y[2];                    // F^T y
  const double num = y[0] * fx[0] + y[1] * fx[1] + y[2] * fx[2];
  const double a = fx[0] * fx[0] + fx[1] * fx[1], bq = fty[0] * fty[0] + fty[1] * fty[1];
  if (kind == 0) {
    double e = num * num * (1.0 / (a + (double)eps) + 1.0 / (bq + (double)eps));
    if (clamp_at >= 0.f) e = fmin(e, (double)clamp_at);  // negative: no clamp (clamp_at=None)
    out[idx] = (float)e;
  } else if (kind == 1) {
    out[idx] = (float)(num * num / (a + bq));
  } else {
    const double d1 = fabs(num) / sqrt(a), d2 = fabs(num) / sqrt(bq);
    const size_t plane = (size_t)B * N;
    out[idx] = (float)(0.5 * (d1 + d2));
    out[plane + idx] = (float)d1;
    out[2 * plane + idx] = (float)d2;
  }
}

// ------------------------------------------------------------------------------------------------------
// small pose helpers, one lane per item
// ------------------------------------------------------------------------------------------------------
__device__ inline void quat_of(const double* R, double* q) {
#define M_(i, j) R[3 * (j) + (i)]
  double v[4], t;
  if (M_(2, 2) < 0.0) {
    if (M_(0, 0) > M_(1, 1)) {
      t = 1.0 + M_(0, 0) - M_(1, 1) - M_(2, 2);
      v[0] = M_(1, 2) - M_(2, 1); v[1] = t; v[2] = M_(0, 1) + M_(1, 0); v[3] = M_(2, 0) + M_(0, 2);
    } else {
      t = 1.0 - M_(0, 0) + M_(1, 1) - M_(2, 2);
      v[0] = M_(2, 0) - M_(0, 2); v[1] = M_(0, 1) + M_(1, 0); v[2] = t; v[3] = M_(1, 2) + M_(2, 1);
    }
  } else {
    if (M_(0, 0) < -M_(1, 1)) {
      t = 1.0 - M_(0, 0) - M_(1, 1) + M_(2, 2);
      v[0] = M_(0, 1) - M_(1, 0); v[1] = M_(2, 0) + M_(0, 2); v[2] = M_(1, 2) + M_(2, 1); v[3] = t;
    } else {
      t = 1.0 + M_(0, 0) + M_(1, 1) + M_(2, 2);
      v[0] = t; v[1] = M_(1, 2) - M_(2, 1); v[2] = M_(2, 0) - M_(0, 2); v[3] = M_(0, 1) - M_(1, 0);
    }
  }
#undef M_
  double sc = 0.5 / sqrt(t);
  if (v[0] * sc < 0.0) sc = -sc;
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = sc * v[k];
}

// the four-fold ambiguity of utils_F._get_M2s (utils_F.py:478-498): R1 = U W V^T, R2 = U W^T V^T (both negated when
// det < 0), t = u3 / |u3|
__device__ inline void decompose_E(const double* E, double* R1, double* R2, double* t) {
  double U[9], S[3], V[9];
  svd3<double>(E, U, S, V);
  double UW[9], UWt[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    UW[3 * r + 0] = U[3 * r + 1]; UW[3 * r + 1] = -U[3 * r + 0]; UW[3 * r + 2] = U[3 * r + 2];
    UWt[3 * r + 0] = -U[3 * r + 1]; UWt[3 * r + 1] = U[3 * r + 0]; UWt[3 * r + 2] = U[3 * r + 2];
  }
  mat3_mul_nt(UW, V, R1);
  mat3_mul_nt(UWt, V, R2);
  const double det = R1[0] * (R1[4] * R1[8] - R1[5] * R1[7]) - R1[1] * (R1[3] * R1[8] - R1[5] * R1[6]) +
                     R1[2] * (R1[3] * R1[7] - R1[4] * R1[6]);
  if (det < 0.0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) { R1[k] = -R1[k]; R2[k] = -R2[k]; }
  }
  const double un = sqrt(U[2] * U[2] + U[5] * U[5] + U[8] * U[8]);
  t[0] = U[2] / un; t[1] = U[5] / un; t[2] = U[8] / un;
}

__global__ void __launch_bounds__(256)
geo_misc_kernel(int kind, const float* __restrict__ in0, const float* __restrict__ in1, int n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n) return;
  const double rad2deg = 57.29577951308232;
  if (kind == 0) {
    double R[9], q[4];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = (double)in0[i * 9 + k];
    quat_of(R, q);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[i * 4 + k] = (float)q[k];
  } else if (kind == 1) {
    double A[9], Bm[9], D[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { A[k] = (double)in0[i * 9 + k]; Bm[k] = (double)in1[i * 9 + k]; }
    mat3_mul_nt(A, Bm, D);
    const double ax = D[7] - D[5], ay = D[2] - D[6], az = D[3] - D[1];
    out[i] = (float)(atan2(sqrt(ax * ax + ay * ay + az * az), D[0] + D[4] + D[8] - 1.0) * rad2deg);
  } else if (kind == 2) {
    double dot = 0.0, n1 = 0.0, n2 = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double a = (double)in0[i * 3 + k], b = (double)in1[i * 3 + k];
      dot += a * b; n1 += a * a; n2 += b * b;
    }
    const double den = (sqrt(n1) + 1e-10) * (sqrt(n2) + 1e-10) + 1e-10;
    out[i] = (float)(acos(fmin(fmax(dot / den, -1.0), 1.0)) * rad2deg);
  } else if (kind == 3) {
    double E[9], U[9], S[3], V[9], Ed[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = (double)in0[i * 9 + k];
    svd3<double>(E, U, S, V);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Ed[3 * r + c] = U[3 * r] * V[3 * c] + U[3 * r + 1] * V[3 * c + 1];
#pragma unroll
    for (int k = 0; k < 9; ++k) out[i * 9 + k] = (float)Ed[k];
  } else if (kind == 5) {
    double F[9], A[9], tmp[9], E[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { F[k] = (double)in0[i * 9 + k]; A[k] = (double)in1[i * 9 + k]; }
    mat3_mul_tn(A, F, tmp);
    mat3_mul(tmp, A, E);
#pragma unroll
    for (int k = 0; k < 9; ++k) out[i * 9 + k] = (float)E[k];
  } else {
    double E[9], R1[9], R2[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = (double)in0[i * 9 + k];
    decompose_E(E, R1, R2, t);
#pragma unroll
    for (int k = 0; k < 9; ++k) { out[i * 21 + k] = (float)R1[k]; out[i * 21 + 9 + k] = (float)R2[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) out[i * 21 + 18 + k] = (float)t[k];
  }
}

// ------------------------------------------------------------------------------------------------------
// cheirality (utils_F._E_to_M_train, utils_F.py:679-763), one workgroup (4 wavefronts) per pair, one correspondence per lane
// ------------------------------------------------------------------------------------------------------
// Smallest eigenvector of a symmetric positive semi-definite 4x4 (the DLT normal matrix A^T A), fp64, in registers:
//   two Householder reflections -> tridiagonal T;  Laguerre's iteration from lam = 0 on det(T - lam I) through the
//   three-term recurrence (for a real-rooted polynomial it climbs monotonically to the smallest root from below, cubic
//   rate, and does not care whether the outliers' lam4 / lam3 is 1e-7 or 0.9: <= 6 steps on DLT matrices);
//   eigenvector of T as the best-conditioned column of adj(T - lam I), whose entries are products of the leading and
//   trailing principal minors the recurrence already yields (the same vector a twisted factorisation gives, pivot chosen
//   where |gamma_r| is smallest = where the diagonal cofactor is largest, without its six divisions);  back-transformation.
// Reciprocals and square roots are the branch-free Newton forms with ONE step on the fp32 seed (1e-14 relative): inside the
// iteration only the convergence speed depends on them, in the reflectors 1e-14 is the orthogonality they are built to.
// scripts/proto_eig4.py checks the algorithm against numpy.linalg.eigh on 16 000 DLT matrices with 25 % outliers (eigenvalue
// error 7e-16 lam_max, eigenvector error x gap 7e-16, no cheirality decision changed).
__device__ inline void smallest_eigvec4(const double* S /*4x4 row-major, symmetric*/, double* x) {
  // ---- Householder 1 on (S10, S20, S30)
  const double a0 = S[4], a1 = S[8], a2 = S[12];
  const double sg1 = a0 * a0 + a1 * a1 + a2 * a2;
  const double alpha = (a0 < 0.0) ? sqrt_nr<1>(sg1) : -sqrt_nr<1>(sg1);
  const double v0 = a0 - alpha, v1 = a1, v2 = a2;
  const double vv = v0 * v0 + v1 * v1 + v2 * v2;
  const bool ok1 = vv > 0.0;
  const double beta = ok1 ? 2.0 * rcp_nr<1>(vv) : 0.0;
  const double p0 = beta * (S[5] * v0 + S[6] * v1 + S[7] * v2);
  const double p1 = beta * (S[6] * v0 + S[10] * v1 + S[11] * v2);
  const double p2 = beta * (S[7] * v0 + S[11] * v1 + S[15] * v2);
  const double kc = 0.5 * beta * (p0 * v0 + p1 * v1 + p2 * v2);
  const double q0 = p0 - kc * v0, q1 = p1 - kc * v1, q2 = p2 - kc * v2;
  const double B00 = S[5] - 2.0 * v0 * q0;
  const double B01 = S[6] - v0 * q1 - q0 * v1, B02 = S[7] - v0 * q2 - q0 * v2;
  const double B11 = S[10] - 2.0 * v1 * q1, B12 = S[11] - v1 * q2 - q1 * v2, B22 = S[15] - 2.0 * v2 * q2;
  // ---- Householder 2 on (B10, B20)
  const double sg2 = B01 * B01 + B02 * B02;
  const double alpha2 = (B01 < 0.0) ? sqrt_nr<1>(sg2) : -sqrt_nr<1>(sg2);
  const double w0 = B01 - alpha2, w1 = B02;
  const double ww = w0 * w0 + w1 * w1;
  const bool ok2 = ww > 0.0;
  const double beta2 = ok2 ? 2.0 * rcp_nr<1>(ww) : 0.0;
  const double r0 = beta2 * (B11 * w0 + B12 * w1), r1 = beta2 * (B12 * w0 + B22 * w1);
  const double k2 = 0.5 * beta2 * (r0 * w0 + r1 * w1);
  const double s0 = r0 - k2 * w0, s1 = r1 - k2 * w1;
  const double d0 = S[0], d1 = B00, d2 = B11 - 2.0 * w0 * s0, d3 = B22 - 2.0 * w1 * s1;
  const double e0 = ok1 ? alpha : a0, e1 = ok2 ? alpha2 : B01, e2 = B12 - w0 * s1 - s0 * w1;
  const double f0 = e0 * e0, f1 = e1 * e1, f2 = e2 * e2;
  const double scale = fmax(fmax(fabs(d0), fabs(d1)), fmax(fabs(d2), fabs(d3))) + fmax(fabs(e0), fmax(fabs(e1), fabs(e2)));
  // ---- Laguerre from below (all roots are >= 0: the start lam = 0 is left of, or on, the smallest one)
  double lam = 0.0;
  bool done = false;
  for (int it = 0; it < 12; ++it) {
    const double c0 = d0 - lam, c1 = d1 - lam, c2 = d2 - lam, c3 = d3 - lam;
    // p_k = c_k p_{k-1} - f_{k-1} p_{k-2} and its first two derivatives with respect to lam
    const double P2 = c1 * c0 - f0, D2 = -c1 - c0, E2 = 2.0;
    const double P3 = c2 * P2 - f1 * c0, D3 = c2 * D2 - P2 + f1, E3 = c2 * E2 - 2.0 * D2;
    const double P4 = c3 * P3 - f2 * P2, D4 = c3 * D3 - P3 - f2 * D2, E4 = c3 * E3 - 2.0 * D3 - f2 * E2;
    const bool good = (P4 > 0.0) && !done;  // p <= 0: on the root (or past it by round-off)
    const double ip = good ? rcp_nr<1>(P4) : 0.0;
    const double G = D4 * ip, H = G * G - E4 * ip;
    const double den = G - sqrt_nr<1>(fmax(3.0 * (4.0 * H - G * G), 0.0));  // G < 0 left of the smallest root
    const double step = (good && den < 0.0) ? -4.0 * rcp_nr<1>(den) : 0.0;
    const double nl = lam + step;
    done = done || !good || !(step > 1e-16 * scale) || (nl == lam);
    if (!done) lam = nl;
    if (__ballot(!done) == 0ull) break;  // wave-uniform exit
  }
  // ---- null vector of T - lam: column r of the adjugate, adj[i][j] = (-1)^(i+j) P_i e_i..e_(j-1) Q_(j+1) for i <= j, with
  // P_i the leading principal minor of order i and Q_j the trailing one starting at row j; r = the largest diagonal cofactor
  const double c0 = d0 - lam, c1 = d1 - lam, c2 = d2 - lam, c3 = d3 - lam;
  const double P1 = c0, P2 = c1 * c0 - f0, P3 = c2 * P2 - f1 * P1;
  const double Q3 = c3, Q2 = c2 * c3 - f2, Q1 = c1 * Q2 - f1 * Q3;
  const double g0 = fabs(Q1), g1 = fabs(P1 * Q2), g2 = fabs(P2 * Q3), g3 = fabs(P3);
  int r = 0;
  double gm = g0;
  if (g1 > gm) { gm = g1; r = 1; }
  if (g2 > gm) { gm = g2; r = 2; }
  if (g3 > gm) { gm = g3; r = 3; }
  const double e01 = e0 * e1, e12 = e1 * e2, e012 = e01 * e2;
  double y0 = (r == 0) ? Q1 : ((r == 1) ? -e0 * Q2 : ((r == 2) ? e01 * Q3 : -e012));
  double y1 = (r == 0) ? -e0 * Q2 : ((r == 1) ? P1 * Q2 : ((r == 2) ? -P1 * e1 * Q3 : P1 * e12));
  double y2 = (r == 0) ? e01 * Q3 : ((r == 1) ? -P1 * e1 * Q3 : ((r == 2) ? P2 * Q3 : -P2 * e2));
  double y3 = (r == 0) ? -e012 : ((r == 1) ? P1 * e12 : ((r == 2) ? -P2 * e2 : P3));
  // ---- back-transformation x = H1 H2 y
  const double t2 = beta2 * (w0 * y2 + w1 * y3);
  y2 -= t2 * w0; y3 -= t2 * w1;
  const double t1 = beta * (v0 * y1 + v1 * y2 + v2 * y3);
  y1 -= t1 * v0; y2 -= t1 * v1; y3 -= t1 * v2;
  x[0] = y0; x[1] = y1; x[2] = y2; x[3] = y3;  // not normalised: the caller only uses ratios
}

__global__ void __launch_bounds__(256)
cheirality_kernel(const float* __restrict__ E, const float* __restrict__ pre, const float* __restrict__ K,
                  const float* __restrict__ matches, int B, int N, float depth_thres, float* __restrict__ Rt_cam,
                  int* __restrict__ winner, int* __restrict__ counts) {
  // one workgroup per pair; with 256 threads the four wavefronts take every fourth group of 64 correspondences and meet in LDS
  __shared__ int wcnt[4][4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform by construction
  const size_t pair = blockIdx.x;
  double Ed[9], Kd[9], R[2][9], t[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) { Ed[k] = (double)E[pair * 9 + k]; Kd[k] = to_sgpr((double)K[pair * 9 + k]); }
  if (pre != nullptr) {  // E-from-F fused: the matrix decomposed is pre^T E pre (E = F, pre = T K; train_good_utils.py:356-358)
    double Ad[9], tmp[9], Ef[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Ad[k] = (double)pre[pair * 9 + k];
    mat3_mul_tn(Ad, Ed, tmp);
    mat3_mul(tmp, Ad, Ef);
#pragma unroll
    for (int k = 0; k < 9; ++k) Ed[k] = (double)(float)Ef[k];  // through fp32 like the stand-alone congruence kernel's output
  }
  decompose_E(Ed, R[0], R[1], t);
  // per-pair (wave-uniform) quantities live in scalar registers; the per-correspondence DLT owns the VGPRs
#pragma unroll
  for (int k = 0; k < 9; ++k) { R[0][k] = to_sgpr(R[0][k]); R[1][k] = to_sgpr(R[1][k]); }
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = to_sgpr(t[k]);
  // the two candidate projection matrices K [R | t], formed once per pair (measured: parking them in scalar registers and
  // capping the kernel at 128 VGPRs for four wavefronts per SIMD is slower -- SGPR spills and constant-bus moves in the loop)
  double P2s[2][12];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        P2s[rr][4 * r + c] = Kd[3 * r] * R[rr][c] + Kd[3 * r + 1] * R[rr][3 + c] + Kd[3 * r + 2] * R[rr][6 + c];
      P2s[rr][4 * r + 3] = Kd[3 * r] * t[0] + Kd[3 * r + 1] * t[1] + Kd[3 * r + 2] * t[2];
    }
  int cnt[4] = {0, 0, 0, 0};
  const int nw = blockDim.x >> 6;  // 4 wavefronts per pair for small batches (latency), 1 for large ones (throughput)
  // the next group's correspondence is loaded (index clamped, no branch) before the current one is triangulated: ~1 600
  // instructions of DLT work cover its latency
  const float4* mrow = reinterpret_cast<const float4*>(matches) + pair * N;
  float4 mnext = mrow[min(wave * WAVE + lane, N - 1)];
  for (int base = wave * WAVE; base < N; base += nw * WAVE) {
    const int i = base + lane;
    const bool live = i < N;
    const float4 m = mnext;
    mnext = mrow[min(i + nw * WAVE, N - 1)];
    // One DLT per rotation: flipping t negates the 4th column of the view-2 rows, hence the 4th component of the null
    // vector, hence both depths exactly -- candidates (R,t) and (R,-t) are counted from the same triangulation.
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const double* Rc = R[rr];
      const double* P2 = P2s[rr];
      // DLT rows: x*P[2]-P[0], y*P[2]-P[1] for both views (P1 = K [I|0])
      double A[16];
      const double x1 = m.x, y1 = m.y, x2 = m.z, y2 = m.w;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double p1r0 = (c < 3) ? Kd[c] : 0.0, p1r1 = (c < 3) ? Kd[3 + c] : 0.0, p1r2 = (c < 3) ? Kd[6 + c] : 0.0;
        A[c] = x1 * p1r2 - p1r0;
        A[4 + c] = y1 * p1r2 - p1r1;
        A[8 + c] = x2 * P2[8 + c] - P2[c];
        A[12 + c] = y2 * P2[8 + c] - P2[4 + c];
      }
      // normal matrix A^T A (symmetric: 10 distinct entries), scaled to unit trace (the null vector does not care; the
      // eigen-solver's Newton seeds want O(1) operands whatever the pixel scale)
      double S[16];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = r; c < 4; ++c) S[4 * r + c] = A[r] * A[c] + A[4 + r] * A[4 + c] + A[8 + r] * A[8 + c] + A[12 + r] * A[12 + c];
      const double itr = rcp_nr<1>(fmax(S[0] + S[5] + S[10] + S[15], 1e-30));
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = r; c < 4; ++c) { S[4 * r + c] *= itr; S[4 * c + r] = S[4 * r + c]; }
      double X[4];
      smallest_eigvec4(S, X);
      // depths z1 = X2 / X3 and z2 = (R_3 . X_012 + t_3 X3) / X3 tested without the division: 0 < z < thr  <=>  z' w > 0 and
      // |z'| < thr |w| for z = z' / w
      const double wq = X[3];
      const double z1n = X[2];
      const double z2n = Rc[6] * X[0] + Rc[7] * X[1] + Rc[8] * X[2] + t[2] * wq;
      const double thr = (double)depth_thres;
      const double aw = thr * fabs(wq);
      const bool inr = live && (fabs(z1n) < aw) && (fabs(z2n) < aw) && (wq != 0.0);
      const bool s1p = (z1n > 0.0) == (wq > 0.0), s2p = (z2n > 0.0) == (wq > 0.0);
      const bool nz = (z1n != 0.0) && (z2n != 0.0);
      const bool pos = inr && nz && s1p && s2p;    // both depths in (0, thr)
      const bool neg = inr && nz && !s1p && !s2p;  // both in (-thr, 0): the (R, -t) candidate sees them in (0, thr)
      cnt[2 * rr] += __popcll(__ballot(pos));
      cnt[2 * rr + 1] += __popcll(__ballot(neg));
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) wcnt[wave][c] = cnt[c];
  }
  __syncthreads();
  if (wave != 0) return;
  if (nw > 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) cnt[c] = (wcnt[0][c] + wcnt[1][c]) + (wcnt[2][c] + wcnt[3][c]);
  }
  int win = 0;
#pragma unroll
  for (int c = 1; c < 4; ++c)
    if (cnt[c] > cnt[win]) win = c;  // first maximum, like max(enumerate(...)) (utils_F.py:730)
  if (lane == 0) {
    if (counts != nullptr) {
#pragma unroll
      for (int c = 0; c < 4; ++c) counts[pair * 4 + c] = cnt[c];
    }
    if (winner != nullptr) winner[pair] = (cnt[win] > 0) ? win : -1;
    // camera motion = inverse of [R|t]: [R^T | -R^T t]   (utils_misc._inv_Rt, utils_misc.py:115-121)
    double Rc[9];  // selected by value: a runtime index into R would put the whole array into scratch memory
#pragma unroll
    for (int k = 0; k < 9; ++k) Rc[k] = (win >> 1) ? R[1][k] : R[0][k];
    const double sg = (win & 1) ? -1.0 : 1.0;
    float* dst = Rt_cam + pair * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dst[4 * r + c] = (cnt[win] > 0) ? (float)Rc[3 * c + r] : 0.0f;
      const double tc = -(Rc[r] * t[0] + Rc[3 + r] * t[1] + Rc[6 + r] * t[2]) * sg;
      dst[4 * r + 3] = (cnt[win] > 0) ? (float)tc : 0.0f;
    }
  }
}

}  // namespace

extern "C" int dfepe_epi_residual_fwd(const float* pts1, const float* pts2, const float* F, int B, int N, float clamp_at,
                                      float* out, void* stream) {
  if (B < 0 || N <= 0) return DFEPE_ERR_INVALID_ARG;
  if (B == 0) return DFEPE_OK;
  if (!pts1 || !pts2 || !F || !out) return DFEPE_ERR_INVALID_ARG;
  hipLaunchKernelGGL(epi_residual_kernel<false>, dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), pts1, pts2,
                     F, B, N, clamp_at, out, nullptr, nullptr);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_epi_residual_bwd(const float* pts1, const float* pts2, const float* F, int B, int N, float clamp_at,
                                      const float* g_out, float* g_F, void* stream) {
  if (B < 0 || N <= 0) return DFEPE_ERR_INVALID_ARG;
  if (B == 0) return DFEPE_OK;
  if (!pts1 || !pts2 || !F || !g_out || !g_F) return DFEPE_ERR_INVALID_ARG;
  hipLaunchKernelGGL(epi_residual_kernel<true>, dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), pts1, pts2,
                     F, B, N, clamp_at, nullptr, g_out, g_F);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_epi_metrics(int kind, const float* F, const float* X, const float* Y, int B, int N, float clamp_at,
                                 float eps, float* out, void* stream) {
  if (kind < 0 || (kind & ~DFEPE_EPI_HOMOGENEOUS) > 2 || B < 0 || N <= 0) return DFEPE_ERR_INVALID_ARG;
  if (B == 0) return DFEPE_OK;
  if (!F || !X || !Y || !out) return DFEPE_ERR_INVALID_ARG;
  const size_t n = (size_t)B * N;
  hipLaunchKernelGGL(epi_metrics_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), kind,
                     F, X, Y, B, N, clamp_at, eps, out);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_geo_misc(int kind, const float* in0, const float* in1, int n, float* out, void* stream) {
  if (kind < 0 || kind > 5 || n < 0) return DFEPE_ERR_INVALID_ARG;
  if (n == 0) return DFEPE_OK;
  if (!in0 || !out || ((kind == 1 || kind == 2 || kind == 5) && !in1)) return DFEPE_ERR_INVALID_ARG;
  hipLaunchKernelGGL(geo_misc_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), kind, in0, in1, n, out);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_cheirality(const float* E, const float* pre, const float* K, const float* matches, int B, int N,
                                float depth_thres, float* Rt_cam, int* winner, int* counts, void* stream) {
  if (B < 0 || N <= 0) return DFEPE_ERR_INVALID_ARG;
  if (B == 0) return DFEPE_OK;
  if (!E || !K || !matches || !Rt_cam) return DFEPE_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(matches) & 15u) return DFEPE_ERR_INVALID_ARG;
  hipLaunchKernelGGL(cheirality_kernel, dim3(B), dim3(B >= 2048 ? 64 : 256), 0, static_cast<hipStream_t>(stream), E, pre, K, matches,
                     B, N, depth_thres, Rt_cam, winner, counts);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
