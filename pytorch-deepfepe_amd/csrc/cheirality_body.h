// cheirality_body.h -- cheirality-checked pose selection of one pair by one workgroup (utils_F._E_to_M_train,
// deepFEPE/dsac_tools/utils_F.py:679-763; candidates of utils_F._get_M2s :478-498).  Shared by the stand-alone kernel (geom.hip)
// and the fused fit + E-from-F + pose kernel of the cooperative fit (w8pt16.hip).
#pragma once
#include <type_traits>

#include "dfepe_common.h"

// internal linkage on purpose: with external (inline) linkage hipcc keeps decompose_E / svd3_closed as real calls, which drags
// the module's LDS and the full ABI into the kernel (35 instead of 23 us at 512 pairs)
namespace {

// the four-fold ambiguity of utils_F._get_M2s (utils_F.py:478-498): R1 = U W V^T, R2 = U W^T V^T (both negated when
// det < 0), t = u3 / |u3|
__device__ inline void decompose_E(const double* E, double* R1, double* R2, double* t) {
  double U[9], S[3], V[9];
  svd3_closed(E, U, S, V);
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

// Smallest eigenvector of a symmetric positive semi-definite 4x4 (the DLT normal matrix A^T A), in registers, two stages.
// Stage 1 (smallest_eigvec4_pk), packed fp32, TWO matrices at once (v_pk_fma_f32: component 0 / 1 = rotation candidate 1 / 2):
//   two Householder reflections -> tridiagonal T;  Laguerre's iteration from lam = 0 on det(T - lam I) through the
//   three-term recurrence (for a real-rooted polynomial it climbs monotonically to the smallest root from below, cubic
//   rate, and does not care whether the outliers' lam4 / lam3 is 1e-7 or 0.9: <= 6 steps on DLT matrices);
//   eigenvector of T as the best-conditioned column of adj(T - lam I), whose entries are products of the leading and
//   trailing principal minors the recurrence already yields (the vector a twisted factorisation gives -- its pivot gamma_r is
//   smallest where the diagonal cofactor is largest -- without the divisions);  back-transformation.
//   scripts/proto_eig4.py checks this route against numpy.linalg.eigh in fp64 (eigenvector error x gap 8e-16);
//   scripts/proto_dlt_refine.py runs it in fp32: median eigenvector error 2e-7, 1e-4 at the 99.9th percentile.
// Stage 2 (rqi_refine4) makes that an fp64 answer with ONE Rayleigh-quotient iteration on the fp64 matrix: cubic
//   convergence, error <= 1e-9 on every DLT matrix of the prototype's scenes (25-50 % outliers), no cheirality decision
//   different from eigh.  ~900 instructions per correspondence (both candidates) against ~1 200 for two fp64 tridiagonal
//   solves (round 2's first version: 161 us at B = 4096, N = 1000) and ~4 000 for the cyclic Jacobi of round 1.
__device__ __forceinline__ double guard_piv(double z, double tiny) { return (fabs(z) < tiny) ? ((z < 0.0) ? -tiny : tiny) : z; }

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
// the stage-1 eigenvector routine is written once for a value type V = f2 (two matrices at once, v_pk_*_f32) or float (one matrix:
// the fp64 route, which handles one rotation candidate at a time to stay inside 128 registers); M = its comparison-mask type
__device__ __forceinline__ f2 f2s(float v) { return f2{v, v}; }
template <class V> __device__ __forceinline__ V vsplat(float v);
template <> __device__ __forceinline__ f2 vsplat<f2>(float v) { return f2{v, v}; }
template <> __device__ __forceinline__ float vsplat<float>(float v) { return v; }
__device__ __forceinline__ i2 m_lt(f2 a, f2 b) { return a < b; }
__device__ __forceinline__ i2 m_gt(f2 a, f2 b) { return a > b; }
__device__ __forceinline__ i2 m_eq(f2 a, f2 b) { return a == b; }
__device__ __forceinline__ int m_lt(float a, float b) { return (a < b) ? -1 : 0; }
__device__ __forceinline__ int m_gt(float a, float b) { return (a > b) ? -1 : 0; }
__device__ __forceinline__ int m_eq(float a, float b) { return (a == b) ? -1 : 0; }
__device__ __forceinline__ bool m_all(i2 m) { return m.x && m.y; }
__device__ __forceinline__ bool m_all(int m) { return m != 0; }
__device__ __forceinline__ f2 pk_sel(i2 m, f2 a, f2 b) { return m ? a : b; }
__device__ __forceinline__ float pk_sel(int m, float a, float b) { return m ? a : b; }
__device__ __forceinline__ f2 pk_abs(f2 a) { return __builtin_elementwise_abs(a); }
__device__ __forceinline__ float pk_abs(float a) { return fabsf(a); }
__device__ __forceinline__ f2 pk_max(f2 a, f2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ float pk_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ f2 pk_sqrt(f2 a) { return f2{hw_sqrt(a.x), hw_sqrt(a.y)}; }
__device__ __forceinline__ float pk_sqrt(float a) { return hw_sqrt(a); }
__device__ __forceinline__ f2 pk_rcp(f2 a) { return f2{hw_rcp(a.x), hw_rcp(a.y)}; }
__device__ __forceinline__ float pk_rcp(float a) { return hw_rcp(a); }

// gapprod (optional): |d det(T - lam I) / d lam| at the returned eigenvalue = (lam1 - lam4)(lam2 - lam4)(lam3 - lam4) for a unit-trace
// matrix: the quantity that bounds the error of x (see cheirality_pair: the fp32 stage decides a correspondence on its own only
// where that bound leaves its depth tests unambiguous).
template <class f2>  // f2 (packed pair) or float: see the helpers above
__device__ inline void smallest_eigvec4_pk(const f2* S /*4x4 row-major, symmetric, unit trace*/, f2* x, f2* gapprod = nullptr) {
  typedef decltype(m_lt(f2{}, f2{})) i2;
  auto f2s = [](float v) { return vsplat<f2>(v); };
  const f2 zero = f2s(0.0f), two = f2s(2.0f);
  // ---- Householder 1 on (S10, S20, S30)
  const f2 a0 = S[4], a1 = S[8], a2 = S[12];
  const f2 n1 = pk_sqrt(a0 * a0 + a1 * a1 + a2 * a2);
  const f2 alpha = pk_sel(m_lt(a0, zero), n1, -n1);
  const f2 v0 = a0 - alpha, v1 = a1, v2 = a2;
  const f2 vv = v0 * v0 + v1 * v1 + v2 * v2;
  const i2 ok1 = m_gt(vv, zero);
  const f2 beta = pk_sel(ok1, two * pk_rcp(pk_max(vv, f2s(1e-30f))), zero);
  const f2 p0 = beta * (S[5] * v0 + S[6] * v1 + S[7] * v2);
  const f2 p1 = beta * (S[6] * v0 + S[10] * v1 + S[11] * v2);
  const f2 p2 = beta * (S[7] * v0 + S[11] * v1 + S[15] * v2);
  const f2 kc = f2s(0.5f) * beta * (p0 * v0 + p1 * v1 + p2 * v2);
  const f2 q0 = p0 - kc * v0, q1 = p1 - kc * v1, q2 = p2 - kc * v2;
  const f2 B00 = S[5] - two * v0 * q0;
  const f2 B01 = S[6] - v0 * q1 - q0 * v1, B02 = S[7] - v0 * q2 - q0 * v2;
  const f2 B11 = S[10] - two * v1 * q1, B12 = S[11] - v1 * q2 - q1 * v2, B22 = S[15] - two * v2 * q2;
  // ---- Householder 2 on (B10, B20)
  const f2 n2 = pk_sqrt(B01 * B01 + B02 * B02);
  const f2 alpha2 = pk_sel(m_lt(B01, zero), n2, -n2);
  const f2 w0 = B01 - alpha2, w1 = B02;
  const f2 ww = w0 * w0 + w1 * w1;
  const i2 ok2 = m_gt(ww, zero);
  const f2 beta2 = pk_sel(ok2, two * pk_rcp(pk_max(ww, f2s(1e-30f))), zero);
  const f2 r0 = beta2 * (B11 * w0 + B12 * w1), r1 = beta2 * (B12 * w0 + B22 * w1);
  const f2 k2 = f2s(0.5f) * beta2 * (r0 * w0 + r1 * w1);
  const f2 s0 = r0 - k2 * w0, s1 = r1 - k2 * w1;
  const f2 d0 = S[0], d1 = B00, d2 = B11 - two * w0 * s0, d3 = B22 - two * w1 * s1;
  const f2 e0 = pk_sel(ok1, alpha, a0), e1 = pk_sel(ok2, alpha2, B01), e2 = B12 - w0 * s1 - s0 * w1;
  const f2 f0 = e0 * e0, f1 = e1 * e1, f2_ = e2 * e2;
  const f2 scale = pk_max(pk_max(pk_abs(d0), pk_abs(d1)), pk_max(pk_abs(d2), pk_abs(d3))) + pk_max(pk_abs(e0), pk_max(pk_abs(e1), pk_abs(e2)));
  // ---- Laguerre from below (all roots are >= 0: the start lam = 0 is left of, or on, the smallest one); a step below the
  // fp32 resolution of the spectrum ends the iteration
  f2 lam = zero;
  i2 done = m_lt(zero, zero);  // all false
  const f2 stop = f2s(1e-7f) * scale;
  for (int it = 0; it < 10; ++it) {
    const f2 c0 = d0 - lam, c1 = d1 - lam, c2 = d2 - lam, c3 = d3 - lam;
    const f2 P2 = c1 * c0 - f0, D2 = -c1 - c0;
    const f2 P3 = c2 * P2 - f1 * c0, D3 = c2 * D2 - P2 + f1, E3 = two * c2 - two * D2;
    const f2 P4 = c3 * P3 - f2_ * P2, D4 = c3 * D3 - P3 - f2_ * D2, E4 = c3 * E3 - two * D3 - two * f2_;
    const i2 good = m_gt(P4, zero) & ~done;
    const f2 ip = pk_sel(good, pk_rcp(pk_max(P4, f2s(1e-37f))), zero);
    const f2 G = D4 * ip, H = G * G - E4 * ip;
    const f2 den = G - pk_sqrt(pk_max(f2s(3.0f) * (f2s(4.0f) * H - G * G), zero));
    const i2 stepok = good & m_lt(den, zero);
    const f2 step = pk_sel(stepok, f2s(-4.0f) * pk_rcp(pk_sel(stepok, den, f2s(-1.0f))), zero);
    const f2 nl = lam + step;
    done = done | ~good | ~m_gt(step, stop) | m_eq(nl, lam);
    lam = pk_sel(done, lam, nl);
    if (__ballot(!m_all(done)) == 0ull) break;  // wave-uniform exit
  }
  // ---- null vector of T - lam: the adjugate column with the largest diagonal cofactor
  const f2 c0 = d0 - lam, c1 = d1 - lam, c2 = d2 - lam, c3 = d3 - lam;
  const f2 P1 = c0, P2 = c1 * c0 - f0, P3 = c2 * P2 - f1 * P1;
  if (gapprod != nullptr) {  // derivative of the characteristic polynomial by the same recurrence the iteration uses
    const f2 D2 = -c1 - c0, D3 = c2 * D2 - P2 + f1;
    *gapprod = pk_abs(c3 * D3 - P3 - f2_ * D2);
  }
  const f2 Q3 = c3, Q2 = c2 * c3 - f2_, Q1 = c1 * Q2 - f1 * Q3;
  const f2 g0 = pk_abs(Q1), g1 = pk_abs(P1 * Q2), g2 = pk_abs(P2 * Q3), g3 = pk_abs(P3);
  const i2 b1 = m_gt(g1, g0);               // best of (0, 1)
  const f2 g01 = pk_sel(b1, g1, g0);
  const i2 b3 = m_gt(g3, g2);               // best of (2, 3)
  const f2 g23 = pk_sel(b3, g3, g2);
  const i2 hi = m_gt(g23, g01);               // r in {2, 3} else {0, 1}; ties keep the lower index
  const f2 e01 = e0 * e1, e12 = e1 * e2, e012 = e01 * e2;
  // columns r = 0..3 of the adjugate
  const f2 y0 = pk_sel(hi, pk_sel(b3, -e012, e01 * Q3), pk_sel(b1, -e0 * Q2, Q1));
  const f2 y1 = pk_sel(hi, pk_sel(b3, P1 * e12, -P1 * e1 * Q3), pk_sel(b1, P1 * Q2, -e0 * Q2));
  f2 y2 = pk_sel(hi, pk_sel(b3, -P2 * e2, P2 * Q3), pk_sel(b1, -P1 * e1 * Q3, e01 * Q3));
  f2 y3 = pk_sel(hi, pk_sel(b3, P3, -P2 * e2), pk_sel(b1, P1 * e12, -e012));
  f2 yy1 = y1;
  // ---- back-transformation x = H1 H2 y
  const f2 t2 = beta2 * (w0 * y2 + w1 * y3);
  y2 -= t2 * w0; y3 -= t2 * w1;
  const f2 t1 = beta * (v0 * yy1 + v1 * y2 + v2 * y3);
  yy1 -= t1 * v0; y2 -= t1 * v1; y3 -= t1 * v2;
  x[0] = y0; x[1] = yy1; x[2] = y2; x[3] = y3;
}

// One Rayleigh-quotient iteration in fp64: y = (S - rho I)^-1 x0, rho = x0^T S x0 / x0^T x0, by LDL^T without pivoting (S - rho I
// is positive semi-definite up to the error of rho: only the last pivot is small, and a floored pivot only scales y).
// S symmetric 4x4 (upper triangle read), x0 any non-zero vector.  y is not normalised: the caller uses ratios.
__device__ inline void rqi_refine4(const double* S, const double* x0, double* y) {
  const double a = x0[0], b = x0[1], c = x0[2], d = x0[3];
  const double nn = a * a + b * b + c * c + d * d;
  const double Sa = S[0] * a + S[1] * b + S[2] * c + S[3] * d, Sb = S[1] * a + S[5] * b + S[6] * c + S[7] * d;
  const double Sc = S[2] * a + S[6] * b + S[10] * c + S[11] * d, Sd = S[3] * a + S[7] * b + S[11] * c + S[15] * d;
  const double rho = (nn > 0.0) ? (a * Sa + b * Sb + c * Sc + d * Sd) * rcp_nr<1>(nn) : 0.0;
  const double tiny = 1e-30;
  const double m00 = S[0] - rho, m11 = S[5] - rho, m22 = S[10] - rho, m33 = S[15] - rho;
  const double D0 = guard_piv(m00, tiny), i0 = rcp_nr<1>(D0);
  const double l10 = S[1] * i0, l20 = S[2] * i0, l30 = S[3] * i0;
  const double D1 = guard_piv(m11 - l10 * S[1], tiny), i1 = rcp_nr<1>(D1);
  const double u21 = S[6] - l20 * S[1], u31 = S[7] - l30 * S[1];  // (row i, col 1) after eliminating column 0
  const double l21 = u21 * i1, l31 = u31 * i1;
  const double D2 = guard_piv(m22 - l20 * S[2] - l21 * u21, tiny), i2_ = rcp_nr<1>(D2);
  const double u32 = S[11] - l30 * S[2] - l31 * u21;
  const double l32 = u32 * i2_;
  const double D3 = guard_piv(m33 - l30 * S[3] - l31 * u31 - l32 * u32, tiny), i3 = rcp_nr<1>(D3);
  // L z = x0
  const double z0 = a, z1 = b - l10 * z0, z2 = c - l20 * z0 - l21 * z1, z3 = d - l30 * z0 - l31 * z1 - l32 * z2;
  // D w = z, L^T y = w
  const double w3 = z3 * i3, w2 = z2 * i2_, w1 = z1 * i1, w0 = z0 * i0;
  y[3] = w3;
  y[2] = w2 - l32 * y[3];
  y[1] = w1 - l21 * y[2] - l31 * y[3];
  y[0] = w0 - l10 * y[1] - l20 * y[2] - l30 * y[3];
}

constexpr int kCheirQueue = 128;  // ints of LDS per wavefront: indices of the correspondences waiting for the fp64 route (< 64 + 64)
// What the correspondence loop needs of a pair, all in doubles: R1 (0..8), R2 (9..17), t (18..20), the two candidate projection
// matrices K [R | t] (21..32, 33..44), K (45..53).  Formed ONCE per pair -- by dfepe_cheirality_ex's preparation launch (one LANE per
// pair, into the caller's workspace; the main kernel then fetches it with scalar loads), or by wavefront 0 of the pair's workgroup
// (into LDS) -- instead of once per wavefront: it is a per-pair scalar computation (closed-form fp64 SVD of E, ~1 400 vector
// instructions when issued for 64 lanes), 12 % of the kernel at one wavefront per pair and half of it at eight.
constexpr int kCheirPrep = 56;
constexpr int kPrepR = 0, kPrepT = 18, kPrepP = 21, kPrepK = 45;
struct CheirLds {
  double prep[kCheirPrep];
  int wcnt[8][4];
  int queue[8 * kCheirQueue];
};

// One workgroup per pair (every thread of the workgroup must call; W = blockDim.x / 64 wavefronts each take every W-th group of 64
// correspondences and meet in `wcnt`, LDS owned by the caller).  E9: the matrix to decompose (or F when `pre` is given: then
// pre^T E9 pre is decomposed), identical in every thread.
__device__ inline void cheir_prepare(const float* E9, const float* pre9 /* this pair's, or nullptr */, const float* K9, double* out /* kCheirPrep */) {
  double Ed[9], Kd[9], R1[9], R2[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) { Ed[k] = (double)E9[k]; Kd[k] = (double)K9[k]; }
  if (pre9 != nullptr) {  // E-from-F fused: the matrix decomposed is pre^T E pre (E = F, pre = T K; train_good_utils.py:356-358)
    double Ad[9], tmp[9], Ef[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Ad[k] = (double)pre9[k];
    mat3_mul_tn(Ad, Ed, tmp);
    mat3_mul(tmp, Ad, Ef);
#pragma unroll
    for (int k = 0; k < 9; ++k) Ed[k] = (double)(float)Ef[k];  // through fp32 like the stand-alone congruence kernel's output
  }
  decompose_E(Ed, R1, R2, t);
#pragma unroll
  for (int k = 0; k < 9; ++k) { out[kPrepR + k] = R1[k]; out[kPrepR + 9 + k] = R2[k]; out[kPrepK + k] = Kd[k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) out[kPrepT + k] = t[k];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const double* R = rr ? R2 : R1;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) out[kPrepP + 12 * rr + 4 * r + c] = Kd[3 * r] * R[c] + Kd[3 * r + 1] * R[3 + c] + Kd[3 * r + 2] * R[6 + c];
      out[kPrepP + 12 * rr + 4 * r + 3] = Kd[3 * r] * t[0] + Kd[3 * r + 1] * t[1] + Kd[3 * r + 2] * t[2];
    }
  }
  out[54] = out[55] = 0.0;
}

// FP64_ONLY: every correspondence through the fp64 route (DFEPE_CHEIR_FP64_ONLY: the reference build the adaptive one is tested
// against for exact equality of the counts; also its upper bound in time).
// WS: the pair's constants come from the workspace `ws` (filled by cheir_prepare in a launch of its own) instead of being formed here.
template <bool FP64_ONLY = false, bool WS = false>
__device__ __forceinline__ void cheirality_pair(const float* E9, const float* __restrict__ pre, const float* __restrict__ K,
                                                const float* __restrict__ matches, const size_t pair, const int N, const float depth_thres,
                                                float* __restrict__ Rt_cam, int* __restrict__ winner, int* __restrict__ counts,
                                                CheirLds& cl, const double* __restrict__ ws = nullptr) {
  int (*wcnt)[4] = cl.wcnt;
  int* queue = cl.queue;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform by construction
  // ---- the pair's constants (see kCheirPrep) into scalar registers: from the workspace a preparation launch filled (uniform
  // address: scalar loads), else formed by wavefront 0 and shared through LDS
  const double* wsp = ws + pair * kCheirPrep;  // WS only
  if constexpr (!WS) {
    if (wave == 0) {
      double mine[kCheirPrep];
      cheir_prepare(E9, pre != nullptr ? pre + pair * 9 : nullptr, K + pair * 9, mine);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 54; ++k) cl.prep[k] = mine[k];
      }
    }
    __syncthreads();
  }
  auto prep = [&](const int k) { if constexpr (WS) return wsp[k]; else return cl.prep[k]; };
  // scalar-register budget of the loop: K (18), the two projection matrices (48), the third rows of R1 / R2 and t_z (14); the rest
  // of the decomposition is only needed for the winner's output and is re-read there
  double Kd[9], P2s[2][12];
#pragma unroll
  for (int k = 0; k < 9; ++k) Kd[k] = to_sgpr(prep(kPrepK + k));
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int k = 0; k < 12; ++k) P2s[rr][k] = to_sgpr(prep(kPrepP + 12 * rr + k));
  const double Rz[2][3] = {{to_sgpr(prep(kPrepR + 6)), to_sgpr(prep(kPrepR + 7)), to_sgpr(prep(kPrepR + 8))},
                           {to_sgpr(prep(kPrepR + 15)), to_sgpr(prep(kPrepR + 16)), to_sgpr(prep(kPrepR + 17))}};
  const double tz = to_sgpr(prep(kPrepT + 2));
  int cnt[4] = {0, 0, 0, 0};
  const int nw = blockDim.x >> 6;  // 4 wavefronts per pair for small batches (latency), 1 for large ones (throughput)
  const float4* mrow = reinterpret_cast<const float4*>(matches) + pair * N;
  // the rows of one correspondence in fp64 (formed where they are used, from the fp32 pixel coordinates)
  auto rows1 = [&](const float4& m, double* A1) {  // view-1 rows, columns 0..2 (column 3 is zero)
    const double x1 = m.x, y1 = m.y;
#pragma unroll
    for (int c = 0; c < 3; ++c) { A1[c] = x1 * Kd[6 + c] - Kd[c]; A1[3 + c] = y1 * Kd[6 + c] - Kd[3 + c]; }
  };
  auto rows2 = [&](const float4& m, const int rr, double* A2) {  // view-2 rows of candidate rr
    const double x2 = m.z, y2 = m.w;
#pragma unroll
    for (int c = 0; c < 4; ++c) { A2[c] = x2 * P2s[rr][8 + c] - P2s[rr][c]; A2[4 + c] = y2 * P2s[rr][8 + c] - P2s[rr][4 + c]; }
  };
  // One DLT per rotation: flipping t negates the 4th column of the view-2 rows, hence the 4th component of the null
  // vector, hence both depths exactly -- candidates (R,t) and (R,-t) are counted from the same triangulation.
  // DLT rows: x*P[2]-P[0], y*P[2]-P[1] for both views (P1 = K [I|0]); the view-1 rows are shared by the two candidates.
  //
  // ---- the fp64 route (rounds 2-3: every correspondence; round 4: only the ambiguous ones, see below): normal matrices of both
  // candidates in fp64, scaled to unit trace (the null vector does not care; the Newton seeds and the fp32 stage want O(1) operands
  // whatever the pixel scale), both smallest eigenvectors to fp32 accuracy (packed), ONE Rayleigh-quotient iteration in fp64 each
  // (cubic convergence: <= 1e-9 from the fp32 vector's 2e-7 median / 1e-4 tail error), division-free depth tests.
  // Round 5: ONE candidate at a time (the packed stage-1 routine instantiated for plain floats), the second candidate's
  // instructions fenced behind the first's, so that the route's working set -- one fp64 normal matrix, not two -- fits the 128
  // registers of the fast path around it.
  auto dlt_fp64 = [&](const float4& m, const bool live) {
    auto candidate = [&](auto rc) {
      constexpr int rr = decltype(rc)::value;
      double Sr[16];
      {
        double A1[6], A2[8];
        rows1(m, A1);
        rows2(m, rr, A2);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = r; c < 4; ++c) Sr[4 * r + c] = A2[r] * A2[c] + A2[4 + r] * A2[4 + c];
        // view-1 contribution to the upper-left 3x3 of A^T A (its rows have no 4th column)
        Sr[0] += A1[0] * A1[0] + A1[3] * A1[3]; Sr[1] += A1[0] * A1[1] + A1[3] * A1[4]; Sr[2] += A1[0] * A1[2] + A1[3] * A1[5];
        Sr[5] += A1[1] * A1[1] + A1[4] * A1[4]; Sr[6] += A1[1] * A1[2] + A1[4] * A1[5]; Sr[10] += A1[2] * A1[2] + A1[5] * A1[5];
      }
      const double itr = rcp_nr<1>(fmax(Sr[0] + Sr[5] + Sr[10] + Sr[15], 1e-30));
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = r; c < 4; ++c) Sr[4 * r + c] *= itr;
      float Sf[16], Xf[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = r; c < 4; ++c) { Sf[4 * r + c] = (float)Sr[4 * r + c]; Sf[4 * c + r] = Sf[4 * r + c]; }
      smallest_eigvec4_pk<float>(Sf, Xf);
      const double x0[4] = {(double)Xf[0], (double)Xf[1], (double)Xf[2], (double)Xf[3]};
      double X[4];
      rqi_refine4(Sr, x0, X);
      // depths z1 = X2 / X3 and z2 = (R_3 . X_012 + t_3 X3) / X3 tested without the division: 0 < z < thr  <=>  z' w > 0 and
      // |z'| < thr |w| for z = z' / w
      const double wq = X[3];
      const double z1n = X[2];
      const double z2n = Rz[rr][0] * X[0] + Rz[rr][1] * X[1] + Rz[rr][2] * X[2] + tz * wq;
      const double thr = (double)depth_thres;
      const double aw = thr * fabs(wq);
      const bool inr = live && (fabs(z1n) < aw) && (fabs(z2n) < aw) && (wq != 0.0);
      const bool s1p = (z1n > 0.0) == (wq > 0.0), s2p = (z2n > 0.0) == (wq > 0.0);
      const bool nz = (z1n != 0.0) && (z2n != 0.0);
      const bool pos = inr && nz && s1p && s2p;    // both depths in (0, thr)
      const bool neg = inr && nz && !s1p && !s2p;  // both in (-thr, 0): the (R, -t) candidate sees them in (0, thr)
      cnt[2 * rr] += __popcll(__ballot(pos));
      cnt[2 * rr + 1] += __popcll(__ballot(neg));
      __builtin_amdgcn_sched_barrier(0);
    };
    candidate(std::integral_constant<int, 0>{});
    candidate(std::integral_constant<int, 1>{});
  };
  // ---- ROUND 4: the fp32 stage decides on its own wherever it safely can.  Every correspondence goes through the FAST PATH -- the
  // whole eigenproblem of both candidates in packed fp32: normal matrices from the fp32-rounded rows, stage 1 -- and the depth
  // tests on that vector.  Measured (scripts/proto_cheirality_margin.py, a numpy twin of this path; 6 scene families, 288 000 tests
  // against numpy.linalg.eigh): the fp32 vector decides every test like the fp64 one, and its error obeys
  //     |x~ - x| <= 2.9e-8 / ((lam1 - lam4)(lam2 - lam4)(lam3 - lam4))        (unit trace, unit vectors; 99.9th percentile 1.5e-8)
  // so a correspondence is AMBIGUOUS when one of its test quantities lies within delta = 8 x 4e-8 / gapprod (at least 1e-6) times
  // |x| of its bound (0.1-2 % of them, depending on how good E is).  The ambiguous ones are not decided here: their indices go to
  // a per-wavefront queue in LDS and take the fp64 route above 64 at a time -- a full wavefront of them, not the group of 64 they
  // happened to sit in (forcing the whole group through the fp64 stage sent 39 % of the groups of the benchmark's config 5 there:
  // 99 us against 83 us for the fast path alone and 124 us for the fp64 route alone, scripts/ab_cheirality.sh).
  int qn = 0;  // queued indices of this wavefront (uniform)
  int* q = queue + wave * kCheirQueue;
  // the next group's correspondence is loaded (index clamped, no branch) before the current one is triangulated
  float4 mnext = mrow[min(wave * WAVE + lane, N - 1)];
  for (int base = wave * WAVE; base < N; base += nw * WAVE) {
    const int i = base + lane;
    const bool live = i < N;
    const float4 m = mnext;
    mnext = mrow[min(i + nw * WAVE, N - 1)];
    if constexpr (FP64_ONLY) {
      dlt_fp64(m, live);
      continue;
    }
    f2 Xp[4];
    bool amb_lane = false;
    bool pos_f[2], neg_f[2];
    {
      float A1f[6];
      {
        double A1[6];
        rows1(m, A1);
#pragma unroll
        for (int c = 0; c < 6; ++c) A1f[c] = (float)A1[c];
      }
      const float s00 = A1f[0] * A1f[0] + A1f[3] * A1f[3], s01 = A1f[0] * A1f[1] + A1f[3] * A1f[4], s02 = A1f[0] * A1f[2] + A1f[3] * A1f[5];
      const float s11 = A1f[1] * A1f[1] + A1f[4] * A1f[4], s12 = A1f[1] * A1f[2] + A1f[4] * A1f[5], s22 = A1f[2] * A1f[2] + A1f[5] * A1f[5];
      f2 A2p[8];
      {
        double Aa[8], Ab[8];
        rows2(m, 0, Aa);
        rows2(m, 1, Ab);
#pragma unroll
        for (int c = 0; c < 8; ++c) A2p[c] = f2{(float)Aa[c], (float)Ab[c]};
      }
      f2 Sp[16];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = r; c < 4; ++c) Sp[4 * r + c] = A2p[r] * A2p[c] + A2p[4 + r] * A2p[4 + c];
      Sp[0] += f2s(s00); Sp[1] += f2s(s01); Sp[2] += f2s(s02); Sp[5] += f2s(s11); Sp[6] += f2s(s12); Sp[10] += f2s(s22);
      const f2 itr = pk_rcp(pk_max(Sp[0] + Sp[5] + Sp[10] + Sp[15], f2s(1e-30f)));
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = r; c < 4; ++c) { Sp[4 * r + c] *= itr; Sp[4 * c + r] = Sp[4 * r + c]; }
      f2 gp;
      smallest_eigvec4_pk(Sp, Xp, &gp);
      const float thrf = depth_thres;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const float X0 = Xp[0][rr], X1 = Xp[1][rr], X2 = Xp[2][rr], X3 = Xp[3][rr];
        const float nrm = hw_sqrt(X0 * X0 + X1 * X1 + X2 * X2 + X3 * X3);
        // R and t live in scalar registers as doubles: used as they are (fp32 copies would be loop-invariant VECTOR registers)
        const float z2n = (float)(Rz[rr][0] * (double)X0 + Rz[rr][1] * (double)X1 + Rz[rr][2] * (double)X2 + tz * (double)X3);
        const float aw = thrf * fabsf(X3);
        const float g = gp[rr];
        const float dl = fmaxf(3.2e-7f * hw_rcp(fmaxf(g, 1e-30f)), 1e-6f) * nrm;
        // (1 + thr) dl and (2 + thr) dl as FMAs on the scalar thr: no loop-invariant vector register for the sums; `|`, not `||`:
        // mask arithmetic on the scalar unit instead of a branch per term
        const float m1 = fmaf(thrf, dl, dl), m2 = fmaf(thrf, dl, dl + dl);
        const bool amb = !(g > 0.0f) | !(nrm > 0.0f) | !(nrm < 1e30f) | (fabsf(X3) < dl) | (fabsf(X2) < dl) | (fabsf(z2n) < dl + dl) |
                         (fabsf(fabsf(X2) - aw) < m1) | (fabsf(fabsf(z2n) - aw) < m2);
        amb_lane = amb_lane | (amb & live);
        const bool inr = live & (fabsf(X2) < aw) & (fabsf(z2n) < aw);
        const bool s1p = (X2 > 0.0f) == (X3 > 0.0f), s2p = (z2n > 0.0f) == (X3 > 0.0f);
        pos_f[rr] = inr & s1p & s2p;      // both depths in (0, thr); zeros are ambiguous and never decided here
        neg_f[rr] = inr & !s1p & !s2p;    // both in (-thr, 0): the (R, -t) candidate sees them in (0, thr)
      }
    }
#if defined(DFEPE_CHEIR_ALWAYS_FAST)   // A/B timing builds only (scripts/ab_cheirality.sh): the lower bound of the adaptive kernel
    amb_lane = false;                  // (its upper bound is the FP64_ONLY instantiation, DFEPE_CHEIR_FP64_ONLY at run time)
#endif
    const unsigned long long amask = __ballot(amb_lane);
    // a lane that is safe for BOTH candidates counts now; an ambiguous lane counts nothing here (both candidates again in fp64)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      cnt[2 * rr] += __popcll(__ballot(pos_f[rr] & !amb_lane));
      cnt[2 * rr + 1] += __popcll(__ballot(neg_f[rr] & !amb_lane));
    }
    if (amask != 0ull) {  // wave-uniform
      const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(amask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)amask, 0u));
      if (amb_lane) q[qn + pos] = i;
      qn += __popcll(amask);
      if (qn >= WAVE) {  // a full wavefront of ambiguous correspondences: through the fp64 route, the rest moves to the front
        wave_sync();
        const int qi = q[lane];
        const int rest = qn - WAVE;
        const int carry = (lane < rest) ? q[WAVE + lane] : 0;
        wave_sync();
        if (lane < rest) q[lane] = carry;
        qn = rest;
        dlt_fp64(mrow[qi], true);
      }
    }
  }
  if (qn > 0) {  // what is left in the queue (uniform)
    wave_sync();
    const bool live = lane < qn;
    const int qi = live ? q[lane] : 0;
    dlt_fp64(mrow[qi], live);
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) wcnt[wave][c] = cnt[c];
  }
  __syncthreads();
  if (wave != 0) return;
  if (nw > 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int tsum = 0;
#pragma unroll
      for (int w = 0; w < 8; ++w) tsum += (w < nw) ? wcnt[w][c] : 0;  // fully unrolled: a runtime trip count here turns cnt[] into
      cnt[c] = tsum;                                                   // a dynamically indexed array, which hipcc parks in LDS
    }
  }
  int win = 0, best = cnt[0];  // `best` instead of best: a runtime index would turn cnt[] into an array in (LDS-promoted) memory
#pragma unroll
  for (int c = 1; c < 4; ++c)
    if (cnt[c] > best) { best = cnt[c]; win = c; }  // first maximum, like max(enumerate(...)) (utils_F.py:730)
  if (lane == 0) {
    if (counts != nullptr) {
#pragma unroll
      for (int c = 0; c < 4; ++c) counts[pair * 4 + c] = cnt[c];
    }
    if (winner != nullptr) winner[pair] = (best > 0) ? win : -1;
    // camera motion = inverse of [R|t]: [R^T | -R^T t]   (utils_misc._inv_Rt, utils_misc.py:115-121)
    double Rc[9], tw[3];  // the winner's rotation and the translation
#pragma unroll
    for (int k = 0; k < 9; ++k) Rc[k] = prep(kPrepR + 9 * (win >> 1) + k);
#pragma unroll
    for (int k = 0; k < 3; ++k) tw[k] = prep(kPrepT + k);
    const double sg = (win & 1) ? -1.0 : 1.0;
    float* dst = Rt_cam + pair * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dst[4 * r + c] = (best > 0) ? (float)Rc[3 * c + r] : 0.0f;
      const double tc = -(Rc[r] * tw[0] + Rc[3 + r] * tw[1] + Rc[6 + r] * tw[2]) * sg;
      dst[4 * r + 3] = (best > 0) ? (float)tc : 0.0f;
    }
  }
}

}  // namespace
