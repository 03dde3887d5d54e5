// Scalar math helpers shared by the dfepe kernels: fp64 reciprocal / square root from the fp32 hardware seeds, the
// design-matrix row of one correspondence, 3x3 products and the 3x3 Jacobi SVDs.  Pure per-lane C++: the only
// primitives it needs are hw_rsq / hw_rcp (v_rsq_f32 / v_rcp_f32), which rowgroup.h provides.
#pragma once
#include <math.h>

// fp64 reciprocal square root / reciprocal from the fp32 hardware approximation plus Newton-Raphson in fp64:
// one step takes the 1e-7 seed to ~2e-14 relative, two steps to full fp64.  ~6-10 instructions instead of the ~30
// of the IEEE sqrt/div expansions.  Arguments outside the fp32 range fall back to the exact routines.
__device__ __forceinline__ double fast_rsqrt(double x) {
  if (!(x > 1e-30 && x < 1e30)) return 1.0 / sqrt(x);
  double y = (double)hw_rsq((float)x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
  const double ax = fabs(x);
  if (!(ax > 1e-30 && ax < 1e30)) return 1.0 / x;
  double y = (double)hw_rcp((float)x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}
__device__ __forceinline__ double fast_sqrt(double x) { return (x > 0.0) ? x * fast_rsqrt(x) : 0.0; }

// Branch-free variants: the fp64 hardware seed (v_rsq_f64 / v_rcp_f64: ~2^-23 relative on gfx950, scripts/ubench/seed_precision.hip)
// plus Newton-Raphson in fp64.  STEPS = 1 gives ~1e-14 relative, 2 full fp64.  One 16-cycle instruction for the seed instead of the
// fp32 detour (convert, clamp into the fp32 range, v_rsq_f32, convert back: five instructions of 5-8 cycles each).  No range test,
// hence no divergent fall-back path.  SAFE (default): a zero (or a negative rounding residue) operand gives finite garbage times
// zero, never NaN -- one v_max_f64 on the operand; callers whose operand is known to be positive and normal pass SAFE = false.
template <int STEPS, bool SAFE = true>
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = hw_rsq64(SAFE ? fmax(x, 1e-290) : x);
  const double h = -0.5 * x;
#pragma unroll
  for (int k = 0; k < STEPS; ++k) y = fma(y, fma(h, y * y, 0.5), y);  // y (1.5 - x y^2 / 2) with inline constants only
  return y;
}
template <int STEPS, bool SAFE = true>
__device__ __forceinline__ double sqrt_nr(double x) { return x * rsqrt_nr<STEPS, SAFE>(x); }  // 0 for x = 0
// SAFE: |x| is raised to 1e-290 (sign kept): 1 / 0 is a huge finite number, not NaN after the Newton step
template <int STEPS, bool SAFE = true>
__device__ __forceinline__ double rcp_nr(double x) {
  double y = hw_rcp64(SAFE ? copysign(fmax(fabs(x), 1e-290), x) : x);
#pragma unroll
  for (int k = 0; k < STEPS; ++k) y = fma(y, fma(-x, y, 1.0), y);  // y (2 - x y)
  return y;
}

// One correspondence in image-size-normalised homogeneous coordinates.
struct Pt {
  float x1, y1, z1, x2, y2, z2;
};

// Unit row of the design matrix: ph = p / max(|p|, 1e-12) with p = [x2~ a, y2~ a, a], a = (x1~, y1~, z1)
// (DeepFNet.py:203-212), fp64.  Returns false (and a zero row) for non-finite rows.
__device__ __forceinline__ bool unit_row(const Pt& p, double s1, double c1x, double c1y, double s2, double c2x,
                                         double c2y, double* ph) {
  const double z1 = p.z1, z2 = p.z2;
  const double a0 = s1 * ((double)p.x1 - c1x * z1), a1 = s1 * ((double)p.y1 - c1y * z1), a2 = z1;
  const double b0 = s2 * ((double)p.x2 - c2x * z2), b1 = s2 * ((double)p.y2 - c2y * z2);
  const double n2 = (a0 * a0 + a1 * a1 + a2 * a2) * (b0 * b0 + b1 * b1 + 1.0);
  const bool ok = n2 < 1e300;
  const double inv = ok ? ((n2 > 1e-24) ? fast_rsqrt(n2) : 1e12) : 0.0;  // 1 / max(|p|, 1e-12)
  const double ia0 = ok ? a0 * inv : 0.0, ia1 = ok ? a1 * inv : 0.0, ia2 = ok ? a2 * inv : 0.0;
  ph[0] = b0 * ia0; ph[1] = b0 * ia1; ph[2] = b0 * ia2;
  ph[3] = b1 * ia0; ph[4] = b1 * ia1; ph[5] = b1 * ia2;
  ph[6] = ia0;      ph[7] = ia1;      ph[8] = ia2;
  if (!ok) {
#pragma unroll
    for (int k = 0; k < 9; ++k) ph[k] = 0.0;
  }
  return ok;
}

// 3x3 helpers on row-major arrays -------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void mat3_mul(const T* A, const T* B, T* C) {  // C = A B
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
template <typename T>
__device__ __forceinline__ void mat3_mul_tn(const T* A, const T* B, T* C) {  // C = A^T B
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}
template <typename T>
__device__ __forceinline__ void mat3_mul_nt(const T* A, const T* B, T* C) {  // C = A B^T
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c * 3] + A[r * 3 + 1] * B[c * 3 + 1] + A[r * 3 + 2] * B[c * 3 + 2];
}

// fp32 one-sided Jacobi SVD of a 3x3 matrix on the hardware transcendentals (v_rsq_f32 / v_rcp_f32, ~1 ulp):
// F = U diag(S) V^T, S descending, singular vectors in the columns of the row-major U, V; ~50 instructions per rotation with two
// dependent transcendentals
// (r = rsq(d^2+b^2); x = (1+|d| r)/2; y = rsq(x); c = x y; s = sgn(d) b r y / 2) and a division-free skip test.
// Used for the rank-2 step of the solver, where the dropped triplet is re-measured in fp64 afterwards.
__device__ inline void svd3_fast(const float* F, float* U, float* S, float* V) {
  float G[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    G[i] = F[i];
    V[i] = (i % 4 == 0) ? 1.0f : 0.0f;
  }
  const float tol2 = 1e-14f;  // (1e-7)^2: columns are orthogonal to fp32 round-off
  for (int sweep = 0; sweep < 10; ++sweep) {
    bool any = false, big = false;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = (pq == 2) ? 1 : 0;
      const int q = (pq == 0) ? 1 : 2;
      const float al = fmaf(G[p], G[p], fmaf(G[3 + p], G[3 + p], G[6 + p] * G[6 + p]));
      const float be = fmaf(G[q], G[q], fmaf(G[3 + q], G[3 + q], G[6 + q] * G[6 + q]));
      const float ga = fmaf(G[p], G[q], fmaf(G[3 + p], G[3 + q], G[6 + p] * G[6 + q]));
      const float gg = ga * ga, ab = al * be;
      const bool rot = gg > tol2 * ab;
      any = any || rot;
      big = big || (gg > 1e-9f * ab);
      const float d = be - al, b = 2.0f * ga;
      const float r = hw_rsq(fmaf(d, d, b * b));
      const float x = fmaf(0.5f * fabsf(d), r, 0.5f);
      const float y = hw_rsq(x);
      const float c = rot ? x * y : 1.0f;
      const float s = rot ? copysignf(0.5f, d) * b * r * y : 0.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float gp = G[3 * k + p], gq = G[3 * k + q];
        G[3 * k + p] = fmaf(c, gp, -s * gq);
        G[3 * k + q] = fmaf(s, gp, c * gq);
        const float vp = V[3 * k + p], vq = V[3 * k + q];
        V[3 * k + p] = fmaf(c, vp, -s * vq);
        V[3 * k + q] = fmaf(s, vp, c * vq);
      }
    }
    // cosines below 3e-5 are squared by the sweep that just ran (quadratic convergence): nothing left above tol
    if (!big) break;
  }
  float n2[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) n2[k] = fmaf(G[k], G[k], fmaf(G[3 + k], G[3 + k], G[6 + k] * G[6 + k]));
#define DFEPE_SWAPCOL(a, b)                                      \
  if (n2[a] < n2[b]) {                                           \
    float tn = n2[a]; n2[a] = n2[b]; n2[b] = tn;                 \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {              \
      float tg = G[3 * r + a]; G[3 * r + a] = G[3 * r + b]; G[3 * r + b] = tg; \
      float tv = V[3 * r + a]; V[3 * r + a] = V[3 * r + b]; V[3 * r + b] = tv; \
    }                                                            \
  }
  DFEPE_SWAPCOL(0, 1)
  DFEPE_SWAPCOL(1, 2)
  DFEPE_SWAPCOL(0, 1)
#undef DFEPE_SWAPCOL
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float inv = hw_rsq(fmaxf(n2[k], 1e-36f));
    U[k] = G[k] * inv; U[3 + k] = G[3 + k] * inv; U[6 + k] = G[6 + k] * inv;
    S[k] = n2[k] * inv;
  }
  S[2] = (n2[2] > 0.0f) ? n2[2] * hw_rsq(n2[2]) : 0.0f;
  {
    const float dt = U[0] * U[1] + U[3] * U[4] + U[6] * U[7];
    const float a0 = U[1] - dt * U[0], a1 = U[4] - dt * U[3], a2 = U[7] - dt * U[6];
    const float inv = hw_rsq(fmaxf(a0 * a0 + a1 * a1 + a2 * a2, 1e-36f));
    U[1] = a0 * inv; U[4] = a1 * inv; U[7] = a2 * inv;
  }
  const float c0 = U[3] * U[7] - U[6] * U[4];
  const float c1 = U[6] * U[1] - U[0] * U[7];
  const float c2 = U[0] * U[4] - U[3] * U[1];
  const float sg = (c0 * G[2] + c1 * G[5] + c2 * G[8] < 0.0f) ? -1.0f : 1.0f;
  U[2] = sg * c0; U[5] = sg * c1; U[8] = sg * c2;
}

// ---- smallest singular triplet of a 3x3 matrix in closed form (fp64) -----------------------------------------------
// The rank-2 step of the solver (DeepFNet.py:236-237) only drops the smallest singular triplet: F' = F - s3 u3 v3^T.
// v3 / u3 are the null vectors of F^T F - lam I / F F^T - lam I, lam = s3^2 the smallest root of the characteristic
// cubic: Newton from 0 converges monotonically from below (all roots real, p > 0, p' < 0, p'' > 0 left of the smallest
// one), and the null vector of a rank-2 symmetric 3x3 is the largest cross product of two of its rows.  s3 = u3^T F v3 is
// stationary w.r.t. first-order errors of the vectors.  ~250 straight-line instructions instead of a Jacobi SVD.
// ROBUST: also right when the matrix has rank 1 (every cross product is round-off): then any vector orthogonal to its dominant
// row is a null vector.  The fit never needs it (its matrix comes from a generic eigenvector); the pose path takes whatever E
// it is given.
template <bool ROBUST = false>
__device__ __forceinline__ void sym3_null_vector(double c00, double c01, double c02, double c11, double c12, double c22, double* n) {
  // rows r0 = (c00,c01,c02), r1 = (c01,c11,c12), r2 = (c02,c12,c22)
  const double a0 = c01 * c12 - c02 * c11, a1 = c02 * c01 - c00 * c12, a2 = c00 * c11 - c01 * c01;  // r0 x r1
  const double b0 = c01 * c22 - c02 * c12, b1 = c02 * c02 - c00 * c22, b2 = c00 * c12 - c01 * c02;  // r0 x r2
  const double d0 = c11 * c22 - c12 * c12, d1 = c12 * c02 - c01 * c22, d2 = c01 * c12 - c11 * c02;  // r1 x r2
  const double na = a0 * a0 + a1 * a1 + a2 * a2, nb = b0 * b0 + b1 * b1 + b2 * b2, nd = d0 * d0 + d1 * d1 + d2 * d2;
  double x0 = a0, x1 = a1, x2 = a2, nx = na;
  if (nb > nx) { x0 = b0; x1 = b1; x2 = b2; nx = nb; }
  if (nd > nx) { x0 = d0; x1 = d1; x2 = d2; nx = nd; }
  const double inv = (nx > 0.0) ? rsqrt_nr<2, false>(nx) : 0.0;
  n[0] = x0 * inv; n[1] = x1 * inv; n[2] = x2 * inv;
  if (!(nx > 0.0)) { n[0] = 0.0; n[1] = 0.0; n[2] = 1.0; }  // the zero matrix: any unit vector
  if constexpr (ROBUST) {
    const double r0 = c00 * c00 + c01 * c01 + c02 * c02, r1 = c01 * c01 + c11 * c11 + c12 * c12, r2 = c02 * c02 + c12 * c12 + c22 * c22;
    const double rmax = fmax(r0, fmax(r1, r2));
    if (nx <= 1e-24 * rmax * rmax && rmax > 0.0) {  // |row x row| below 1e-12 |row|^2: rank 1 to working precision
      const bool k0 = r0 >= r1 && r0 >= r2, k1 = !k0 && r1 >= r2;
      const double d0 = k0 ? c00 : (k1 ? c01 : c02), d1 = k0 ? c01 : (k1 ? c11 : c12), d2 = k0 ? c02 : (k1 ? c12 : c22);
      // e_k x d for the axis d is least aligned with
      const double ax = fabs(d0), ay = fabs(d1), az = fabs(d2);
      const bool kx = ax <= ay && ax <= az, ky = !kx && ay <= az;
      const double y0 = kx ? 0.0 : (ky ? d2 : -d1), y1 = kx ? -d2 : (ky ? 0.0 : d0), y2 = kx ? d1 : (ky ? -d0 : 0.0);
      const double iy = rsqrt_nr<2>(fmax(y0 * y0 + y1 * y1 + y2 * y2, 1e-300));
      n[0] = y0 * iy; n[1] = y1 * iy; n[2] = y2 * iy;
    }
  }
}

// F row-major, assumed O(1) in magnitude (the solver passes a unit-Frobenius F).  Returns s3 >= 0, unit u3, v3.
template <bool ROBUST = false>
__device__ __forceinline__ void smallest_singular_triplet3(const double* F, double* u3, double* v3, double& s3) {
  // B = F^T F
  const double b00 = F[0] * F[0] + F[3] * F[3] + F[6] * F[6], b01 = F[0] * F[1] + F[3] * F[4] + F[6] * F[7];
  const double b02 = F[0] * F[2] + F[3] * F[5] + F[6] * F[8], b11 = F[1] * F[1] + F[4] * F[4] + F[7] * F[7];
  const double b12 = F[1] * F[2] + F[4] * F[5] + F[7] * F[8], b22 = F[2] * F[2] + F[5] * F[5] + F[8] * F[8];
  const double c2 = b00 + b11 + b22;
  const double c1 = (b00 * b11 - b01 * b01) + (b00 * b22 - b02 * b02) + (b11 * b22 - b12 * b12);
  const double c0 = b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02);
  double x = 0.0;
  const double tol = 1e-17 * c2;
  auto newton = [&]() {
    const double p = fma(fma(c2 - x, x, -c1), x, c0);            // -x^3 + c2 x^2 - c1 x + c0
    const double dp = fma(fma(-3.0, x, 2.0 * c2), x, -c1);       // p'(x) < 0 left of the smallest root
    const double dx = (dp < 0.0) ? -p * rcp_nr<2, false>(dp) : 0.0;
    return (dx > tol) ? dx : 0.0;                                 // 0 also on NaN; p <= 0: at (or rounded past) the root
  };
  // Four steps without a branch: the relative error goes (s3/s2)^2 -> its square -> ..., so a generic matrix is converged after
  // three; the loop behind them (entered only while the fourth step still moved x) covers s3 ~ s2.  A data-dependent exit after
  // every step costs more in exec-mask bookkeeping than the arithmetic it skips.
  double dx = 0.0;
#pragma unroll
  for (int it = 0; it < 4; ++it) { dx = newton(); x += dx; }
  for (int it = 4; it < 12 && dx > 0.0; ++it) { dx = newton(); x += dx; }
  sym3_null_vector<ROBUST>(b00 - x, b01, b02, b11 - x, b12, b22 - x, v3);
  // D = F F^T
  const double d00 = F[0] * F[0] + F[1] * F[1] + F[2] * F[2], d01 = F[0] * F[3] + F[1] * F[4] + F[2] * F[5];
  const double d02 = F[0] * F[6] + F[1] * F[7] + F[2] * F[8], d11 = F[3] * F[3] + F[4] * F[4] + F[5] * F[5];
  const double d12 = F[3] * F[6] + F[4] * F[7] + F[5] * F[8], d22 = F[6] * F[6] + F[7] * F[7] + F[8] * F[8];
  sym3_null_vector<ROBUST>(d00 - x, d01, d02, d11 - x, d12, d22 - x, u3);
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) s += u3[r] * F[3 * r + c] * v3[c];
  if (s < 0.0) { s = -s; u3[0] = -u3[0]; u3[1] = -u3[1]; u3[2] = -u3[2]; }
  s3 = s;
}

// Unit vectors a1, a2 completing n (unit) to a right-handed orthonormal basis (a1, a2, n): a1 = e_k x n normalised with k the
// axis n is least aligned with (|e_k x n|^2 >= 2/3), a2 = n x a1.  Branch-free.
__device__ __forceinline__ void orthonormal_complement3(const double* n, double* a1, double* a2) {
  const double ax = fabs(n[0]), ay = fabs(n[1]), az = fabs(n[2]);
  const bool kx = ax <= ay && ax <= az, ky = !kx && ay <= az;
  // e_x x n = (0, -n2, n1);  e_y x n = (n2, 0, -n0);  e_z x n = (-n1, n0, 0)
  const double c0 = kx ? 0.0 : (ky ? n[2] : -n[1]);
  const double c1 = kx ? -n[2] : (ky ? 0.0 : n[0]);
  const double c2 = kx ? n[1] : (ky ? -n[0] : 0.0);
  const double inv = rsqrt_nr<2>(c0 * c0 + c1 * c1 + c2 * c2);
  a1[0] = c0 * inv; a1[1] = c1 * inv; a1[2] = c2 * inv;
  a2[0] = n[1] * a1[2] - n[2] * a1[1]; a2[1] = n[2] * a1[0] - n[0] * a1[2]; a2[2] = n[0] * a1[1] - n[1] * a1[0];
}

// Closed-form SVD of a 3x3 matrix in fp64: F = U diag(S) V^T, S descending and >= 0, singular vectors in the columns of the
// row-major U, V, u3 oriented along F v3 -- without a Jacobi iteration: the smallest singular triplet as
// above, then the 2x2 problem in the orthogonal complements of u3 / v3, whose one Jacobi rotation is exact.  ~450 straight-line
// instructions against ~170 per sweep of the one-sided Jacobi it replaced (4-6 sweeps).  Repeated s1 = s2 (essential matrices) is fine:
// any orthonormal pair of the plane is a valid answer, and U, V are produced consistently (u_i = F v_i / s_i).
__device__ inline void svd3_closed(const double* Fin, double* U, double* S, double* V) {
  // exact power-of-two prescale to max|F| in [0.5, 1): the vectors do not depend on the scale, S is scaled back exactly, and the
  // closed forms / Newton seeds see O(1) entries whatever the scale of the input (an essential matrix times 1e-6 used to leave
  // their range: scripts/stress_pose.py)
  double big = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) big = fmax(big, fabs(Fin[i]));
  int ex = 0;
  if (big > 0.0 && big < 1e300) (void)frexp(big, &ex);
  double F[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) F[i] = ldexp(Fin[i], -ex);
  double u3[3], v3[3], s3;
  smallest_singular_triplet3<true>(F, u3, v3, s3);
  double a1[3], a2[3], b1[3], b2[3];
  orthonormal_complement3(v3, a1, a2);
  orthonormal_complement3(u3, b1, b2);
  // G = F [a1 a2] (3x2), B = [b1 b2]^T G (2x2)
  double g1[3], g2[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    g1[r] = F[3 * r] * a1[0] + F[3 * r + 1] * a1[1] + F[3 * r + 2] * a1[2];
    g2[r] = F[3 * r] * a2[0] + F[3 * r + 1] * a2[1] + F[3 * r + 2] * a2[2];
  }
  const double B00 = b1[0] * g1[0] + b1[1] * g1[1] + b1[2] * g1[2], B01 = b1[0] * g2[0] + b1[1] * g2[1] + b1[2] * g2[2];
  const double B10 = b2[0] * g1[0] + b2[1] * g1[1] + b2[2] * g1[2], B11 = b2[0] * g2[0] + b2[1] * g2[1] + b2[2] * g2[2];
  // one rotation of the columns of B diagonalises B^T B = [[al, ga], [ga, be]]: r = 1/h, h = sqrt(d^2 + b^2);
  // x = (1 + |d|/h)/2 = cos^2; c = sqrt(x), s = sgn(d) b / (2 h c)
  const double al = B00 * B00 + B10 * B10, be = B01 * B01 + B11 * B11, ga = B00 * B01 + B10 * B11;
  const double d = be - al, b = 2.0 * ga;
  const double h2 = d * d + b * b;
  const bool rot = h2 > 0.0;
  const double rh = rot ? rsqrt_nr<2>(h2) : 0.0;
  const double x = 0.5 + 0.5 * fabs(d) * rh;  // cos^2; 1/2 .. 1
  const double y = rsqrt_nr<2>(x);
  const double c = rot ? x * y : 1.0;
  const double sn = rot ? copysign(0.5, d) * b * rh * y : 0.0;
  // rotated columns: first = c col0 - s col1, second = s col0 + c col1
  double p0x = c * B00 - sn * B01, p0y = c * B10 - sn * B11;
  double p1x = sn * B00 + c * B01, p1y = sn * B10 + c * B11;
  double q0x = c, q0y = -sn, q1x = sn, q1y = c;  // the corresponding right vectors in the (a1, a2) basis
  double n0 = p0x * p0x + p0y * p0y, n1 = p1x * p1x + p1y * p1y;
  if (n0 < n1) {  // descending
    double t;
    t = p0x; p0x = p1x; p1x = t; t = p0y; p0y = p1y; p1y = t; t = n0; n0 = n1; n1 = t;
    t = q0x; q0x = q1x; q1x = t; t = q0y; q0y = q1y; q1y = t;
  }
  const double s1 = sqrt_nr<2>(n0), s2 = sqrt_nr<2>(n1);
  // left vectors in the (b1, b2) basis: the first from its column, the second perpendicular to it, oriented along its column
  // (well defined even when s2 = 0)
  const double i1 = (n0 > 0.0) ? rsqrt_nr<2>(n0) : 0.0;
  const double l0x = (n0 > 0.0) ? p0x * i1 : 1.0, l0y = (n0 > 0.0) ? p0y * i1 : 0.0;
  const double sg = (-l0y * p1x + l0x * p1y < 0.0) ? -1.0 : 1.0;
  const double l1x = -sg * l0y, l1y = sg * l0x;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    U[3 * r + 0] = b1[r] * l0x + b2[r] * l0y;
    U[3 * r + 1] = b1[r] * l1x + b2[r] * l1y;
    U[3 * r + 2] = u3[r];
    V[3 * r + 0] = a1[r] * q0x + a2[r] * q0y;
    V[3 * r + 1] = a1[r] * q1x + a2[r] * q1y;
    V[3 * r + 2] = v3[r];
  }
  S[0] = ldexp(s1, ex); S[1] = ldexp(s2, ex); S[2] = ldexp(s3, ex);
}

// y = C^+ x for a symmetric 3x3 C (6 distinct entries) with unit null vector n: (C + n n^T)^-1 (x - n (n.x)) by the
// adjugate.  A second (near-)null direction gives a large-but-finite result, never NaN.
__device__ __forceinline__ void sym3_pinv_apply(double c00, double c01, double c02, double c11, double c12, double c22,
                                                const double* n, const double* x, double* y) {
  const double nx = n[0] * x[0] + n[1] * x[1] + n[2] * x[2];
  const double r0 = x[0] - nx * n[0], r1 = x[1] - nx * n[1], r2 = x[2] - nx * n[2];
  const double m00 = c00 + n[0] * n[0], m01 = c01 + n[0] * n[1], m02 = c02 + n[0] * n[2];
  const double m11 = c11 + n[1] * n[1], m12 = c12 + n[1] * n[2], m22 = c22 + n[2] * n[2];
  const double k00 = m11 * m22 - m12 * m12, k01 = m02 * m12 - m01 * m22, k02 = m01 * m12 - m02 * m11;
  const double k11 = m00 * m22 - m02 * m02, k12 = m01 * m02 - m00 * m12, k22 = m00 * m11 - m01 * m01;
  double det = m00 * k00 + m01 * k01 + m02 * k02;
  det = (fabs(det) < 1e-30) ? ((det < 0.0) ? -1e-30 : 1e-30) : det;
  const double idet = rcp_nr<2>(det);
  y[0] = (k00 * r0 + k01 * r1 + k02 * r2) * idet;
  y[1] = (k01 * r0 + k11 * r1 + k12 * r2) * idet;
  y[2] = (k02 * r0 + k12 * r1 + k22 * r2) * idet;
}
