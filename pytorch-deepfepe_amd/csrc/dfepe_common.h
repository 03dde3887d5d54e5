// Shared device helpers for the dfepe HIP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dfepe.h"

// ---- layout of the per-pair `save` record (DFEPE_SAVE_FLOATS floats) -----------------------
#define SV_T1 0     // Hartley transform of image 1: s, cx, cy   (T = [[s,0,-s cx],[0,s,-s cy],[0,0,1]])
#define SV_T2 3     // Hartley transform of image 2
#define SV_LAM 6    // 9 eigenvalues of X^T X (Jacobi order, unsorted)
#define SV_Q 15     // 81: Q[k*9+c] = component c of eigenvector k
#define SV_KMIN 96  // index of the smallest eigenvalue (stored as float)
#define SV_SIGN 97  // +-1: orientation applied to Q[kmin] to get f
#define SV_U3 98    // 3x3 SVD of F = reshape(f): U row-major (columns = left singular vectors)
#define SV_S3 107   // singular values, descending
#define SV_V3 110   // V row-major (columns = right singular vectors)

#define WAVE 64

// Wave-wide sums without the LDS pipe: four DPP steps (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) leave the
// sum of each 16-lane row in all of its lanes; the four row sums are then combined through v_readlane.  ~12 VALU/SALU
// instructions and ~60 cycles of latency instead of six dependent ds_bpermute round trips; the result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  union { double d; int i[2]; } a, r;
  a.d = v;
  r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xf, 0xf, false);
  r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xf, 0xf, false);
  return r.d;
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);  // row_half_mirror
  v += dpp_f32<0x140>(v);  // row_mirror
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  union { double d; int i[2]; } a, r[4];
  a.d = v;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r[k].i[0] = __builtin_amdgcn_readlane(a.i[0], 16 * k);
    r[k].i[1] = __builtin_amdgcn_readlane(a.i[1], 16 * k);
  }
  return (r[0].d + r[1].d) + (r[2].d + r[3].d);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Move a wave-uniform value into scalar registers (frees VGPRs; the value must be identical in all lanes).
__device__ __forceinline__ double to_sgpr(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}
__device__ __forceinline__ float to_sgpr(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// fp64 reciprocal square root / reciprocal from the fp32 hardware approximation plus Newton-Raphson in fp64:
// one step takes the 1e-7 seed to ~2e-14 relative, two steps to full fp64.  ~6-10 instructions instead of the ~30
// of the IEEE sqrt/div expansions.  Arguments outside the fp32 range fall back to the exact routines.
__device__ __forceinline__ double fast_rsqrt(double x) {
  if (!(x > 1e-30 && x < 1e30)) return 1.0 / sqrt(x);
  double y = (double)__builtin_amdgcn_rsqf((float)x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
  const double ax = fabs(x);
  if (!(ax > 1e-30 && ax < 1e30)) return 1.0 / x;
  double y = (double)__builtin_amdgcn_rcpf((float)x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}
__device__ __forceinline__ double fast_sqrt(double x) { return (x > 0.0) ? x * fast_rsqrt(x) : 0.0; }

// Compiler-level fence between wave-synchronous LDS phases.  A single wavefront's LDS operations
// execute in issue order, so no s_barrier is needed; this only stops the compiler from moving
// memory operations across the phase boundary.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One correspondence in image-size-normalised homogeneous coordinates.
struct Pt {
  float x1, y1, z1, x2, y2, z2;
};

template <bool RAW>
__device__ __forceinline__ Pt global_point(const float* __restrict__ pts1, const float* __restrict__ pts2, size_t pair,
                                           int i, int N, float hw_sx, float hw_sy) {
  Pt p;
  if (RAW) {
    const float4 m = reinterpret_cast<const float4*>(pts1)[pair * N + i];
    p.x1 = fmaf(m.x, hw_sx, -1.0f); p.y1 = fmaf(m.y, hw_sy, -1.0f); p.z1 = 1.0f;
    p.x2 = fmaf(m.z, hw_sx, -1.0f); p.y2 = fmaf(m.w, hw_sy, -1.0f); p.z2 = 1.0f;
  } else {
    const float* a = pts1 + (pair * N + i) * 3;
    const float* b = pts2 + (pair * N + i) * 3;
    p.x1 = a[0]; p.y1 = a[1]; p.z1 = a[2]; p.x2 = b[0]; p.y2 = b[1]; p.z2 = b[2];
  }
  return p;
}

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
// same contract as svd3<float> below, ~50 instructions per rotation with two dependent transcendentals
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
      const float r = __builtin_amdgcn_rsqf(fmaf(d, d, b * b));
      const float x = fmaf(0.5f * fabsf(d), r, 0.5f);
      const float y = __builtin_amdgcn_rsqf(x);
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
    const float inv = __builtin_amdgcn_rsqf(fmaxf(n2[k], 1e-36f));
    U[k] = G[k] * inv; U[3 + k] = G[3 + k] * inv; U[6 + k] = G[6 + k] * inv;
    S[k] = n2[k] * inv;
  }
  S[2] = (n2[2] > 0.0f) ? n2[2] * __builtin_amdgcn_rsqf(n2[2]) : 0.0f;
  {
    const float dt = U[0] * U[1] + U[3] * U[4] + U[6] * U[7];
    const float a0 = U[1] - dt * U[0], a1 = U[4] - dt * U[3], a2 = U[7] - dt * U[6];
    const float inv = __builtin_amdgcn_rsqf(fmaxf(a0 * a0 + a1 * a1 + a2 * a2, 1e-36f));
    U[1] = a0 * inv; U[4] = a1 * inv; U[7] = a2 * inv;
  }
  const float c0 = U[3] * U[7] - U[6] * U[4];
  const float c1 = U[6] * U[1] - U[0] * U[7];
  const float c2 = U[0] * U[4] - U[3] * U[1];
  const float sg = (c0 * G[2] + c1 * G[5] + c2 * G[8] < 0.0f) ? -1.0f : 1.0f;
  U[2] = sg * c0; U[5] = sg * c1; U[8] = sg * c2;
}

__device__ __forceinline__ double svd_rsqrt(double x) { return fast_rsqrt(x); }
__device__ __forceinline__ float svd_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }

// One-sided (Hestenes) Jacobi SVD of a 3x3 matrix: F = U diag(S) V^T, S descending, S[2] >= 0 given the
// orientation chosen for u3.  U, V row-major with singular vectors in columns.  Straight-line code on
// values in registers; `T` is float (forward rank-2 step, backward bookkeeping) or double (pose kernels).
template <typename T>
__device__ inline void svd3(const T* F, T* U, T* S, T* V) {
  T G[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    G[i] = F[i];
    V[i] = (i % 4 == 0) ? T(1) : T(0);
  }
  const T tol = (sizeof(T) == 4) ? T(1e-7) : T(1e-15);
  for (int sweep = 0; sweep < 12; ++sweep) {
    T worst = T(0);
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = (pq == 2) ? 1 : 0;
      const int q = (pq == 0) ? 1 : 2;
      T al = G[p] * G[p] + G[3 + p] * G[3 + p] + G[6 + p] * G[6 + p];
      T be = G[q] * G[q] + G[3 + q] * G[3 + q] + G[6 + q] * G[6 + q];
      T ga = G[p] * G[q] + G[3 + p] * G[3 + q] + G[6 + p] * G[6 + q];
      const bool rot = ga * ga > tol * tol * al * be;  // division-free skip test
      worst = rot ? T(1) : worst;
      if (rot) {
        // r = 1/h, h = sqrt(d^2+b^2); x = (1 + |d|/h)/2 = cos^2; c = sqrt(x), s = sgn(d) b / (2 h c)
        const T d = be - al, b = T(2) * ga;
        const T rh = svd_rsqrt(d * d + b * b);
        const T x = T(0.5) + T(0.5) * fabs(d) * rh;
        const T y = svd_rsqrt(x);
        const T c = x * y;
        const T s = copysign(T(0.5), d) * b * rh * y;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          T gp = G[3 * r + p], gq = G[3 * r + q];
          G[3 * r + p] = c * gp - s * gq;
          G[3 * r + q] = s * gp + c * gq;
          T vp = V[3 * r + p], vq = V[3 * r + q];
          V[3 * r + p] = c * vp - s * vq;
          V[3 * r + q] = s * vp + c * vq;
        }
      }
    }
    if (worst == T(0)) break;
  }
  T n[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) n[k] = sqrt(G[k] * G[k] + G[3 + k] * G[3 + k] + G[6 + k] * G[6 + k]);
  // sort columns descending (3-element network)
#define DFEPE_SWAPCOL(a, b)                                   \
  if (n[a] < n[b]) {                                          \
    T tn = n[a]; n[a] = n[b]; n[b] = tn;                      \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {           \
      T tg = G[3 * r + a]; G[3 * r + a] = G[3 * r + b]; G[3 * r + b] = tg; \
      T tv = V[3 * r + a]; V[3 * r + a] = V[3 * r + b]; V[3 * r + b] = tv; \
    }                                                         \
  }
  DFEPE_SWAPCOL(0, 1)
  DFEPE_SWAPCOL(1, 2)
  DFEPE_SWAPCOL(0, 1)
#undef DFEPE_SWAPCOL
  const T tiny = T(1e-30);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    T inv = T(1) / fmax(n[k], tiny);
    U[k] = G[k] * inv;
    U[3 + k] = G[3 + k] * inv;
    U[6 + k] = G[6 + k] * inv;
  }
  // u2 re-orthogonalised against u1 (matters only when s2 is tiny), u3 = u1 x u2 oriented along g3
  {
    T d = U[0] * U[1] + U[3] * U[4] + U[6] * U[7];
    T a0 = U[1] - d * U[0], a1 = U[4] - d * U[3], a2 = U[7] - d * U[6];
    T inv = T(1) / fmax(sqrt(a0 * a0 + a1 * a1 + a2 * a2), tiny);
    U[1] = a0 * inv; U[4] = a1 * inv; U[7] = a2 * inv;
  }
  T c0 = U[3] * U[7] - U[6] * U[4];
  T c1 = U[6] * U[1] - U[0] * U[7];
  T c2 = U[0] * U[4] - U[3] * U[1];
  T sg = (c0 * G[2] + c1 * G[5] + c2 * G[8] < T(0)) ? T(-1) : T(1);
  U[2] = sg * c0; U[5] = sg * c1; U[8] = sg * c2;
  S[0] = n[0]; S[1] = n[1]; S[2] = n[2];
}
