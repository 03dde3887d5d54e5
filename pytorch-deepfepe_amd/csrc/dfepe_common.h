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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
  return v;
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
  const double inv = ok ? 1.0 / fmax(sqrt(n2), 1e-12) : 0.0;
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
      T lim = sqrt(al * be);
      T rel = (lim > T(0)) ? fabs(ga) / lim : T(0);
      worst = fmax(worst, rel);
      if (rel > tol) {
        T zeta = (be - al) / (T(2) * ga);
        T t = copysign(T(1), zeta) / (fabs(zeta) + sqrt(T(1) + zeta * zeta));
        T c = T(1) / sqrt(T(1) + t * t);
        T s = c * t;
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
    if (worst <= tol) break;
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
