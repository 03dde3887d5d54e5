// Shared device helpers for the dfepe HIP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dfepe.h"
#include "rowgroup.h"   // hw_rsq / hw_rcp and the 16-lane row primitives
#include "dfepe_math.h"

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


// Compiler-level fence between wave-synchronous LDS phases.  A single wavefront's LDS operations
// execute in issue order, so no s_barrier is needed; this only stops the compiler from moving
// memory operations across the phase boundary.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


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


