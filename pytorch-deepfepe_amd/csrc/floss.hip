// floss_fwd / floss_bwd — epipolar "F-loss" on the virtual correspondences for all layers, and E-from-F.
//
// Restates the per-layer body of get_all_loss_DeepF (deepFEPE/train_good_utils.py:325-358):
//   pts*_eval = T* virt*^T                                            (:325-326)
//   losses_l  = compute_epi_residual(pts1_eval, pts2_eval, F_l, clamp) (:340-342, utils_F.py:400-413)
//   E_l       = K^T T2^T F_l T1 K                                      (:356-358)
// One wavefront per image pair: lanes stride over the M virtual points (each point is transformed once and
// reused for every layer), per-layer sums are wave-reduced; lanes 0..L-1 then form the L essential matrices.
// The means over M and over the batch are left to the caller (a [L,B] tensor; under data parallelism the
// batch mean is the all-reduced sum), exactly the quantities get_all_loss_DeepF derives from `losses`.
#include "dfepe_common.h"

namespace {

constexpr int kMaxLayers = 16;

// Per-point epipolar terms in fp32 (the reference computes the F-loss in fp32, train_good_utils.py:340-342) on the
// hardware sqrt/rcp; per-lane partial sums are combined across the wave and the layers in fp64.
struct Epi {
  float d, dd, n1, n2, i1, i2;
  float l1[3], l2[3];
};

__device__ __forceinline__ Epi epi_terms(const float* x1, const float* x2, const float* o) {
  Epi e;
#pragma unroll
  for (int c = 0; c < 3; ++c) e.l1[c] = fmaf(x2[0], o[c], fmaf(x2[1], o[3 + c], x2[2] * o[6 + c]));
#pragma unroll
  for (int r = 0; r < 3; ++r) e.l2[r] = fmaf(o[3 * r], x1[0], fmaf(o[3 * r + 1], x1[1], o[3 * r + 2] * x1[2]));
  e.dd = fmaf(x1[0], e.l1[0], fmaf(x1[1], e.l1[1], x1[2] * e.l1[2]));
  e.n1 = __builtin_amdgcn_sqrtf(fmaf(e.l1[0], e.l1[0], e.l1[1] * e.l1[1]));
  e.n2 = __builtin_amdgcn_sqrtf(fmaf(e.l2[0], e.l2[0], e.l2[1] * e.l2[1]));
  e.i1 = __builtin_amdgcn_rcpf(e.n1 + 1e-6f);
  e.i2 = __builtin_amdgcn_rcpf(e.n2 + 1e-6f);
  e.d = fabsf(e.dd) * (e.i1 + e.i2);
  return e;
}

// 3x3 wave-uniform matrix -> scalar registers
__device__ __forceinline__ void load9(const float* p, double* m) {
#pragma unroll
  for (int c = 0; c < 9; ++c) m[c] = to_sgpr((double)p[c]);
}

__device__ __forceinline__ void eval_point(const float* v, const double* T, float* x) {
  // T * pixel in fp64 (pixels ~1e3 times 2/W minus 1 cancels ~3 digits), rounded once to fp32 like the reference's pts_eval
  const double a = v[0], b = v[1], c = v[2];
#pragma unroll
  for (int r = 0; r < 3; ++r) x[r] = (float)(T[3 * r] * a + T[3 * r + 1] * b + T[3 * r + 2] * c);
}

// Sums of nine per-lane values over the wavefront, total[c] delivered in every lane whose (lane & 15) == c: four DPP
// steps per value give the 16-lane row sums, each lane then picks the value of its own index and two cross-row exchanges
// finish all nine at once (50 instructions instead of nine separate wave sums at ~11 each).
__device__ __forceinline__ float wave_sum9_scattered(const float* a, int lane) {
  float r[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    float v = a[c];
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    r[c] = v;
  }
  const int k = lane & 15;
  float mine = r[0];
#pragma unroll
  for (int c = 1; c < 9; ++c) mine = (k == c) ? r[c] : mine;
  mine += __shfl_xor(mine, 16, WAVE);
  mine += __shfl_xor(mine, 32, WAVE);
  return mine;
}

template <bool BWD, bool CACHED>
__global__ void __launch_bounds__(256, 4)
floss_kernel(const float* __restrict__ F_layers, int L, int B, const float* __restrict__ T1, const float* __restrict__ T2,
             int t_stride, const float* __restrict__ K, const float* __restrict__ virt1, const float* __restrict__ virt2,
             int M, float clamp_at, float* __restrict__ loss_sum, float* __restrict__ E_layers,
             const float* __restrict__ g_loss_sum, float g_loss_coef, const float* __restrict__ g_scale,
             const float* __restrict__ g_E, float* __restrict__ g_F_layers) {
  __shared__ float gsum[4][kMaxLayers][9];  // BWD: per-layer sums of the point gradients, one slab per wavefront of the block
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform by construction
  const size_t pair = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
  if (pair >= (size_t)B) return;
  double t1[9], t2[9];
  load9(T1 + pair * t_stride, t1);
  load9(T2 + pair * t_stride, t2);
  const float* v1 = virt1 + pair * M * 3;
  const float* v2 = virt2 + pair * M * 3;
  // A = T2 K, C = T1 K (E = A^T F C), formed once per pair and parked in scalar registers
  double Am[9], Cm[9];
  if (K != nullptr && (BWD ? (g_E != nullptr) : (E_layers != nullptr))) {
    double k[9], ta[9], tc[9];
    load9(K + pair * 9, k);
    mat3_mul(t2, k, ta);
    mat3_mul(t1, k, tc);
#pragma unroll
    for (int c = 0; c < 9; ++c) { Am[c] = to_sgpr(ta[c]); Cm[c] = to_sgpr(tc[c]); }
  }

  // M <= 128 (the reference uses a 10x10 grid): the (at most two) transformed points of a lane are formed once and kept
  // in registers for all layers
  float cx1[2][3], cx2[2][3];
  if (CACHED) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = lane + WAVE * k;
      if (i < M) {
        eval_point(v1 + 3 * i, t1, cx1[k]);
        eval_point(v2 + 3 * i, t2, cx2[k]);
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) { cx1[k][c] = 0.0f; cx2[k][c] = 0.0f; }
      }
    }
  }

  for (int l = 0; l < L; ++l) {
    float o[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c] = to_sgpr(F_layers[((size_t)l * B + pair) * 9 + c]);
    if (!BWD) {
      float accf = 0.0f;
      if (CACHED) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const Epi e = epi_terms(cx1[k], cx2[k], o);
          accf += (lane + WAVE * k < M) ? fminf(e.d, clamp_at) : 0.0f;
        }
      } else {
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (int i = lane; i < M; i += WAVE) {
          float x1[3], x2[3];
          eval_point(v1 + 3 * i, t1, x1);
          eval_point(v2 + 3 * i, t2, x2);
          const Epi e = epi_terms(x1, x2, o);
          accf += fminf(e.d, clamp_at);
        }
      }
      const float acc = wave_sum(accf);  // <= 128 terms of at most clamp_at each: fp32 like the reference's own sum
      if (lane == 0) loss_sum[(size_t)l * B + pair] = acc;
    } else {
      float gof[9];  // per-lane partial sums in fp32 (like the reference's backward); scaled and combined in fp64 below
#pragma unroll
      for (int c = 0; c < 9; ++c) gof[c] = 0.0f;
      const bool has_gl = (g_loss_sum != nullptr) || (g_loss_coef != 0.0f);
      if (has_gl) {
        auto point_grad = [&](const float* x1, const float* x2, bool live) {
          const Epi e = epi_terms(x1, x2, o);
          if (live && e.d <= clamp_at) {
            const float S = e.i1 + e.i2, ad = fabsf(e.dd);
            const float sg = (e.dd > 0.0f) ? 1.0f : ((e.dd < 0.0f) ? -1.0f : 0.0f);
            const float k1 = (e.n1 > 0.0f) ? ad * e.i1 * e.i1 * __builtin_amdgcn_rcpf(e.n1) : 0.0f;
            const float k2 = (e.n2 > 0.0f) ? ad * e.i2 * e.i2 * __builtin_amdgcn_rcpf(e.n2) : 0.0f;
            // d d / d F[r][c] = sg S x2[r] x1[c] - k1 l1[c] x2[r] [c<2] - k2 l2[r] x1[c] [r<2]
            //                 = x2[r] a[c] - b[r] x1[c],  a[c] = sg S x1[c] - k1 l1[c] [c<2],  b[r] = k2 l2[r] [r<2]
            const float sS = sg * S;
            const float a0 = fmaf(sS, x1[0], -k1 * e.l1[0]), a1 = fmaf(sS, x1[1], -k1 * e.l1[1]), a2 = sS * x1[2];
            const float b0 = k2 * e.l2[0], b1 = k2 * e.l2[1];
            const float av[3] = {a0, a1, a2};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              gof[c] += fmaf(x2[0], av[c], -b0 * x1[c]);
              gof[3 + c] += fmaf(x2[1], av[c], -b1 * x1[c]);
              gof[6 + c] = fmaf(x2[2], av[c], gof[6 + c]);
            }
          }
        };
        if (CACHED) {
#pragma unroll
          for (int k = 0; k < 2; ++k) point_grad(cx1[k], cx2[k], lane + WAVE * k < M);
        } else {
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
          for (int i = lane; i < M; i += WAVE) {
            float x1[3], x2[3];
            eval_point(v1 + 3 * i, t1, x1);
            eval_point(v2 + 3 * i, t2, x2);
            point_grad(x1, x2, true);
          }
        }
        const float tot = wave_sum9_scattered(gof, lane);
        if (lane < 9) gsum[wave][l][lane] = tot;
      } else if (lane < 9) {
        gsum[wave][l][lane] = 0.0f;
      }
    }
  }
  if (BWD) {
    // combine per layer, one layer per lane: g_F_l = gl_l * sums_l + (T2 K) g_E_l (T1 K)^T   (E = (T2 K)^T F (T1 K))
    wave_sync();
    if (lane < L) {
      const size_t lb = (size_t)lane * B + pair;
      const double gl = (g_loss_sum != nullptr) ? (double)g_loss_sum[lb]
                                                : (double)g_loss_coef * ((g_scale != nullptr) ? (double)g_scale[0] : 1.0);
      double go[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) go[c] = gl * (double)gsum[wave][lane][c];
      if (g_E != nullptr) {
        double ge[9], tmp[9], add[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) ge[c] = (double)g_E[lb * 9 + c];
        mat3_mul(Am, ge, tmp);
        mat3_mul_nt(tmp, Cm, add);
#pragma unroll
        for (int c = 0; c < 9; ++c) go[c] += add[c];
      }
#pragma unroll
      for (int c = 0; c < 9; ++c) g_F_layers[lb * 9 + c] = (float)go[c];
    }
  }
  if (!BWD && E_layers != nullptr && lane < L) {
    double o[9], tmp[9], e[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c] = (double)F_layers[((size_t)lane * B + pair) * 9 + c];  // per-lane layer: not uniform
    mat3_mul_tn(Am, o, tmp);
    mat3_mul(tmp, Cm, e);
#pragma unroll
    for (int c = 0; c < 9; ++c) E_layers[((size_t)lane * B + pair) * 9 + c] = (float)e[c];
  }
}

int check_common(const float* F_layers, int L, int B, const float* T1, const float* T2, int t_stride, const float* K,
                 const float* virt1, const float* virt2, int M) {
  if (L <= 0 || L > kMaxLayers || B < 0 || M <= 0) return DFEPE_ERR_INVALID_ARG;
  if (t_stride != 0 && t_stride != 9) return DFEPE_ERR_INVALID_ARG;
  if (B > 0 && (!F_layers || !T1 || !T2 || !K || !virt1 || !virt2)) return DFEPE_ERR_INVALID_ARG;
  return DFEPE_OK;
}

}  // namespace

extern "C" int dfepe_floss_fwd(const float* F_layers, int L, int B, const float* T1, const float* T2, int t_stride,
                               const float* K, const float* virt1, const float* virt2, int M, float clamp_at,
                               float* loss_sum, float* E_layers, void* stream) {
  const int rc = check_common(F_layers, L, B, T1, T2, t_stride, K, virt1, virt2, M);
  if (rc != DFEPE_OK) return rc;
  if (B == 0) return DFEPE_OK;
  if (!loss_sum) return DFEPE_ERR_INVALID_ARG;
  const dim3 grid((B + 3) / 4), block(256);
  if (M <= 2 * WAVE)
    hipLaunchKernelGGL((floss_kernel<false, true>), grid, block, 0, static_cast<hipStream_t>(stream), F_layers, L, B, T1, T2,
                       t_stride, K, virt1, virt2, M, clamp_at, loss_sum, E_layers, nullptr, 0.0f, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL((floss_kernel<false, false>), grid, block, 0, static_cast<hipStream_t>(stream), F_layers, L, B, T1, T2,
                       t_stride, K, virt1, virt2, M, clamp_at, loss_sum, E_layers, nullptr, 0.0f, nullptr, nullptr, nullptr);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_floss_bwd(const float* F_layers, int L, int B, const float* T1, const float* T2, int t_stride,
                               const float* K, const float* virt1, const float* virt2, int M, float clamp_at,
                               const float* g_loss_sum, float g_loss_coef, const float* g_scale, const float* g_E,
                               float* g_F_layers, void* stream) {
  const int rc = check_common(F_layers, L, B, T1, T2, t_stride, K, virt1, virt2, M);
  if (rc != DFEPE_OK) return rc;
  if (B == 0) return DFEPE_OK;
  if (!g_F_layers) return DFEPE_ERR_INVALID_ARG;
  const dim3 grid((B + 3) / 4), block(256);
  if (M <= 2 * WAVE)
    hipLaunchKernelGGL((floss_kernel<true, true>), grid, block, 0, static_cast<hipStream_t>(stream), F_layers, L, B, T1, T2,
                       t_stride, K, virt1, virt2, M, clamp_at, nullptr, nullptr, g_loss_sum, g_loss_coef, g_scale, g_E, g_F_layers);
  else
    hipLaunchKernelGGL((floss_kernel<true, false>), grid, block, 0, static_cast<hipStream_t>(stream), F_layers, L, B, T1, T2,
                       t_stride, K, virt1, virt2, M, clamp_at, nullptr, nullptr, g_loss_sum, g_loss_coef, g_scale, g_E, g_F_layers);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
