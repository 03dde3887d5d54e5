// Stand-alone geometry entry points of the dsac_tools API: epipolar residual / metrics, small pose helpers,
// cheirality-checked pose selection.  These sit on either side of the solver.
#include "dfepe_common.h"
#include <cstdlib>

#include "cheirality_body.h"

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
    svd3_closed(E, U, S, V);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Ed[3 * r + c] = U[3 * r] * V[3 * c] + U[3 * r + 1] * V[3 * c + 1];
#pragma unroll
    for (int k = 0; k < 9; ++k) out[i * 9 + k] = (float)Ed[k];
  } else if (kind == 6) {
    // camera-motion rotation of a scene motion: inv(delta)[:3,:3] for a general 4x4 (cofactors of the full matrix, so the
    // last row need not be (0,0,0,1)); np.linalg.inv(delta_Rtij)[:3,:3] of get_Rt_loss, train_good_utils.py:134,170
    double m[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k] = (double)in0[i * 16 + k];
    // 2x2 minors of the lower two rows / upper two rows
    const double s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    const double s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const double c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    const double c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    const double id = 1.0 / det;
    double r[9];
    r[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * id;
    r[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * id;
    r[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * id;
    r[3] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * id;
    r[4] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * id;
    r[5] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * id;
    r[6] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * id;
    r[7] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * id;
    r[8] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * id;
#pragma unroll
    for (int k = 0; k < 9; ++k) out[i * 9 + k] = (float)r[k];
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
// cheirality (utils_F._E_to_M_train, utils_F.py:679-763): body in cheirality_body.h (shared with the fused fit + pose kernel)
// ------------------------------------------------------------------------------------------------------
// <= 128 registers = FOUR wavefronts per SIMD (round 5; rounds 2-4: 166 registers, three): the projection matrices live in scalar
// registers and the fp64 route handles one rotation candidate at a time, so the packed-fp32 body sets the register count.  The bound
// also admits two 8-wavefront workgroups per CU: a pair's 1000 correspondences are then two groups of 64 per wavefront, not four.
template <bool FP64_ONLY, bool WS>
__global__ void __launch_bounds__(512, 2)
cheirality_kernel(const float* __restrict__ E, const float* __restrict__ pre, const float* __restrict__ K,
                  const float* __restrict__ matches, int B, int N, float depth_thres, const double* __restrict__ ws,
                  float* __restrict__ Rt_cam, int* __restrict__ winner, int* __restrict__ counts) {
  __shared__ CheirLds cl;
  const size_t pair = blockIdx.x;
  float Ef[9];
  if constexpr (!WS) {
#pragma unroll
    for (int k = 0; k < 9; ++k) Ef[k] = E[pair * 9 + k];
  }
  cheirality_pair<FP64_ONLY, WS>(Ef, pre, K, matches, pair, N, depth_thres, Rt_cam, winner, counts, cl, ws);
  (void)B;
}

// the per-pair constants of the loop (cheirality_body.h: kCheirPrep doubles per pair), one LANE per pair
__global__ void __launch_bounds__(64)
cheirality_prepare_kernel(const float* __restrict__ E, const float* __restrict__ pre, const float* __restrict__ K, int B,
                          double* __restrict__ ws) {
  const int pair = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (pair >= B) return;
  float Ef[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Ef[k] = E[(size_t)pair * 9 + k];
  double out[kCheirPrep];
  cheir_prepare(Ef, pre != nullptr ? pre + (size_t)pair * 9 : nullptr, K + (size_t)pair * 9, out);
#pragma unroll
  for (int k = 0; k < kCheirPrep; ++k) ws[(size_t)pair * kCheirPrep + k] = out[k];
}

// ------------------------------------------------------------------------------------------------------
// DeepFNet.get_input (DeepFNet.py:362-391) with NormalizeAndExpand_HW (:93-120): one thread per correspondence
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
deepf_input_kernel(const float4* __restrict__ matches, const float* __restrict__ quality, int B, int N, int Q, float sx, float sy,
                   float* __restrict__ weight_in, size_t channel_stride, size_t batch_stride, int n_copies, size_t copy_stride,
                   float* __restrict__ pts1, float* __restrict__ pts2) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * N) return;
  const size_t b = idx / (size_t)N;
  const int i = (int)(idx - b * (size_t)N);
  const float4 m = matches[idx];
  // the same arithmetic as the fit kernels' fused prologue (global_point<RAW>): T_HW (x, y, 1)
  const float x1 = fmaf(m.x, sx, -1.0f), y1 = fmaf(m.y, sy, -1.0f), x2 = fmaf(m.z, sx, -1.0f), y2 = fmaf(m.w, sy, -1.0f);
  if (pts1 != nullptr) { pts1[idx * 3] = x1; pts1[idx * 3 + 1] = y1; pts1[idx * 3 + 2] = 1.0f; }
  if (pts2 != nullptr) { pts2[idx * 3] = x2; pts2[idx * 3 + 1] = y2; pts2[idx * 3 + 2] = 1.0f; }
  if (weight_in == nullptr) return;
  const float c0 = (x1 + 1.0f) * 0.5f, c1 = (y1 + 1.0f) * 0.5f, c2 = (x2 + 1.0f) * 0.5f, c3 = (y2 + 1.0f) * 0.5f;  // (:375-378)
  for (int k = 0; k < n_copies; ++k) {
    float* w = weight_in + (size_t)k * copy_stride + b * batch_stride + i;
    w[0] = c0; w[channel_stride] = c1; w[2 * channel_stride] = c2; w[3 * channel_stride] = c3;
    for (int q = 0; q < Q; ++q) w[(size_t)(4 + q) * channel_stride] = quality[idx * Q + q];
  }
}

// sum_n a[l,b,n] b[l,b,n] of every (layer, pair): one 16-lane row per item, the layers of each operand a fixed distance apart
__global__ void __launch_bounds__(256)
row_dot_kernel(const float* __restrict__ a, size_t a_layer_stride, const float* __restrict__ b, size_t b_layer_stride, int n_layers,
               int B, int N, float* __restrict__ out) {
  const size_t item = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (item >= (size_t)n_layers * B) return;
  const int l = (int)(item / (size_t)B);
  const size_t p = item - (size_t)l * B;
  const float* x = a + (size_t)l * a_layer_stride + p * N;
  const float* y = b + (size_t)l * b_layer_stride + p * N;
  float acc = 0.0f;
  for (int i = (int)(threadIdx.x & 15u); i < N; i += 16) acc = fmaf(x[i], y[i], acc);
  acc = rg_sum(acc);
  if ((threadIdx.x & 15u) == 0) out[item] = acc;
}

}  // namespace

extern "C" int dfepe_row_dot(const float* a, size_t a_layer_stride, const float* b, size_t b_layer_stride, int n_layers, int B, int N,
                             float* out, void* stream) {
  if (n_layers < 0 || B < 0 || N <= 0) return DFEPE_ERR_INVALID_ARG;
  if (n_layers == 0 || B == 0) return DFEPE_OK;
  if (!a || !b || !out) return DFEPE_ERR_INVALID_ARG;
  const size_t n = (size_t)n_layers * B;
  hipLaunchKernelGGL(row_dot_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, static_cast<hipStream_t>(stream), a, a_layer_stride, b,
                     b_layer_stride, n_layers, B, N, out);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_deepf_input(const float* matches, const float* quality, int B, int N, int Q, float image_w, float image_h,
                                 float* weight_in, size_t channel_stride, size_t batch_stride, int n_copies, size_t copy_stride,
                                 float* pts1, float* pts2, void* stream) {
  if (B < 0 || N <= 0 || Q < 0 || !(image_w > 0.f) || !(image_h > 0.f)) return DFEPE_ERR_INVALID_ARG;
  if (B == 0) return DFEPE_OK;
  if (!matches || (Q > 0 && !quality) || (reinterpret_cast<uintptr_t>(matches) & 15u)) return DFEPE_ERR_INVALID_ARG;
  if (weight_in && (n_copies < 1 || channel_stride < 1 || batch_stride < 1)) return DFEPE_ERR_INVALID_ARG;
  const size_t n = (size_t)B * N;
  hipLaunchKernelGGL(deepf_input_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(matches), quality, B, N, Q, 2.0f / image_w, 2.0f / image_h, weight_in,
                     channel_stride, batch_stride, n_copies, copy_stride, pts1, pts2);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

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
  if (kind < 0 || kind > 6 || n < 0) return DFEPE_ERR_INVALID_ARG;
  if (n == 0) return DFEPE_OK;
  if (!in0 || !out || ((kind == 1 || kind == 2 || kind == 5) && !in1)) return DFEPE_ERR_INVALID_ARG;
  hipLaunchKernelGGL(geo_misc_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), kind, in0, in1, n, out);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" size_t dfepe_cheirality_workspace_bytes(int B) { return (B > 0) ? (size_t)B * kCheirPrep * sizeof(double) : 0; }

extern "C" int dfepe_cheirality_ex(const float* E, const float* pre, const float* K, const float* matches, int B, int N,
                                   float depth_thres, unsigned flags, void* workspace, float* Rt_cam, int* winner, int* counts,
                                   void* stream) {
  if (B < 0 || N <= 0 || (flags & ~DFEPE_CHEIR_FP64_ONLY)) return DFEPE_ERR_INVALID_ARG;
  if (B == 0) return DFEPE_OK;
  if (!E || !K || !matches || !Rt_cam) return DFEPE_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(matches) & 15u) || (reinterpret_cast<uintptr_t>(workspace) & 7u)) return DFEPE_ERR_INVALID_ARG;
  // wavefronts per pair: throughput wants one (B >= 2048: every SIMD already holds >= 2 pairs), latency wants several -- four:
  // eight were measured slower at 512 x 1000 (24.8 against 20.7 us: every wavefront ends with a pass of its own few ambiguous
  // correspondences through the fp64 route, which costs what a full group of 64 does, and a pair has eight such passes then)
  const int groups = (N + 63) / 64;
  // (A/B switch, read once: DFEPE_CHEIR_WAVES = wavefronts per pair below 2048 pairs)
  static const int forced_waves = [] { const char* e = getenv("DFEPE_CHEIR_WAVES"); return e ? atoi(e) : 0; }();
  const int wmax = (forced_waves >= 1 && forced_waves <= 8) ? forced_waves : 4;
  const int threads = (B >= 2048) ? 64 : 64 * (groups < wmax ? groups : wmax);
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* ws = static_cast<double*>(workspace);
  const bool f64 = (flags & DFEPE_CHEIR_FP64_ONLY) != 0;
  if (ws != nullptr) {
    hipLaunchKernelGGL(cheirality_prepare_kernel, dim3((B + 63) / 64), dim3(64), 0, st, E, pre, K, B, ws);
    if (f64) hipLaunchKernelGGL((cheirality_kernel<true, true>), dim3(B), dim3(threads), 0, st, E, pre, K, matches, B, N, depth_thres, ws, Rt_cam, winner, counts);
    else hipLaunchKernelGGL((cheirality_kernel<false, true>), dim3(B), dim3(threads), 0, st, E, pre, K, matches, B, N, depth_thres, ws, Rt_cam, winner, counts);
  } else {
    if (f64) hipLaunchKernelGGL((cheirality_kernel<true, false>), dim3(B), dim3(threads), 0, st, E, pre, K, matches, B, N, depth_thres, ws, Rt_cam, winner, counts);
    else hipLaunchKernelGGL((cheirality_kernel<false, false>), dim3(B), dim3(threads), 0, st, E, pre, K, matches, B, N, depth_thres, ws, Rt_cam, winner, counts);
  }
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_cheirality(const float* E, const float* pre, const float* K, const float* matches, int B, int N,
                                float depth_thres, float* Rt_cam, int* winner, int* counts, void* stream) {
  return dfepe_cheirality_ex(E, pre, K, matches, B, N, depth_thres, 0u, nullptr, Rt_cam, winner, counts, stream);
}
