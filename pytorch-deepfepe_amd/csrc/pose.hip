// pose_fwd / pose_bwd — four-fold pose decomposition of E^T, quaternion / translation L2 errors against the
// ground truth, candidate selection and angular metrics; analytic adjoint w.r.t. E.
//
// Restates the per-layer, per-sample loop of get_Rt_loss (deepFEPE/train_good_utils.py:96-239):
//   _get_M2s      dsac_tools/utils_F.py:478-498   (U W V^T, W negated when det < 0, t = u3/|u3|)
//   _R_to_q       dsac_tools/utils_geo.py:58-86   (trace method on m = R^T, 4 branches, q0 >= 0)
//   _l2_error     dsac_tools/utils_geo.py:165-167
//   selection by strict '<'                       train_good_utils.py:160-168 (R and t picked independently)
//   rot12_to_angle_error / vector_angle           dsac_tools/utils_geo.py:150-155, 175-179
// One lane per (layer, pair): everything is 3x3 work in fp64 registers (3x3 one-sided Jacobi SVD).
// The adjoint of the SVD uses the combined (1,2)-block form Z12/(s1+s2): for R = U W V^T the generic
// 1/(s1^2-s2^2) terms cancel analytically, so true essential matrices (s1 == s2) stay finite.
// cv2.Rodrigues is replaced by atan2(|axis|, trace-1) (same angle; OpenCV arithmetic is unpinned).
#include "dfepe_common.h"
#include "pose_math.h"

namespace {

__global__ void __launch_bounds__(256)
pose_fwd_kernel(const float* __restrict__ E_layers, int L, int B, const float* __restrict__ q_gt,
                const float* __restrict__ t_gt, const float* __restrict__ R_gt, float* __restrict__ q_l2,
                float* __restrict__ t_l2, float* __restrict__ R_deg, float* __restrict__ t_deg, int* __restrict__ sel) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)L * B) return;
  const size_t b = idx % B;
  Pose P;
  pose_forward(E_layers + idx * 9, q_gt + b * 4, t_gt + b * 3, P);
  q_l2[idx] = (float)P.qe[P.qi];
  t_l2[idx] = (float)P.te[P.ti];
  if (sel != nullptr) sel[idx] = P.qi | (P.ti << 1);
  if (R_deg != nullptr && R_gt != nullptr) R_deg[idx] = (float)pose_R_deg(P, R_gt + b * 9);
  if (t_deg != nullptr) t_deg[idx] = (float)pose_t_deg(P);
}

__global__ void __launch_bounds__(256)
pose_bwd_kernel(const float* __restrict__ E_layers, int L, int B, const float* __restrict__ q_gt,
                const float* __restrict__ t_gt, const float* __restrict__ g_q_l2, const float* __restrict__ g_t_l2,
                float coef_q, float clamp_q, float coef_t, float clamp_t, const float* __restrict__ g_scale,
                float* __restrict__ g_E) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)L * B) return;
  const size_t b = idx % B;
  Pose P;
  pose_forward(E_layers + idx * 9, q_gt + b * 4, t_gt + b * 3, P);
  const double gs = (g_scale != nullptr) ? (double)g_scale[0] : 1.0;
  // torch.clamp passes the gradient on [min, max] inclusive
  const double gql = (g_q_l2 != nullptr) ? (double)g_q_l2[idx] : ((P.qe[P.qi] <= (double)clamp_q) ? (double)coef_q * gs : 0.0);
  const double gtl = (g_t_l2 != nullptr) ? (double)g_t_l2[idx] : ((P.te[P.ti] <= (double)clamp_t) ? (double)coef_t * gs : 0.0);
  double gE[9];
  pose_backward(P, q_gt + b * 4, gql, gtl, gE);
#pragma unroll
  for (int k = 0; k < 9; ++k) g_E[idx * 9 + k] = (float)gE[k];
}

// loss head: one block of 1024 threads, one pass over the three [L,B] arrays with per-layer register accumulators,
// DPP wave sums, then a 16-wave combine through LDS (two block barriers in total)
constexpr int kHeadMaxL = 16;
__global__ void __launch_bounds__(1024)
loss_head_kernel(const float* __restrict__ loss_sum, const float* __restrict__ q_l2, const float* __restrict__ t_l2, int L,
                 int B, int M, float clamp_q, float clamp_t, float balance_q, float balance_t, double* __restrict__ packed,
                 float* __restrict__ scalars) {
  __shared__ double red[16][kHeadMaxL + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float acc[kHeadMaxL];
#pragma unroll
  for (int l = 0; l < kHeadMaxL; ++l) acc[l] = 0.0f;
  float q = 0.0f, t = 0.0f;
  const bool pose = (q_l2 != nullptr);
  if ((B & 3) == 0) {
    // four pairs per thread and float4 loads: at B = 4096 every load of the kernel is in flight at once, so the
    // single block pays one memory latency instead of one per 1024 pairs
    const int B4 = B >> 2;
    for (int b4 = threadIdx.x; b4 < B4; b4 += blockDim.x) {
#pragma unroll
      for (int l = 0; l < kHeadMaxL; ++l) {
        if (l < L) {
          const float4 v = reinterpret_cast<const float4*>(loss_sum + (size_t)l * B)[b4];
          acc[l] += (v.x + v.y) + (v.z + v.w);
          if (pose) {
            const float4 qv = reinterpret_cast<const float4*>(q_l2 + (size_t)l * B)[b4];
            const float4 tv = reinterpret_cast<const float4*>(t_l2 + (size_t)l * B)[b4];
            q += (fminf(fmaxf(qv.x, 0.0f), clamp_q) + fminf(fmaxf(qv.y, 0.0f), clamp_q)) +
                 (fminf(fmaxf(qv.z, 0.0f), clamp_q) + fminf(fmaxf(qv.w, 0.0f), clamp_q));
            t += (fminf(fmaxf(tv.x, 0.0f), clamp_t) + fminf(fmaxf(tv.y, 0.0f), clamp_t)) +
                 (fminf(fmaxf(tv.z, 0.0f), clamp_t) + fminf(fmaxf(tv.w, 0.0f), clamp_t));
          }
        }
      }
    }
  } else {
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
#pragma unroll
      for (int l = 0; l < kHeadMaxL; ++l) {
        if (l < L) {
          acc[l] += loss_sum[(size_t)l * B + b];
          if (pose) {
            q += fminf(fmaxf(q_l2[(size_t)l * B + b], 0.0f), clamp_q);
            t += fminf(fmaxf(t_l2[(size_t)l * B + b], 0.0f), clamp_t);
          }
        }
      }
    }
  }
#pragma unroll
  for (int l = 0; l < kHeadMaxL; ++l) {
    if (l < L) {
      const double s = wave_sum((double)acc[l]);
      if (lane == 0) red[wave][l] = s;
    }
  }
  {
    const double sq = wave_sum((double)q), stt = wave_sum((double)t);
    if (lane == 0) { red[wave][kHeadMaxL] = sq; red[wave][kHeadMaxL + 1] = stt; }
  }
  __syncthreads();
  if (threadIdx.x < kHeadMaxL + 2) {
    double s = 0.0;
    for (int k = 0; k < nw; ++k) s += red[k][threadIdx.x];
    red[0][threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double totF = 0.0;
    for (int l = 0; l < L; ++l) { packed[l] = red[0][l]; totF += red[0][l]; }
    const double tq = red[0][kHeadMaxL], tt = red[0][kHeadMaxL + 1];
    packed[L] = tq;
    packed[L + 1] = tt;
    packed[L + 2] = (double)B;
    packed[L + 3] = (double)M;
    const double n = (double)B;
    const double loss_F = totF / (n * (double)M * (double)L);
    const double loss_qt = (tq * (double)balance_q + tt * (double)balance_t) / (n * (double)L);
    scalars[0] = (float)(loss_F + loss_qt);
    scalars[1] = (float)loss_F;
    scalars[2] = (float)loss_qt;
    scalars[3] = 0.0f;
    for (int l = 0; l < L; ++l) scalars[4 + l] = (float)(red[0][l] / (n * (double)M));  // losses.mean() of layer l
  }
}

}  // namespace

extern "C" int dfepe_loss_head(const float* loss_sum, const float* q_l2, const float* t_l2, int L, int B, int M, float clamp_q,
                               float clamp_t, float balance_q, float balance_t, double* packed, float* scalars, void* stream) {
  if (L <= 0 || L > kHeadMaxL || B <= 0 || M <= 0) return DFEPE_ERR_INVALID_ARG;
  if (!loss_sum || !packed || !scalars || ((q_l2 == nullptr) != (t_l2 == nullptr))) return DFEPE_ERR_INVALID_ARG;
  hipLaunchKernelGGL(loss_head_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), loss_sum, q_l2, t_l2, L, B, M,
                     clamp_q, clamp_t, balance_q, balance_t, packed, scalars);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_pose_fwd(const float* E_layers, int L, int B, const float* q_gt, const float* t_gt, const float* R_gt,
                              float* q_l2, float* t_l2, float* R_deg, float* t_deg, int* sel, void* stream) {
  if (L <= 0 || B < 0) return DFEPE_ERR_INVALID_ARG;
  if (B == 0) return DFEPE_OK;
  if (!E_layers || !q_gt || !t_gt || !q_l2 || !t_l2) return DFEPE_ERR_INVALID_ARG;
  if (R_deg && !R_gt) return DFEPE_ERR_INVALID_ARG;
  const size_t n = (size_t)L * B;
  hipLaunchKernelGGL(pose_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     E_layers, L, B, q_gt, t_gt, R_gt, q_l2, t_l2, R_deg, t_deg, sel);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_pose_bwd(const float* E_layers, int L, int B, const float* q_gt, const float* t_gt,
                              const float* g_q_l2, const float* g_t_l2, float coef_q, float clamp_q, float coef_t,
                              float clamp_t, const float* g_scale, float* g_E, void* stream) {
  if (L <= 0 || B < 0) return DFEPE_ERR_INVALID_ARG;
  if (B == 0) return DFEPE_OK;
  if (!E_layers || !q_gt || !t_gt || !g_E) return DFEPE_ERR_INVALID_ARG;
  const size_t n = (size_t)L * B;
  hipLaunchKernelGGL(pose_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     E_layers, L, B, q_gt, t_gt, g_q_l2, g_t_l2, coef_q, clamp_q, coef_t, clamp_t, g_scale, g_E);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
