// dfepe_loss_tail -- the fused loss tail of the hot-path step (body: loss_tail_body.h): one 16-lane row per pair, 16 pairs
// per 256-thread workgroup, ONE launch for F-loss + E-from-F + pose errors + the loss-head sums + d loss / d F of every layer.
// The batch sums of the loss head are combined deterministically: per-workgroup partial sums in fixed row order, then the
// last workgroup to finish (atomic ticket) adds the partials in workgroup order -- no floating-point atomics.
#include "dfepe_common.h"
#include "loss_tail_body.h"

namespace {

constexpr int kPairsPerBlock = 16;

struct TailHead {
  double* partials;    // [gridDim.x][kTailParts]
  unsigned* ticket;    // zero before the first launch; the kernel leaves it zero
  double* packed;      // [L+4]
  float* scalars;      // [4+L]
  float balance_F, balance_q, balance_t;
};

template <int IT>
__global__ void __launch_bounds__(256) loss_tail_kernel(const TailArgs A, const TailHead H) {
  __shared__ float lds[kPairsPerBlock][kTailLdsFloats];
  __shared__ double part[kPairsPerBlock][kTailParts];
  __shared__ double tot[kTailParts];
  __shared__ int is_last;
  const int row = (int)(threadIdx.x >> 4);
  const int pair = (int)blockIdx.x * kPairsPerBlock + row;
  for (int e = (int)(threadIdx.x & 15u); e < kTailParts; e += 16) part[row][e] = 0.0;
  rg_sync();
  if (pair < A.B) loss_tail_pair<IT>(A, pair, lds[row], part[row]);
  __syncthreads();
  if (threadIdx.x < kTailParts) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < kPairsPerBlock; ++r) s += part[r][threadIdx.x];
    H.partials[(size_t)blockIdx.x * kTailParts + threadIdx.x] = s;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) is_last = (atomicAdd(H.ticket, 1u) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  if (threadIdx.x < kTailParts) {
    double s = 0.0;
    for (unsigned b = 0; b < gridDim.x; ++b)
      s += __hip_atomic_load(H.partials + (size_t)b * kTailParts + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tot[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // same quantities as dfepe_loss_head
    const int L = A.L;
    double totF = 0.0, tq = 0.0, tt = 0.0;
    for (int l = 0; l < L; ++l) { H.packed[l] = tot[l]; totF += tot[l]; tq += tot[kTailMaxLayers + l]; tt += tot[2 * kTailMaxLayers + l]; }
    H.packed[L] = tq;
    H.packed[L + 1] = tt;
    H.packed[L + 2] = (double)A.B;
    H.packed[L + 3] = (double)A.M;
    const double n = (double)A.B;
    const double loss_F = totF / (n * (double)A.M * (double)L);
    const double loss_qt = (A.q_gt != nullptr) ? (tq * (double)H.balance_q + tt * (double)H.balance_t) / (n * (double)L) : 0.0;
    H.scalars[0] = (float)((double)H.balance_F * loss_F + loss_qt);
    H.scalars[1] = (float)loss_F;
    H.scalars[2] = (float)loss_qt;
    H.scalars[3] = 0.0f;
    for (int l = 0; l < L; ++l) H.scalars[4 + l] = (float)(tot[l] / (n * (double)A.M));  // losses.mean() of layer l
    *H.ticket = 0u;
  }
}

}  // namespace

extern "C" size_t dfepe_loss_tail_workspace_bytes(int B) {
  const size_t blocks = (size_t)((B > 0 ? B : 0) + kPairsPerBlock - 1) / kPairsPerBlock;
  return 64 + blocks * kTailParts * sizeof(double);
}

extern "C" int dfepe_loss_tail(const float* F_layers, int L, int B, const float* T1, const float* T2, int t_stride,
                               const float* K, const float* virt1, const float* virt2, int M, float clamp_at,
                               const float* q_gt, const float* t_gt, const float* R_gt, float clamp_q, float clamp_t,
                               float balance_F, float balance_q, float balance_t, double grad_pairs, float* loss_sum,
                               float* E_layers, float* q_l2, float* t_l2, float* R_deg, float* t_deg, int* sel,
                               float* g_F_layers, double* packed, float* scalars, void* workspace, void* stream) {
  if (L <= 0 || L > kTailMaxLayers || B <= 0 || M <= 0) return DFEPE_ERR_INVALID_ARG;
  if (M > 128) return DFEPE_ERR_UNSUPPORTED;  // the unfused kernels (dfepe_floss_*, dfepe_pose_*, dfepe_loss_head) serve larger grids
  if (t_stride != 0 && t_stride != 9) return DFEPE_ERR_INVALID_ARG;
  if (!F_layers || !T1 || !T2 || !K || !virt1 || !virt2 || !loss_sum || !E_layers || !packed || !scalars || !workspace)
    return DFEPE_ERR_INVALID_ARG;
  if (q_gt && (!t_gt || !q_l2 || !t_l2)) return DFEPE_ERR_INVALID_ARG;
  if (R_deg && !R_gt) return DFEPE_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 7u) || !(grad_pairs > 0.0)) return DFEPE_ERR_INVALID_ARG;
  TailArgs A;
  A.F_layers = F_layers; A.L = L; A.B = B; A.M = M; A.t_stride = t_stride; A.T1 = T1; A.T2 = T2; A.K = K;
  A.virt1 = virt1; A.virt2 = virt2; A.clamp_at = clamp_at; A.q_gt = q_gt; A.t_gt = t_gt; A.R_gt = R_gt;
  A.clamp_q = clamp_q; A.clamp_t = clamp_t;
  A.coef_F = (float)((double)balance_F / ((double)L * grad_pairs * (double)M));
  A.coef_q = (float)((double)balance_q / ((double)L * grad_pairs));
  A.coef_t = (float)((double)balance_t / ((double)L * grad_pairs));
  A.loss_sum = loss_sum; A.E_layers = E_layers; A.q_l2 = q_l2; A.t_l2 = t_l2; A.R_deg = R_deg; A.t_deg = t_deg; A.sel = sel;
  A.g_F = g_F_layers;
  TailHead H;
  H.ticket = static_cast<unsigned*>(workspace);
  H.partials = reinterpret_cast<double*>(static_cast<unsigned char*>(workspace) + 64);
  H.packed = packed; H.scalars = scalars; H.balance_F = balance_F; H.balance_q = balance_q; H.balance_t = balance_t;
  const dim3 grid((B + kPairsPerBlock - 1) / kPairsPerBlock), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (M <= 16) hipLaunchKernelGGL((loss_tail_kernel<1>), grid, block, 0, st, A, H);
  else if (M <= 32) hipLaunchKernelGGL((loss_tail_kernel<2>), grid, block, 0, st, A, H);
  else if (M <= 64) hipLaunchKernelGGL((loss_tail_kernel<4>), grid, block, 0, st, A, H);
  else if (M <= 112) hipLaunchKernelGGL((loss_tail_kernel<7>), grid, block, 0, st, A, H);
  else hipLaunchKernelGGL((loss_tail_kernel<8>), grid, block, 0, st, A, H);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
