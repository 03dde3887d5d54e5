// dfepe_loss_tail -- the fused loss tail of the hot-path step (body: loss_tail_body.h): one 16-lane row per pair, 16 pairs
// per 256-thread workgroup, ONE launch for F-loss + E-from-F + pose errors + the loss-head sums + d loss / d F of every layer.
// The batch sums of the loss head are combined deterministically (per-workgroup partial sums in fixed row order, then a
// one-workgroup kernel that adds the partials in a fixed order) -- no floating-point atomics.
#include "dfepe_common.h"
#include "loss_tail_body.h"

namespace {

constexpr int kPairsPerBlock = 16;

// Workgroup = 16 pairs: wavefronts 0..3 run the F-loss rows (16 lanes per pair), the wavefronts after them the 3x3 work, one
// lane per (pair, layer) -- 16 L items, layer-major so that a wavefront's lanes run the same layer.  The two parts are
// independent until the final g_F = coef_F * (F-loss part) + (pose part), so they run CONCURRENTLY on the same SIMDs: B = 4096
// is one F-loss wavefront per SIMD, and a lone wavefront leaves a quarter of the issue slots and every memory wait unused.
// (In one row per pair, one part after the other, the kernel took 17 us.)
// Leading scalar / pointer arguments: preloaded into SGPRs at wave launch (-amdgpu-kernarg-preload-count, see w8pt16.hip); the
// kernel reads them instead of the copies inside A.
template <int IT>
__global__ void __launch_bounds__(512)
loss_tail_kernel(const float* F_layers, int L, int B, int M, int t_stride, const float* T1, const float* T2, const float* K,
                 const float* virt1, const TailArgs A0, double* __restrict__ partials) {
  __shared__ float lds[kPairsPerBlock][kTailLdsFloats];
  __shared__ double part[kPairsPerBlock][kTailParts];
  TailArgs A = A0;
  A.F_layers = F_layers; A.L = L; A.B = B; A.M = M; A.t_stride = t_stride; A.T1 = T1; A.T2 = T2; A.K = K; A.virt1 = virt1;
  const int pair0 = (int)blockIdx.x * kPairsPerBlock;
  for (int e = (int)threadIdx.x; e < kPairsPerBlock * kTailParts; e += (int)blockDim.x) (&part[0][0])[e] = 0.0;
  __syncthreads();
  const bool floss_wave = threadIdx.x < 256u;  // uniform per wavefront
  const int row = (int)(threadIdx.x >> 4) & 15;
  if (floss_wave) {
    if (pair0 + row < B) tail_floss_row<IT>(A, pair0 + row, lds[row], part[row], lds[row] + kTailMaxLayers * 9);
  } else {
    const int item = (int)threadIdx.x - 256;  // layer-major: item = layer * 16 + pair-in-workgroup
    const int layer = item >> 4, prow = item & 15;
    if (layer < L && pair0 + prow < B)
      tail_pose_item(A, pair0 + prow, layer, lds[prow] + 2 * kTailMaxLayers * 9 + layer * 9, &part[prow][kTailMaxLayers + layer],
                     &part[prow][2 * kTailMaxLayers + layer]);
  }
  __syncthreads();
  if (floss_wave && pair0 + row < B) tail_floss_finish(A, pair0 + row, lds[row] + kTailMaxLayers * 9, lds[row] + 2 * kTailMaxLayers * 9);
  // per-workgroup partial sums of the loss head, rows added in fixed order
  if (threadIdx.x < kTailParts) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < kPairsPerBlock; ++r) s += part[r][threadIdx.x];
    partials[(size_t)blockIdx.x * kTailParts + threadIdx.x] = s;
  }
}

// The batch sums: one workgroup adds the per-workgroup partials in a fixed order (every thread takes workgroups t, t + 256,
// ...: all loads in flight together; then DPP wave sums and a 4-way combine).  Deterministic, no floating-point atomics.
// (A one-wavefront version with four partials per lane was measured slower: 14.7 us against 8.9 us.)
// A launch of its own: the kernel boundary is what makes the partials of all XCDs visible, for less than an in-kernel
// "last workgroup" protocol costs in agent-scope fences on a multi-XCD part (measured: 24 us against 5 us).
struct TailHead {
  const double* partials;  // [nblocks][kTailParts]
  int nblocks, L, B, M, pose;
  double* packed;          // [L+4]
  float* scalars;          // [4+L]
  float balance_F, balance_q, balance_t;
  double inv_BM, inv_BML, inv_BL;  // 1 / (B M), 1 / (B M L), 1 / (B L)
};

__global__ void __launch_bounds__(256) loss_tail_head_kernel(const TailHead H) {
  __shared__ double red[4][kTailParts];
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  // every thread adds whole rows of partials (workgroups t, t + 256, ...): 24 independent 16-byte loads in flight per row;
  // entries of layers >= L are zeros the tail kernel wrote
  double v[3][kTailMaxLayers];
#pragma unroll
  for (int kind = 0; kind < 3; ++kind)
#pragma unroll
    for (int l = 0; l < kTailMaxLayers; ++l) v[kind][l] = 0.0;
  for (int b = (int)threadIdx.x; b < H.nblocks; b += 256) {
    const double2* row = reinterpret_cast<const double2*>(H.partials + (size_t)b * kTailParts);
    double2 t[kTailParts / 2];
#pragma unroll
    for (int k = 0; k < kTailParts / 2; ++k) t[k] = row[k];
#pragma unroll
    for (int k = 0; k < kTailParts / 2; ++k) {
      v[(2 * k) / kTailMaxLayers][(2 * k) % kTailMaxLayers] += t[k].x;
      v[(2 * k + 1) / kTailMaxLayers][(2 * k + 1) % kTailMaxLayers] += t[k].y;
    }
  }
#pragma unroll
  for (int kind = 0; kind < 3; ++kind)
#pragma unroll
    for (int l = 0; l < kTailMaxLayers; ++l) {
      if (l < H.L) {
        const double s = wave_sum(v[kind][l]);
        if (lane == 0) red[wave][kind * kTailMaxLayers + l] = s;
      }
    }
  __syncthreads();
  // same quantities as dfepe_loss_head; lane l < L finishes layer l, lane 0 the totals (reciprocals come from the host: a
  // dependent chain of fp64 divisions in one lane was a fifth of this kernel)
  const int L = H.L;
  if (threadIdx.x < (unsigned)L) {
    const int l = (int)threadIdx.x;
    const double f = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    H.packed[l] = f;
    H.scalars[4 + l] = (float)(f * H.inv_BM);  // losses.mean() of layer l
  }
  if (threadIdx.x == 0) {
    double totF = 0.0, tq = 0.0, tt = 0.0;
    for (int l = 0; l < L; ++l) {
      const int iq = kTailMaxLayers + l, it = 2 * kTailMaxLayers + l;
      totF += (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
      tq += (red[0][iq] + red[1][iq]) + (red[2][iq] + red[3][iq]);
      tt += (red[0][it] + red[1][it]) + (red[2][it] + red[3][it]);
    }
    H.packed[L] = tq;
    H.packed[L + 1] = tt;
    H.packed[L + 2] = (double)H.B;
    H.packed[L + 3] = (double)H.M;
    const double loss_F = totF * H.inv_BML;
    const double loss_qt = H.pose ? (tq * (double)H.balance_q + tt * (double)H.balance_t) * H.inv_BL : 0.0;
    H.scalars[0] = (float)((double)H.balance_F * loss_F + loss_qt);
    H.scalars[1] = (float)loss_F;
    H.scalars[2] = (float)loss_qt;
    H.scalars[3] = 0.0f;
  }
}

}  // namespace

extern "C" size_t dfepe_loss_tail_workspace_bytes(int B) {
  const size_t blocks = (size_t)((B > 0 ? B : 0) + kPairsPerBlock - 1) / kPairsPerBlock;
  return blocks * kTailParts * sizeof(double);
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
  double* partials = static_cast<double*>(workspace);
  // 4 F-loss wavefronts + one lane per (pair, layer): 16 L lanes, rounded up to wavefronts
  const dim3 grid((B + kPairsPerBlock - 1) / kPairsPerBlock), block(256 + 64 * ((kPairsPerBlock * L + 63) / 64));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (M <= 16) hipLaunchKernelGGL((loss_tail_kernel<1>), grid, block, 0, st, A.F_layers, A.L, A.B, A.M, A.t_stride, A.T1, A.T2, A.K, A.virt1, A, partials);
  else if (M <= 32) hipLaunchKernelGGL((loss_tail_kernel<2>), grid, block, 0, st, A.F_layers, A.L, A.B, A.M, A.t_stride, A.T1, A.T2, A.K, A.virt1, A, partials);
  else if (M <= 64) hipLaunchKernelGGL((loss_tail_kernel<4>), grid, block, 0, st, A.F_layers, A.L, A.B, A.M, A.t_stride, A.T1, A.T2, A.K, A.virt1, A, partials);
  else if (M <= 112) hipLaunchKernelGGL((loss_tail_kernel<7>), grid, block, 0, st, A.F_layers, A.L, A.B, A.M, A.t_stride, A.T1, A.T2, A.K, A.virt1, A, partials);
  else hipLaunchKernelGGL((loss_tail_kernel<8>), grid, block, 0, st, A.F_layers, A.L, A.B, A.M, A.t_stride, A.T1, A.T2, A.K, A.virt1, A, partials);
  TailHead H;
  H.partials = partials; H.nblocks = (int)grid.x; H.L = L; H.B = B; H.M = M; H.pose = (q_gt != nullptr) ? 1 : 0;
  H.packed = packed; H.scalars = scalars; H.balance_F = balance_F; H.balance_q = balance_q; H.balance_t = balance_t;
  H.inv_BM = 1.0 / ((double)B * (double)M); H.inv_BML = H.inv_BM / (double)L; H.inv_BL = 1.0 / ((double)B * (double)L);
  hipLaunchKernelGGL(loss_tail_head_kernel, dim3(1), dim3(256), 0, st, H);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
