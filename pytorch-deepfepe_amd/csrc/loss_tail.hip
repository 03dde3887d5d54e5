// dfepe_loss_tail -- the fused loss tail of the hot-path step (body: loss_tail_body.h): one 16-lane row per pair, 16 pairs
// per 256-thread workgroup, ONE launch for F-loss + E-from-F + pose errors + the loss-head sums + d loss / d F of every layer.
// The batch sums of the loss head are combined deterministically (per-workgroup partial sums in fixed row order, then a
// one-workgroup kernel that adds the partials in a fixed order) -- no floating-point atomics.
#include "dfepe_common.h"
#include "loss_tail_body.h"
#include "loss_head_body.h"

#ifndef DFEPE_TAIL_KL
#define DFEPE_TAIL_KL 2  // layers a lone F-loss wavefront walks together (A/B: -DDFEPE_TAIL_KL=3)
#endif

#ifndef DFEPE_TAIL_STAGGER
#define DFEPE_TAIL_STAGGER 0  // A/B switch: F-loss wavefront w of a workgroup starts w x this many cycles late (w8pt16.hip: DFEPE_FWD_STAGGER
                              // is worth 3 % there); measured here at 8 / 16 / 24 / 64 / 128 / 256 cycles: +-0 (0.1022-0.1038 vs 0.1023-0.1035 ms per step)
#endif

namespace {

constexpr int kPairsPerBlock = 16;
constexpr size_t kTailDescBytes = 256;  // start of the workspace: a TailHead for a deferred head
static_assert(sizeof(TailHead) <= kTailDescBytes, "descriptor slot too small");

// Workgroup = 16 pairs: wavefronts 0..3 run the F-loss rows (16 lanes per pair), the wavefronts after them the 3x3 work, one
// lane per (pair, layer) -- 16 L items, layer-major so that a wavefront's lanes run the same layer.  The two parts are
// independent until the final g_F = coef_F * (F-loss part) + (pose part), so they run CONCURRENTLY on the same SIMDs: B = 4096
// is one F-loss wavefront per SIMD, and a lone wavefront leaves a quarter of the issue slots and every memory wait unused.
// (In one row per pair, one part after the other, the kernel took 17 us.  Round 4 tried TEN wavefronts -- two F-loss wavefronts per
// SIMD, each with half of the layers of its pairs: three wavefronts then share two of the SIMDs, which caps the kernel at 168
// registers; the F-loss row (virtual points of both images in registers through the fp64 transform) and the pose lane each need
// ~200, the build spilled 130-220 bytes per lane and the launch took 19.7 us instead of 12.8: reverted.)
// Leading scalar / pointer arguments: preloaded into SGPRs at wave launch (-amdgpu-kernarg-preload-count, see w8pt16.hip); the
// kernel reads them instead of the copies inside A.
template <int IT, bool JAC = false>
__global__ void __launch_bounds__(512)
loss_tail_kernel(const float* F_layers, int L, int B, int M, int t_stride, const float* T1, const float* T2, const float* K,
                 const float* virt1, const TailArgs A0, double* __restrict__ partials, const TailHead Hd, const int write_desc) {
  __shared__ float lds[kPairsPerBlock][kTailLdsFloats];
  __shared__ double part[kPairsPerBlock][kTailParts];
  TailArgs A = A0;
  A.F_layers = F_layers; A.L = L; A.B = B; A.M = M; A.t_stride = t_stride; A.T1 = T1; A.T2 = T2; A.K = K; A.virt1 = virt1;
  const int pair0 = (int)blockIdx.x * kPairsPerBlock;
  // deferred head: its descriptor travels at the start of the workspace, in front of the partials
  if (write_desc && blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<TailHead*>(reinterpret_cast<char*>(partials) - kTailDescBytes) = Hd;
  // ... and until the first backward launch has run that head, the batch scalars read as NaN, not as uninitialised memory
  if (write_desc && blockIdx.x == 0 && (int)threadIdx.x < L + 4) {
    Hd.packed[threadIdx.x] = __longlong_as_double(0x7FF8000000000000LL);
    Hd.scalars[threadIdx.x] = __uint_as_float(0x7FC00000u);
  }
  if (!JAC) {
    for (int e = (int)threadIdx.x; e < kPairsPerBlock * kTailParts; e += (int)blockDim.x) (&part[0][0])[e] = 0.0;
    __syncthreads();
  }
  const bool floss_wave = threadIdx.x < 256u;  // uniform per wavefront
  const int row = (int)(threadIdx.x >> 4) & 15;
#if DFEPE_TAIL_STAGGER
  if (floss_wave) {
    for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) {
      if constexpr (DFEPE_TAIL_STAGGER >= 64) __builtin_amdgcn_s_sleep(DFEPE_TAIL_STAGGER / 64);
      else {
        if constexpr (DFEPE_TAIL_STAGGER > 16) asm volatile("s_nop 15");
        asm volatile("s_nop %0" ::"n"((DFEPE_TAIL_STAGGER - 1) & 15));
      }
    }
  }
#endif
  if (floss_wave) {
    if (pair0 + row < B) tail_floss_row<IT, JAC, DFEPE_TAIL_KL>(A, pair0 + row, lds[row], part[row], lds[row] + kTailMaxLayers * 9);
  } else {
    const int item = (int)threadIdx.x - 256;  // layer-major: item = layer * 16 + pair-in-workgroup
    const int layer = item >> 4, prow = item & 15;
    if (layer < L && pair0 + prow < B)
      tail_pose_item<JAC>(A, pair0 + prow, layer, lds[prow] + 2 * kTailMaxLayers * 9 + layer * 9, &part[prow][kTailMaxLayers + layer],
                          &part[prow][2 * kTailMaxLayers + layer]);
  }
  if (JAC) return;  // the Jacobians went straight to memory: no meeting point, no batch sums
  __syncthreads();
  if (floss_wave && pair0 + row < B) tail_floss_finish(A, pair0 + row, lds[row] + kTailMaxLayers * 9, lds[row] + 2 * kTailMaxLayers * 9);
  // per-workgroup partial sums of the loss head, rows added in fixed order
  if (threadIdx.x < kTailParts) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < kPairsPerBlock; ++r) s += part[r][threadIdx.x];
    partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;  // element-major: the head's loads are coalesced over the workgroups
  }
}

// The batch sums (loss_head_body.h).  A launch of its own: the kernel boundary is what makes the partials of all XCDs visible,
// for less than an in-kernel "last workgroup" protocol costs in agent-scope fences on a multi-XCD part (measured: 24 us against
// 5 us; a one-wavefront version: 14.7 us).  With defer_head the launch is left to the first backward fit of the step, which
// runs the same code in three spare wavefronts of its workgroup 0 (dfepe_w8pt_bwd, pending_loss_head).
__global__ void __launch_bounds__(192) loss_tail_head_kernel(const TailHead H) {
  __shared__ TailHeadLds lds;
  if (threadIdx.x == 0) lds.arrived = 0u;
  __syncthreads();
  loss_head_run(H, (int)(threadIdx.x & 63u), (int)(threadIdx.x >> 6), &lds);
}
// the same from a descriptor in device memory (the fallback of a deferred head when the backward fit is served by the
// wavefront-per-pair kernels)
__global__ void __launch_bounds__(192) loss_tail_head_desc_kernel(const TailHead* Hp) {
  __shared__ TailHeadLds lds;
  if (threadIdx.x == 0) lds.arrived = 0u;
  __syncthreads();
  const TailHead H = *Hp;
  loss_head_run(H, (int)(threadIdx.x & 63u), (int)(threadIdx.x >> 6), &lds);
}

// d loss / dF of every (layer, pair) from the Jacobians of a dfepe_loss_tail_jac launch and whatever upstream gradients the
// caller's own loss mixing produced: one thread per matrix entry.  Upstream per element (g_ls, g_q, g_t: [L,B]) and / or per
// statistic of dfepe_loss_stats (m_*: [L] gradients of the row means, o_*: [1] gradient of the mean of the row means; a gradient g
// on the mean of row l is g / B on each of its elements, one on the mean of the row means g / (L B)).
__global__ void __launch_bounds__(256)
loss_tail_bwd_kernel(const float* __restrict__ J, int L, int B, const float* __restrict__ g_ls, const float* __restrict__ g_q,
                     const float* __restrict__ g_t, const float* __restrict__ m_ls, const float* __restrict__ o_ls,
                     const float* __restrict__ m_q, const float* __restrict__ o_q, const float* __restrict__ m_t,
                     const float* __restrict__ o_t, float scale_ls, float* __restrict__ g_F) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n_items = (size_t)L * B;
  if (idx >= n_items * 9) return;
  const size_t item = idx / 9;
  const int c = (int)(idx - item * 9);
  const int l = (int)(item / (size_t)B);
  const float* j = J + item * 27 + c;
  const float invB = 1.0f / (float)B, invL = 1.0f / (float)L;
  // gradient of an element = its own + (that of its row's mean + that of the mean of the row means / L) / B
  auto stat = [&](const float* m, const float* o) { return ((m != nullptr) ? m[l] : 0.0f) + ((o != nullptr) ? o[0] * invL : 0.0f); };
  const float a = fmaf(stat(m_ls, o_ls), scale_ls * invB, (g_ls != nullptr) ? g_ls[item] : 0.0f);
  const float q = fmaf(stat(m_q, o_q), invB, (g_q != nullptr) ? g_q[item] : 0.0f);
  const float t = fmaf(stat(m_t, o_t), invB, (g_t != nullptr) ? g_t[item] : 0.0f);
  g_F[idx] = fmaf(a, j[0], fmaf(q, j[9], t * j[18]));
}

// ---- batch statistics of the per-pair loss terms: the means (and minima) get_all_loss_DeepF / get_Rt_loss return ------------
// Up to four sets of rows [R_k, C] (C = pairs); one 1024-thread workgroup per set walks its rows: row means (times the set's
// scale), the mean of the row means, for set 0 also the row minima and the column minima.  Fixed order of additions (thread t
// adds columns t, t + 1024, ... in fp64, wavefront sums, then the sixteen wavefront totals in order): deterministic.
struct StatSets {
  const float* x[4];
  int rows[4];
  float scale[4];
};
constexpr int kStatMaxRows = 64;

__global__ void __launch_bounds__(1024)
loss_stats_kernel(const StatSets S, int C, float* __restrict__ out, float* __restrict__ row_min, float* __restrict__ col_min) {
  constexpr int kG = 8;  // rows walked together: their loads are independent, one pass over the columns serves all of them
  __shared__ double wsum[kG][16];
  __shared__ float wmin[kG][16];
  __shared__ double rowmean[kStatMaxRows];
  const int k = (int)blockIdx.x;
  const int R = S.rows[k];
  if (R <= 0) return;
  int off = 0;
  for (int j = 0; j < k; ++j) off += (S.rows[j] > 0) ? S.rows[j] + 1 : 0;
  const float* x = S.x[k];
  const double scale = (double)S.scale[k];
  const int t = (int)threadIdx.x, wave = t >> 6, lane = t & 63;
  const bool mins = (k == 0) && (row_min != nullptr || col_min != nullptr);
  for (int r0 = 0; r0 < R; r0 += kG) {
    double acc[kG];
    float mn[kG];
#pragma unroll
    for (int j = 0; j < kG; ++j) { acc[j] = 0.0; mn[j] = INFINITY; }
    // four column strips per trip, every load of the trip issued before the first use (index clamped, contribution masked): C = 4096
    // is ONE memory round trip for this workgroup instead of four dependent ones (the rows were just written by another kernel on
    // other XCDs: every trip is a miss in this XCD's L2)
    constexpr int kU = 4;
    for (int c0 = t; c0 < C; c0 += 1024 * kU) {
      float v[kU][kG];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int c = c0 + 1024 * u, cc = (c < C) ? c : C - 1;
#pragma unroll
        for (int j = 0; j < kG; ++j) v[u][j] = x[(size_t)((r0 + j < R) ? r0 + j : r0) * C + cc];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int c = c0 + 1024 * u;
        if (c < C) {
          float cm = INFINITY;
#pragma unroll
          for (int j = 0; j < kG; ++j) {
            if (r0 + j < R) {
              acc[j] += (double)v[u][j];
              mn[j] = fminf(mn[j], v[u][j]);
              cm = fminf(cm, v[u][j]);
            }
          }
          if (mins && col_min != nullptr) {  // the same thread owns column c in every group of rows
            const float sv = (float)((double)cm * scale);
            col_min[c] = (r0 == 0) ? sv : fminf(col_min[c], sv);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kG; ++j) {
      const double s = wave_sum(acc[j]);
      const float m = -wave_max(-mn[j]);
      if (lane == 0) { wsum[j][wave] = s; wmin[j][wave] = m; }
    }
    __syncthreads();
    if (t < kG && r0 + t < R) {
      double s = 0.0;
      float m = INFINITY;
#pragma unroll
      for (int w = 0; w < 16; ++w) { s += wsum[t][w]; m = fminf(m, wmin[t][w]); }
      const double mean = s * scale / (double)C;
      rowmean[r0 + t] = mean;
      out[off + r0 + t] = (float)mean;
      if (mins && row_min != nullptr) row_min[r0 + t] = (float)((double)m * scale);
    }
    __syncthreads();
  }
  if (t == 0) {
    double s = 0.0;
    for (int r = 0; r < R; ++r) s += rowmean[r];
    out[off + R] = (float)(s / (double)R);
  }
}

}  // namespace

extern "C" size_t dfepe_loss_tail_workspace_bytes(int B) {
  const size_t blocks = (size_t)((B > 0 ? B : 0) + kPairsPerBlock - 1) / kPairsPerBlock;
  return kTailDescBytes + blocks * kTailParts * sizeof(double);  // the descriptor of a deferred head, then the partials
}

// launches the pending loss head described in `workspace` (dfepe_loss_tail with defer_head) as a kernel of its own
int dfepe_loss_head_from_workspace(const void* workspace_desc, hipStream_t st) {
  hipLaunchKernelGGL(loss_tail_head_desc_kernel, dim3(1), dim3(192), 0, st, static_cast<const TailHead*>(workspace_desc));
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// C-ABI form of the above: the head of a dfepe_loss_tail(..., defer_head = 1) call as a launch of its own on ANY stream that is
// ordered after that call -- e.g. a side stream that then carries the data-parallel all-reduce of `packed`, in parallel with the
// backward fits on the main stream (a fork / join inside one captured hipGraph).
extern "C" int dfepe_loss_head_pending(const void* workspace, void* stream) {
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15u)) return DFEPE_ERR_INVALID_ARG;
  return dfepe_loss_head_from_workspace(workspace, static_cast<hipStream_t>(stream));
}

extern "C" int dfepe_loss_tail(const float* F_layers, int L, int B, const float* T1, const float* T2, int t_stride,
                               const float* K, const float* virt1, const float* virt2, int M, float clamp_at,
                               const float* q_gt, const float* t_gt, const float* R_gt, float clamp_q, float clamp_t,
                               float balance_F, float balance_q, float balance_t, double grad_pairs, float* loss_sum,
                               float* E_layers, float* q_l2, float* t_l2, float* R_deg, float* t_deg, int* sel,
                               float* g_F_layers, double* packed, float* scalars, void* workspace, int defer_head,
                               void* stream) {
  if (L <= 0 || L > kTailMaxLayers || B <= 0 || M <= 0) return DFEPE_ERR_INVALID_ARG;
  if (M > 128) return DFEPE_ERR_UNSUPPORTED;  // the unfused kernels (dfepe_floss_*, dfepe_pose_*, dfepe_loss_head) serve larger grids
  if (t_stride != 0 && t_stride != 9) return DFEPE_ERR_INVALID_ARG;
  if (!F_layers || !T1 || !T2 || !K || !virt1 || !virt2 || !loss_sum || !E_layers || !packed || !scalars || !workspace)
    return DFEPE_ERR_INVALID_ARG;
  if (q_gt && (!t_gt || !q_l2 || !t_l2)) return DFEPE_ERR_INVALID_ARG;
  if (R_deg && !R_gt) return DFEPE_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 15u) || !(grad_pairs > 0.0)) return DFEPE_ERR_INVALID_ARG;
  TailArgs A;
  A.F_layers = F_layers; A.L = L; A.B = B; A.M = M; A.t_stride = t_stride; A.T1 = T1; A.T2 = T2; A.K = K;
  A.virt1 = virt1; A.virt2 = virt2; A.clamp_at = clamp_at; A.q_gt = q_gt; A.t_gt = t_gt; A.R_gt = R_gt;
  A.clamp_q = clamp_q; A.clamp_t = clamp_t;
  A.coef_F = (float)((double)balance_F / ((double)L * grad_pairs * (double)M));
  A.coef_q = (float)((double)balance_q / ((double)L * grad_pairs));
  A.coef_t = (float)((double)balance_t / ((double)L * grad_pairs));
  A.loss_sum = loss_sum; A.E_layers = E_layers; A.q_l2 = q_l2; A.t_l2 = t_l2; A.R_deg = R_deg; A.t_deg = t_deg; A.sel = sel;
  A.g_F = g_F_layers; A.J = nullptr;
  double* partials = reinterpret_cast<double*>(static_cast<char*>(workspace) + kTailDescBytes);
  // 4 F-loss wavefronts + one lane per (pair, layer): 16 L lanes, rounded up to wavefronts
  const dim3 grid((B + kPairsPerBlock - 1) / kPairsPerBlock), block(256 + 64 * ((kPairsPerBlock * L + 63) / 64));
  hipStream_t st = static_cast<hipStream_t>(stream);
  TailHead H;
  H.partials = partials; H.nblocks = (int)grid.x; H.L = L; H.B = B; H.M = M; H.pose = (q_gt != nullptr) ? 1 : 0;
  H.packed = packed; H.scalars = scalars; H.balance_F = balance_F; H.balance_q = balance_q; H.balance_t = balance_t;
  H.inv_BM = 1.0 / ((double)B * (double)M); H.inv_BML = H.inv_BM / (double)L; H.inv_BL = 1.0 / ((double)B * (double)L);
  const int wd = defer_head ? 1 : 0;
  if (M <= 16) hipLaunchKernelGGL((loss_tail_kernel<1>), grid, block, 0, st, A.F_layers, A.L, A.B, A.M, A.t_stride, A.T1, A.T2, A.K, A.virt1, A, partials, H, wd);
  else if (M <= 32) hipLaunchKernelGGL((loss_tail_kernel<2>), grid, block, 0, st, A.F_layers, A.L, A.B, A.M, A.t_stride, A.T1, A.T2, A.K, A.virt1, A, partials, H, wd);
  else if (M <= 64) hipLaunchKernelGGL((loss_tail_kernel<4>), grid, block, 0, st, A.F_layers, A.L, A.B, A.M, A.t_stride, A.T1, A.T2, A.K, A.virt1, A, partials, H, wd);
  else if (M <= 112) hipLaunchKernelGGL((loss_tail_kernel<7>), grid, block, 0, st, A.F_layers, A.L, A.B, A.M, A.t_stride, A.T1, A.T2, A.K, A.virt1, A, partials, H, wd);
  else hipLaunchKernelGGL((loss_tail_kernel<8>), grid, block, 0, st, A.F_layers, A.L, A.B, A.M, A.t_stride, A.T1, A.T2, A.K, A.virt1, A, partials, H, wd);
  if (defer_head) return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;  // the first backward fit runs the head
  hipLaunchKernelGGL(loss_tail_head_kernel, dim3(1), dim3(192), 0, st, H);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_loss_tail_jac(const float* F_layers, int L, int B, const float* T1, const float* T2, int t_stride,
                                   const float* K, const float* virt1, const float* virt2, int M, float clamp_at,
                                   const float* q_gt, const float* t_gt, const float* R_gt, int want_floss_jac, float* loss_sum,
                                   float* E_layers, float* q_l2, float* t_l2, float* R_deg, float* t_deg, int* sel, float* J,
                                   void* stream) {
  if (L <= 0 || L > kTailMaxLayers || B <= 0 || M <= 0) return DFEPE_ERR_INVALID_ARG;
  // dfepe_floss_fwd/bwd + dfepe_pose_fwd/bwd serve larger grids of virtual points (eight points per lane plus the Jacobian
  // bookkeeping would not fit the 256 registers of this 512-thread workgroup without scratch)
  if (M > 112) return DFEPE_ERR_UNSUPPORTED;
  if (t_stride != 0 && t_stride != 9) return DFEPE_ERR_INVALID_ARG;
  if (!F_layers || !T1 || !T2 || !K || !virt1 || !virt2 || !loss_sum || !E_layers || !J) return DFEPE_ERR_INVALID_ARG;
  if (q_gt && (!t_gt || !q_l2 || !t_l2)) return DFEPE_ERR_INVALID_ARG;
  if (R_deg && !R_gt) return DFEPE_ERR_INVALID_ARG;
  TailArgs A;
  A.F_layers = F_layers; A.L = L; A.B = B; A.M = M; A.t_stride = t_stride; A.T1 = T1; A.T2 = T2; A.K = K;
  A.virt1 = virt1; A.virt2 = virt2; A.clamp_at = clamp_at; A.q_gt = q_gt; A.t_gt = t_gt; A.R_gt = R_gt;
  A.clamp_q = 0.f; A.clamp_t = 0.f;
  A.coef_F = want_floss_jac ? 1.0f : 0.0f; A.coef_q = 0.f; A.coef_t = 0.f;
  A.loss_sum = loss_sum; A.E_layers = E_layers; A.q_l2 = q_l2; A.t_l2 = t_l2; A.R_deg = R_deg; A.t_deg = t_deg; A.sel = sel;
  A.g_F = nullptr; A.J = J;
  const dim3 grid((B + kPairsPerBlock - 1) / kPairsPerBlock), block(256 + 64 * ((kPairsPerBlock * L + 63) / 64));
  hipStream_t st = static_cast<hipStream_t>(stream);
  TailHead H = {};
#define DFEPE_TAIL_JAC(IT_) hipLaunchKernelGGL((loss_tail_kernel<IT_, true>), grid, block, 0, st, A.F_layers, A.L, A.B, A.M, A.t_stride, A.T1, A.T2, A.K, A.virt1, A, nullptr, H, 0)
  if (M <= 16) DFEPE_TAIL_JAC(1);
  else if (M <= 32) DFEPE_TAIL_JAC(2);
  else if (M <= 64) DFEPE_TAIL_JAC(4);
  else DFEPE_TAIL_JAC(7);
#undef DFEPE_TAIL_JAC
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_loss_tail_bwd(const float* J, int L, int B, const float* g_loss_sum, const float* g_q_l2, const float* g_t_l2,
                                   const float* g_mean_loss, const float* g_all_loss, const float* g_mean_q, const float* g_all_q,
                                   const float* g_mean_t, const float* g_all_t, float stat_loss_scale, float* g_F_layers,
                                   void* stream) {
  if (L <= 0 || B < 0) return DFEPE_ERR_INVALID_ARG;
  if (B == 0) return DFEPE_OK;
  if (!J || !g_F_layers) return DFEPE_ERR_INVALID_ARG;
  const size_t n = (size_t)L * B;
  hipLaunchKernelGGL(loss_tail_bwd_kernel, dim3((unsigned)((n * 9 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), J, L, B,
                     g_loss_sum, g_q_l2, g_t_l2, g_mean_loss, g_all_loss, g_mean_q, g_all_q, g_mean_t, g_all_t, stat_loss_scale, g_F_layers);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_loss_stats(const float* x0, int rows0, float scale0, const float* x1, int rows1, float scale1, const float* x2,
                                int rows2, float scale2, const float* x3, int rows3, float scale3, int C, float* out,
                                float* row_min0, float* col_min0, void* stream) {
  if (C <= 0 || !out) return DFEPE_ERR_INVALID_ARG;
  StatSets S;
  const float* xs[4] = {x0, x1, x2, x3};
  const int rs[4] = {rows0, rows1, rows2, rows3};
  const float sc[4] = {scale0, scale1, scale2, scale3};
  for (int k = 0; k < 4; ++k) {
    if (rs[k] < 0 || rs[k] > kStatMaxRows || (rs[k] > 0 && !xs[k])) return DFEPE_ERR_INVALID_ARG;
    S.x[k] = xs[k]; S.rows[k] = rs[k]; S.scale[k] = sc[k];
  }
  if ((row_min0 || col_min0) && rows0 <= 0) return DFEPE_ERR_INVALID_ARG;
  hipLaunchKernelGGL(loss_stats_kernel, dim3(4), dim3(1024), 0, static_cast<hipStream_t>(stream), S, C, out, row_min0, col_min0);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
