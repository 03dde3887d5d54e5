// loss head of the fused loss tail: the batch sums of the per-workgroup partials that loss_tail_kernel leaves in its
// workspace -> per-layer means, the clamped qt mix, the packed sums of the data-parallel exchange (same quantities as
// dfepe_loss_head: train_good_utils.py:340-354 means, Train_model_pipeline.py:580-587 mixing).
// Run by 256 threads (four wavefronts) of ONE workgroup -- either a launch of its own (loss_tail_head_kernel) or four extra
// wavefronts in workgroup 0 of the first backward fit of the step (w8pt16_bwd_kernel<..., HEAD>), where it costs nothing: the
// backward does not need the scalars, and a launch of its own is 7.4 us on the critical path of a 130 us step.
// The four wavefronts meet through LDS with an arrival counter (workgroup-scope release / acquire: a wait on the memory
// counters, no cache maintenance), not with a block barrier -- in the backward kernel the other wavefronts of the workgroup
// never reach one.  Deterministic: fixed order of additions, no floating-point atomics.
#pragma once
#include "dfepe_common.h"
#include "loss_tail_body.h"

struct TailHead {
  const double* partials;  // [nblocks][kTailParts]
  int nblocks, L, B, M, pose;
  double* packed;          // [L+4]
  float* scalars;          // [4+L]
  float balance_F, balance_q, balance_t;
  double inv_BM, inv_BML, inv_BL;  // 1 / (B M), 1 / (B M L), 1 / (B L)
};

struct TailHeadLds {
  double red[4][kTailParts];
  unsigned arrived;  // zeroed by the caller before any of the four wavefronts can arrive
};

// t = 0..255: index of the thread among the four head wavefronts
__device__ __forceinline__ void loss_head_run(const TailHead& H, const int t, TailHeadLds* lds) {
  const int lane = t & 63, wave = t >> 6;
  // every thread adds whole rows of partials (workgroups t, t + 256, ...): 24 independent 16-byte loads in flight per row;
  // entries of layers >= L are zeros the tail kernel wrote
  double v[3][kTailMaxLayers];
#pragma unroll
  for (int kind = 0; kind < 3; ++kind)
#pragma unroll
    for (int l = 0; l < kTailMaxLayers; ++l) v[kind][l] = 0.0;
  for (int b = t; b < H.nblocks; b += 256) {
    const double2* row = reinterpret_cast<const double2*>(H.partials + (size_t)b * kTailParts);
    double2 r[kTailParts / 2];
#pragma unroll
    for (int k = 0; k < kTailParts / 2; ++k) r[k] = row[k];
#pragma unroll
    for (int k = 0; k < kTailParts / 2; ++k) {
      v[(2 * k) / kTailMaxLayers][(2 * k) % kTailMaxLayers] += r[k].x;
      v[(2 * k + 1) / kTailMaxLayers][(2 * k + 1) % kTailMaxLayers] += r[k].y;
    }
  }
#pragma unroll
  for (int kind = 0; kind < 3; ++kind)
#pragma unroll
    for (int l = 0; l < kTailMaxLayers; ++l) {
      if (l < H.L) {
        const double s = wave_sum(v[kind][l]);
        if (lane == 0) lds->red[wave][kind * kTailMaxLayers + l] = s;
      }
    }
  // the last of the four wavefronts to arrive finishes (lane 0 of each wavefront counts; the result is broadcast)
  unsigned prev = 0;
  if (lane == 0) prev = __hip_atomic_fetch_add(&lds->arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  prev = (unsigned)__builtin_amdgcn_readfirstlane((int)prev);
  if (prev != 3u) return;
  const int L = H.L;
  auto tot = [&](int k) { return (lds->red[0][k] + lds->red[1][k]) + (lds->red[2][k] + lds->red[3][k]); };
  // lane l < L finishes layer l, lane 0 the totals (reciprocals come from the host: no chain of fp64 divisions in one lane)
  if (lane < L) {
    const double f = tot(lane);
    H.packed[lane] = f;
    H.scalars[4 + lane] = (float)(f * H.inv_BM);  // losses.mean() of layer l
  }
  if (lane == 0) {
    double totF = 0.0, tq = 0.0, tt = 0.0;
    for (int l = 0; l < L; ++l) {
      totF += tot(l);
      tq += tot(kTailMaxLayers + l);
      tt += tot(2 * kTailMaxLayers + l);
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
