// loss head of the fused loss tail: the batch sums of the per-workgroup partials that loss_tail_kernel leaves in its
// workspace -> per-layer means, the clamped qt mix, the packed sums of the data-parallel exchange (same quantities as
// dfepe_loss_head: train_good_utils.py:340-354 means, Train_model_pipeline.py:580-587 mixing).
// Run by ONE wavefront -- either a launch of its own (loss_tail_head_kernel) or a spare wavefront in workgroup 0 of the first
// backward fit of the step (w8pt16_bwd_head_kernel), where it should cost nothing: the backward does not need the scalars, and
// a launch of its own is 7.4 us on the critical path of a 120 us step.  For that it has to be shorter than the fit beside it
// (6 us), which the round-2 form (four wavefronts, a row of 48 doubles per thread, 15 wavefront sums through v_readlane, an LDS
// arrival counter, one lane finishing) was not: 10.8 us.  Now: the partials are element-major ([48][workgroups], so a load
// instruction of the wavefront touches 4 cache lines instead of 64), every lane adds the workgroups t, t + 64, ..., the 3 L sums
// are reduced inside the wavefront with DPP only (row steps, then row_bcast:15 / row_bcast:31: the total arrives in lane 63, no
// trip through the scalar file), and lane 63 finishes.  Deterministic: fixed order of additions, no floating-point atomics.
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

// sum over the 64 lanes of the wavefront, valid in lane 63 only
__device__ __forceinline__ double wave_sum_lane63(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror: every lane holds its row's sum
  union { double d; int i[2]; } a, r;
  a.d = v;  // rows 1 and 3 add lane 15 of the row before them (row_bcast:15, row_mask 0xA; the other rows add the `old` zero)
  r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x142, 0xa, 0xf, false);
  r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x142, 0xa, 0xf, false);
  v += r.d;
  a.d = v;  // rows 2 and 3 add lane 31 (row_bcast:31, row_mask 0xC)
  r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x143, 0xc, 0xf, false);
  r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x143, 0xc, 0xf, false);
  return v + r.d;
}

// lane = 0..63: the lane of the head's wavefront.  partials: [kTailParts][nblocks], element (kind * kTailMaxLayers + layer).
// The wavefront rides beside a fit wavefront on one SIMD of the first backward launch (w8pt16_bwd_head_kernel), so every instruction
// here delays that fit: the 3 L <= 48 wanted elements are walked as a compact list in groups of eight (one uniform branch per group,
// none per element), a workgroup count that is a multiple of 256 (every B that is a multiple of 4096) takes immediate load offsets
// and no clamps, and lane 63 folds each group into the totals as soon as it is reduced.
__device__ __forceinline__ void loss_head_run(const TailHead& H, const int lane) {
  const int L = H.L, nb = H.nblocks;
  constexpr int kGroup = 8, kGroups = kTailParts / kGroup;
  const bool full = (nb & 255) == 0;  // uniform
  double totF = 0.0, tq = 0.0, tt = 0.0;
#pragma unroll
  for (int g = 0; g < kGroups; ++g) {
    if (g * kGroup < 3 * L) {
      const double* src[kGroup];
      double acc[kGroup];
#pragma unroll
      for (int j = 0; j < kGroup; ++j) {
        const int e = g * kGroup + j;
        const int ec = (e < 3 * L) ? e : 0;                // uniform (scalar) arithmetic
        const int kind = (ec >= L) + (ec >= 2 * L);
        src[j] = H.partials + (size_t)(kind * kTailMaxLayers + (ec - kind * L)) * nb + lane;
        acc[j] = 0.0;
      }
      if (full) {
        for (int b = 0; b < nb; b += 256) {  // four workgroups per lane and trip: 32 independent loads in flight
          double x0[kGroup], x1[kGroup], x2[kGroup], x3[kGroup];
#pragma unroll
          for (int j = 0; j < kGroup; ++j) { x0[j] = src[j][b]; x1[j] = src[j][b + 64]; x2[j] = src[j][b + 128]; x3[j] = src[j][b + 192]; }
#pragma unroll
          for (int j = 0; j < kGroup; ++j) acc[j] += (x0[j] + x1[j]) + (x2[j] + x3[j]);
        }
      } else {
        for (int b = lane; b < nb; b += 64) {
#pragma unroll
          for (int j = 0; j < kGroup; ++j) acc[j] += src[j][b - lane];
        }
      }
      // fold the group into the totals with arithmetic masks (uniform selects, no branch per element); lane 63 stores the per-layer
      // sums in one exec region, strays (e >= L) aimed at a slot that is rewritten below
      double v[kGroup];
#pragma unroll
      for (int j = 0; j < kGroup; ++j) {
        const int e = g * kGroup + j;
        v[j] = wave_sum_lane63(acc[j]);
        totF = fma((e < L) ? 1.0 : 0.0, v[j], totF);
        tq = fma((e >= L && e < 2 * L) ? 1.0 : 0.0, v[j], tq);
        tt = fma((e >= 2 * L && e < 3 * L) ? 1.0 : 0.0, v[j], tt);
      }
      if (lane == 63) {
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
          const int e = g * kGroup + j;
          if (e < kTailMaxLayers) {  // compile time: later groups hold no layer sums
            H.packed[(e < L) ? e : L + 2] = v[j];
            H.scalars[(e < L) ? 4 + e : 3] = (float)(v[j] * H.inv_BM);  // losses.mean() of layer e
          }
        }
      }
    }
  }
  if (lane != 63) return;
  // lane 63 finishes (reciprocals come from the host: no fp64 division here)
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
