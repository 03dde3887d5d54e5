// loss head of the fused loss tail: the batch sums of the per-workgroup partials that loss_tail_kernel leaves in its
// workspace -> per-layer means, the clamped qt mix, the packed sums of the data-parallel exchange (same quantities as
// dfepe_loss_head: train_good_utils.py:340-354 means, Train_model_pipeline.py:580-587 mixing).
// Run by THREE wavefronts (one per kind of partial) -- either a launch of its own (loss_tail_head_kernel) or three spare wavefronts
// in workgroup 0 of the first backward fit of the step (w8pt16_bwd_head_kernel), where it should cost nothing: the backward does
// not need the scalars, and a launch of its own is 7 us on the critical path of a 115 us step.  For that each head wavefront has to
// be much shorter than the fit wavefront it shares a SIMD with.  History: four wavefronts reading a row of 48 doubles per thread,
// 15 wavefront sums through v_readlane and one lane finishing (round 2) made that launch 11.2 us against 6.1 for a plain backward
// fit; one wavefront over element-major partials ([48][workgroups]: a load instruction touches 4 cache lines instead of 64) with
// DPP-only reductions into lane 63 made it 8.45; a third of that work on each of three SIMDs is the present form.
// Deterministic: fixed order of additions, no floating-point atomics.
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

struct TailHeadLds {
  double tot[3];     // batch totals of the three kinds: F-loss sums, clamped q_l2, clamped t_l2
  unsigned arrived;  // zeroed by the caller before any of the three wavefronts can arrive
};

// THREE wavefronts, one per kind of partial (wave = 0: per-layer F-loss sums, 1: clamp(q_l2), 2: clamp(t_l2)); lane = 0..63.
// partials: [kTailParts][nblocks], element (kind * kTailMaxLayers + layer).  The wavefronts ride beside fit wavefronts on three
// SIMDs of the first backward launch (w8pt16_bwd_head_kernel), so every instruction here delays a fit: one wavefront for all 3 L
// elements cost the fit beside it 2.4 us, a third of the work on each of three SIMDs costs ~1.  The L <= 16 layers of a kind are
// walked in groups of eight (one uniform branch per group, none per element; clamped loads, masked sums), a workgroup count that
// is a multiple of 256 (every B that is a multiple of 4096) takes immediate load offsets and no clamps, the sums are reduced with
// DPP only, and the three lanes 63 meet through LDS with an arrival counter (workgroup-scope release / acquire: no cache
// maintenance, no block barrier -- in the backward kernel the other wavefronts of the workgroup never reach one); the last one
// to arrive writes the scalars.
__device__ __forceinline__ void loss_head_run(const TailHead& H, const int lane, const int kind, TailHeadLds* lds) {
  const int L = H.L, nb = H.nblocks;
  constexpr int kGroup = 8, kGroups = kTailMaxLayers / kGroup;
  const bool full = (nb & 255) == 0;  // uniform
  const double* base = H.partials + (size_t)kind * kTailMaxLayers * nb + lane;
  double tot = 0.0;
#pragma unroll
  for (int g = 0; g < kGroups; ++g) {
    if (g * kGroup < L) {
      const double* src[kGroup];
      double acc[kGroup];
#pragma unroll
      for (int j = 0; j < kGroup; ++j) {
        const int l = g * kGroup + j;
        src[j] = base + (size_t)((l < L) ? l : 0) * nb;  // uniform (scalar) arithmetic
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
      // fold the group into the kind's total with arithmetic masks (uniform selects, no branch per element); lane 63 of the F-loss
      // wavefront stores the per-layer sums in one exec region, strays (l >= L) aimed at slots that are rewritten by the finisher
      double v[kGroup];
#pragma unroll
      for (int j = 0; j < kGroup; ++j) {
        v[j] = wave_sum_lane63(acc[j]);
        tot = fma((g * kGroup + j < L) ? 1.0 : 0.0, v[j], tot);
      }
      if (kind == 0 && lane == 63) {
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
          const int l = g * kGroup + j;
          H.packed[(l < L) ? l : L + 2] = v[j];
          H.scalars[(l < L) ? 4 + l : 3] = (float)(v[j] * H.inv_BM);  // losses.mean() of layer l
        }
      }
    }
  }
  if (lane != 63) return;
  lds->tot[kind] = tot;
  const unsigned prev = __hip_atomic_fetch_add(&lds->arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (prev != 2u) return;
  // the last lane 63 to arrive finishes (reciprocals come from the host: no fp64 division here)
  const double totF = lds->tot[0], tq = lds->tot[1], tt = lds->tot[2];
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
