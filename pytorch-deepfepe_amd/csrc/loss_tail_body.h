// loss_tail -- the whole loss tail of the hot-path step for one image pair, forward AND backward, in one 16-lane row:
//   F-loss on the virtual points for every layer      get_all_loss_DeepF, deepFEPE/train_good_utils.py:325-358
//   E_l = K^T T2^T F_l T1 K                            train_good_utils.py:356-358
//   pose errors / metrics of every layer               get_Rt_loss, train_good_utils.py:96-239 (pose_math.h)
//   d loss / d F_l  for  loss = balance_F mean(F-loss) + balance_q mean(clamp q_l2) + balance_t mean(clamp t_l2)
// Every gradient coefficient of such a loss is a constant known before the launch (means and clamps only), so the adjoint
// of each term is formed right where its forward value is: the epipolar terms of a virtual point are computed once and feed
// both the loss sum and the gradient sum; the pose adjoint follows the pose errors in the same lane.  This replaces five
// launches (floss_fwd, pose_fwd, loss_head, pose_bwd, floss_bwd: ~47 us of mostly launch latency and dependent global loads
// at B = 4096) by one.  Two parts: the F-loss rows (lanes stride over the M <= 128 virtual points, transformed once and kept in
// registers for all layers) and the 3x3 work of each (pair, layer) in one lane (E, pose forward + adjoint).  Written against
// rowgroup.h only (tests/emu/).
#pragma once
#include <type_traits>

#include <rowgroup.h>  // angle brackets on purpose: tests/emu/ substitutes its host emulation through the include path

#include "dfepe.h"
#include "dfepe_math.h"
#include "pose_math.h"

constexpr int kTailMaxLayers = 16;
constexpr int kTailLdsFloats = 3 * kTailMaxLayers * 9;  // per pair: F of every layer, the F-loss and the pose parts of d loss / d F
constexpr int kTailParts = 3 * kTailMaxLayers;          // per pair: loss_sum[l], clamp(q_l2[l]), clamp(t_l2[l])

struct TailArgs {
  const float* F_layers;  // [L,B,9]
  int L, B, M, t_stride;
  const float* T1;
  const float* T2;
  const float* K;
  const float* virt1;
  const float* virt2;
  float clamp_at;
  const float* q_gt;  // nullptr: no pose part
  const float* t_gt;
  const float* R_gt;  // may be nullptr (no R_deg)
  float clamp_q, clamp_t;
  float coef_F, coef_q, coef_t;  // d loss / d loss_sum[l,b], d loss / d clamp(q_l2[l,b]), d loss / d clamp(t_l2[l,b])
  float* loss_sum;    // [L,B]
  float* E_layers;    // [L,B,9]
  float* q_l2;        // [L,B] ...
  float* t_l2;
  float* R_deg;
  float* t_deg;
  int* sel;
  float* g_F;         // [L,B,9] or nullptr (forward only)
  float* J;           // JAC instantiations only: [L,B,27] = d loss_sum / dF | d q_l2 / dF | d t_l2 / dF of every (layer, pair)
};

// Per-point epipolar terms in fp32 (the reference computes the F-loss in fp32, train_good_utils.py:340-342; utils_F.py:402-411)
struct TailEpi {
  float d, dd, n1, n2, i1, i2;
  float l1[3], l2[3];
};
__device__ __forceinline__ TailEpi tail_epi_terms(const float* x1, const float* x2, const float* o) {
  TailEpi e;
#pragma unroll
  for (int c = 0; c < 3; ++c) e.l1[c] = fmaf(x2[0], o[c], fmaf(x2[1], o[3 + c], x2[2] * o[6 + c]));
#pragma unroll
  for (int r = 0; r < 3; ++r) e.l2[r] = fmaf(o[3 * r], x1[0], fmaf(o[3 * r + 1], x1[1], o[3 * r + 2] * x1[2]));
  e.dd = fmaf(x1[0], e.l1[0], fmaf(x1[1], e.l1[1], x1[2] * e.l1[2]));
  e.n1 = hw_sqrt(fmaf(e.l1[0], e.l1[0], e.l1[1] * e.l1[1]));
  e.n2 = hw_sqrt(fmaf(e.l2[0], e.l2[0], e.l2[1] * e.l2[1]));
  e.i1 = hw_rcp(e.n1 + 1e-6f);
  e.i2 = hw_rcp(e.n2 + 1e-6f);
  e.d = fabsf(e.dd) * (e.i1 + e.i2);
  return e;
}

// T * pixel in fp64 (pixels ~1e3 times 2/W minus 1 cancels ~3 digits), rounded once to fp32 like the reference's pts_eval
__device__ __forceinline__ void tail_eval_point(const float* v, const double* T, float* x) {
  const double a = v[0], b = v[1], c = v[2];
#pragma unroll
  for (int r = 0; r < 3; ++r) x[r] = (float)(T[3 * r] * a + T[3 * r + 1] * b + T[3 * r + 2] * c);
}

// ---- the 3x3 part of ONE (pair, layer): E = K^T T2^T F T1 K, its pose errors and their adjoint ----------------------------
// Plain single-lane code (no row primitives).  gpose: 9 floats, the pose part of d loss / d F of this (pair, layer);
// part_q / part_t: where clamp(q_l2), clamp(t_l2) of this item go for the loss-head sums.
// JAC: instead of the adjoint for the launch's own coefficients, the two Jacobians d q_l2 / dF and d t_l2 / dF (unclamped) go to
// A.J[.., 9..26] -- the caller mixes clamps and balances itself (Train_model_pipeline.py:580-586) and dfepe_loss_tail_bwd applies
// whatever upstream gradients arrive.
template <bool JAC = false>
__device__ __forceinline__ void tail_pose_item(const TailArgs& A, const int pair, const int layer, float* gpose, double* part_q,
                                               double* part_t) {
  const int B = A.B;
  // every global load first (null-safe addresses): one memory round trip
  double t1[9], t2[9], k[9], o[9];
  float qg[4], tg[3], Rg[9];
  const bool has_pose = A.q_gt != nullptr, has_R = has_pose && A.R_gt != nullptr;
  {
    const float* safe = A.K + (size_t)pair * 9;
    const float* qp = has_pose ? A.q_gt + (size_t)pair * 4 : safe;
    const float* tp = has_pose ? A.t_gt + (size_t)pair * 3 : safe;
    const float* rp = has_R ? A.R_gt + (size_t)pair * 9 : safe;
    const float* fp = A.F_layers + ((size_t)layer * B + pair) * 9;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      t1[c] = A.T1[(size_t)pair * A.t_stride + c]; t2[c] = A.T2[(size_t)pair * A.t_stride + c];
      k[c] = safe[c]; o[c] = fp[c]; Rg[c] = rp[c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) qg[c] = qp[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) tg[c] = tp[c];
  }
  // consumers of the loads above must not sink below the stores of E (the memory counter is in-order: they would wait for the
  // stores to complete)
#pragma unroll
  for (int c = 0; c < 9; ++c) rg_pin(Rg[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) rg_pin(qg[c]);
#pragma unroll
  for (int c = 0; c < 3; ++c) rg_pin(tg[c]);
  double Am[9], Cm[9], tmp[9], e[9];
  mat3_mul(t2, k, Am);  // A = T2 K, C = T1 K:  E = A^T F C
  mat3_mul(t1, k, Cm);
  const size_t lb = (size_t)layer * B + pair;
  mat3_mul_tn(Am, o, tmp);
  mat3_mul(tmp, Cm, e);
  float Ef[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    Ef[c] = (float)e[c]; A.E_layers[lb * 9 + c] = Ef[c];
    if (!JAC) gpose[c] = 0.0f;
  }
  if (!has_pose) {
    if (JAC) {
#pragma unroll
      for (int c = 9; c < 27; ++c) A.J[lb * 27 + c] = 0.0f;
    }
    return;
  }
  Pose P;
  pose_forward(Ef, qg, tg, P);  // on the fp32 E, like dfepe_pose_fwd on E_layers
  const double qe = P.qe[P.qi], te = P.te[P.ti];
  A.q_l2[lb] = (float)qe;
  A.t_l2[lb] = (float)te;
  if (A.sel != nullptr) A.sel[lb] = P.qi | (P.ti << 1);
  if (A.R_deg != nullptr && A.R_gt != nullptr) A.R_deg[lb] = (float)pose_R_deg(P, Rg);
  if (A.t_deg != nullptr) A.t_deg[lb] = (float)pose_t_deg(P);
  if (JAC) {
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      double gE[9], add[9];
      pose_backward(P, qg, which == 0 ? 1.0 : 0.0, which == 0 ? 0.0 : 1.0, gE);
#pragma unroll
      for (int c = 0; c < 9; ++c) gE[c] = (double)(float)gE[c];  // like dfepe_pose_bwd's fp32 g_E
      mat3_mul(Am, gE, tmp);
      mat3_mul_nt(tmp, Cm, add);
#pragma unroll
      for (int c = 0; c < 9; ++c) A.J[lb * 27 + 9 + 9 * which + c] = (float)add[c];
    }
    return;
  }
  *part_q = (double)fminf(fmaxf((float)qe, 0.0f), A.clamp_q);
  *part_t = (double)fminf(fmaxf((float)te, 0.0f), A.clamp_t);
  if (A.g_F != nullptr) {
    // torch.clamp passes the gradient on [min, max] inclusive
    const double gql = ((float)qe <= A.clamp_q) ? (double)A.coef_q : 0.0;
    const double gtl = ((float)te <= A.clamp_t) ? (double)A.coef_t : 0.0;
    double gE[9], add[9];
    pose_backward(P, qg, gql, gtl, gE);
    // through dfepe_pose_bwd's fp32 g_E, then E = A^T F C:  g_F += A g_E C^T
#pragma unroll
    for (int c = 0; c < 9; ++c) gE[c] = (double)(float)gE[c];
    mat3_mul(Am, gE, tmp);
    mat3_mul_nt(tmp, Cm, add);
#pragma unroll
    for (int c = 0; c < 9; ++c) gpose[c] = (float)add[c];
  }
}

// ---- the F-loss of every layer of one pair over its virtual points, with its gradient: one 16-lane row ---------------------
// Phase 1 (tail_floss_row): everything that does not need the pose part; leaves the per-layer gradient sums of lane c < 9 in
// gsum[ly] and writes loss_sum / part[ly].  Phase 2 (tail_floss_finish), after the pose items of this pair are done:
// g_F = coef_F * gsum + gpose (lane c < 9 reads back what it wrote).
// ldsF, gsum: kTailMaxLayers * 9 floats each, private to the row.
// ly_begin .. ly_end: the layers this row works on (the kernel gives each half of a pair's layers to a row in a different
// wavefront, both rows transforming the pair's virtual points: two shorter instruction streams per SIMD instead of one long one);
// KL: layers walked together by one row (2 = a second independent stream for a wavefront that is alone on its SIMD).
template <int IT, bool JAC = false, int KL = 2>
__device__ __forceinline__ void tail_floss_row(const TailArgs& A, const int pair, float* ldsF, double* part, float* gsum,
                                               const int ly_begin = 0, const int ly_end_in = -1) {
  const int l = rg_lane();
  const int L = A.L, B = A.B, M = A.M;
  // every global load first: transforms, the pair's virtual points (index clamped, masked afterwards), F of every layer
  double t1[9], t2[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) { t1[c] = A.T1[(size_t)pair * A.t_stride + c]; t2[c] = A.T2[(size_t)pair * A.t_stride + c]; }
  float x1[IT][3], x2[IT][3], vm[IT];
  float r1[IT][3], r2[IT][3];
  const float* v1 = A.virt1 + (size_t)pair * M * 3;
  const float* v2 = A.virt2 + (size_t)pair * M * 3;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int i = it * 16 + l;
    const int ic = (i < M) ? i : M - 1;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r1[it][k] = v1[3 * ic + k]; r2[it][k] = v2[3 * ic + k]; }
    vm[it] = (i < M) ? 1.0f : 0.0f;
  }
  constexpr int kPer = (kTailMaxLayers * 9 + 15) / 16;
  float fl[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int e = k * 16 + l, ec = (e < L * 9) ? e : 0;
    const int ly = ec / 9, c = ec - 9 * ly;
    fl[k] = A.F_layers[((size_t)ly * B + pair) * 9 + c];
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    tail_eval_point(r1[it], t1, x1[it]);
    tail_eval_point(r2[it], t2, x2[it]);
  }
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int e = k * 16 + l;
    if (e < L * 9) ldsF[e] = fl[k];
  }
  rg_sync();
  // with balance_F = 0 (the reference's objective when if_qt_loss, Train_model_pipeline.py:580-587) the F-loss is evaluated but
  // carries no gradient: its adjoint (more than half of this function) is skipped
  // JAC: the gradient sums are the Jacobian d loss_sum / dF itself (A.J[.., 0..8]); coef_F == 0 there means "not wanted" (zeros)
  const bool grad = (JAC || A.g_F != nullptr) && A.coef_F != 0.0f;
  // K layers at a time: the layers are independent of each other, and a lone wavefront on its SIMD (the F-loss wavefronts are the
  // critical path of this kernel, scripts/tail_time.py) needs the second instruction stream to fill the dependent-issue bubbles of
  // the first; it also halves the trips through the loop's scalar bookkeeping.
  auto layers = [&](auto kc, const int ly0) {
    constexpr int K = decltype(kc)::value;
    float o[K][9], accf[K], gof[K][9];
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int c = 0; c < 9; ++c) { o[k][c] = ldsF[(ly0 + k) * 9 + c]; gof[k][c] = 0.0f; }
      accf[k] = 0.0f;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const TailEpi e = tail_epi_terms(x1[it], x2[it], o[k]);
        accf[k] = fmaf(vm[it], fminf(e.d, A.clamp_at), accf[k]);
        if (grad) {
          // d d / d F[r][c] = sg S x2[r] x1[c] - k1 l1[c] x2[r] [c<2] - k2 l2[r] x1[c] [r<2]
          //                 = x2[r] a[c] - b[r] x1[c],  a[c] = sg S x1[c] - k1 l1[c] [c<2],  b[r] = k2 l2[r] [r<2]
          const float mk = (e.d <= A.clamp_at) ? vm[it] : 0.0f;  // clamp(max=) passes the gradient up to and including the bound
          const float S = e.i1 + e.i2, ad = fabsf(e.dd);
          const float sg = (e.dd > 0.0f) ? mk : ((e.dd < 0.0f) ? -mk : 0.0f);
          const float k1 = (e.n1 > 0.0f) ? mk * ad * e.i1 * e.i1 * hw_rcp(e.n1) : 0.0f;
          const float k2 = (e.n2 > 0.0f) ? mk * ad * e.i2 * e.i2 * hw_rcp(e.n2) : 0.0f;
          const float sS = sg * S;
          const float a0 = fmaf(sS, x1[it][0], -k1 * e.l1[0]), a1 = fmaf(sS, x1[it][1], -k1 * e.l1[1]), a2 = sS * x1[it][2];
          const float b0 = k2 * e.l2[0], b1 = k2 * e.l2[1];
          const float av[3] = {a0, a1, a2};
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            gof[k][c] += fmaf(x2[it][0], av[c], -b0 * x1[it][c]);
            gof[k][3 + c] += fmaf(x2[it][1], av[c], -b1 * x1[it][c]);
            gof[k][6 + c] = fmaf(x2[it][2], av[c], gof[k][6 + c]);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int ly = ly0 + k;
      const float acc = rg_sum(accf[k]);  // <= 128 terms of at most clamp_at each: fp32 like the reference's own sum
      if (l == 0) {
        A.loss_sum[(size_t)ly * B + pair] = acc;
        if (!JAC) part[ly] = (double)acc;
      }
      float mine = 0.0f;
      if (grad) {
#pragma unroll
        for (int c = 0; c < 9; ++c) {
          const float tot = rg_sum(gof[k][c]);
          mine = (l == c) ? tot : mine;
        }
        if (!JAC && l < 9) gsum[ly * 9 + l] = mine;
      }
      if (JAC && l < 9) A.J[((size_t)ly * B + pair) * 27 + l] = mine;
    }
  };
  const int ly_end = (ly_end_in < 0) ? L : ly_end_in;
  int ly = ly_begin;
  if constexpr (KL >= 3) {
    for (; ly + 2 < ly_end; ly += 3) layers(std::integral_constant<int, 3>{}, ly);
  }
  if constexpr (KL >= 2) {
    for (; ly + 1 < ly_end; ly += 2) layers(std::integral_constant<int, 2>{}, ly);
  }
  for (; ly < ly_end; ++ly) layers(std::integral_constant<int, 1>{}, ly);
}
__device__ __forceinline__ void tail_floss_finish(const TailArgs& A, const int pair, const float* gsum, const float* gpose /*[L][9]*/) {
  const int l = rg_lane();
  if (A.g_F == nullptr || l >= 9) return;
  const bool floss_grad = A.coef_F != 0.0f;  // else gsum was not written
  for (int ly = 0; ly < A.L; ++ly)
    A.g_F[((size_t)ly * A.B + pair) * 9 + l] = floss_grad ? fmaf(A.coef_F, gsum[ly * 9 + l], gpose[ly * 9 + l]) : gpose[ly * 9 + l];
}

// One pair in ONE row, both parts in sequence (lane l < L takes layer l's 3x3 work): what tests/emu/ runs; the kernel
// (loss_tail.hip) runs the two parts in different wavefronts of a workgroup so that they overlap.
// lds: kTailLdsFloats floats, part: kTailParts doubles (zero-initialised by the caller), both private to this pair's row
template <int IT>
__device__ __forceinline__ void loss_tail_pair(const TailArgs& A, const int pair, float* lds, double* part) {
  const int l = rg_lane();
  float* ldsF = lds;                        // [L][9] F of every layer
  float* gsum = lds + kTailMaxLayers * 9;       // [L][9] F-loss part of d loss / d F (before coef_F)
  float* gpose = lds + 2 * kTailMaxLayers * 9;  // [L][9] pose part of d loss / d F
  if (l < A.L) tail_pose_item(A, pair, l, gpose + l * 9, part + kTailMaxLayers + l, part + 2 * kTailMaxLayers + l);
  rg_sync();
  tail_floss_row<IT>(A, pair, ldsF, part, gsum);
  tail_floss_finish(A, pair, gsum, gpose);
}
