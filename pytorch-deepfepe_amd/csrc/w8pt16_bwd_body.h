// w8pt16 backward -- analytic adjoint of w8pt16_fwd_pair, one 16-lane row per image pair (any N; IT = 0 re-reads per pass).
//
// Replaces torch.autograd's replay of the per-sample torch.svd calls of Fit.weighted_svd
// (deepFEPE/models/DeepFNet.py:232-256) with closed forms:
//   epipolar residual  d_i(out)            -> g_out      (only when g_epi is given)
//   out = T2^T F' T1                        -> g_F' = T2 g_out T1^T
//   F' = F - s3 u3 v3^T  (rank-2 projection) -> g_F    (3x3 SVD adjoint restricted to the dropped triplet)
//   F = reshape(f), r = X f                 -> g_f = vec(g_F) + X^T g_r
//   f = eigenvector of M = X^T X            -> u = sum_k q_k (q_k . g_f) / (lam_f - lam_k) = -(M - lam_f I)^+ g_f
//   X_i = w_i ph_i                          -> g_w_i = 2 w_i (ph_i.f)(ph_i.u) + g_r_i (ph_i.f)
// The pseudo-inverse is applied in the tridiagonal form the forward saved (M / trace = H T H^T):
//   u = -(1/trace) H (T - lam I)^+ H^T g_f,
// with (T - lam I)^+ b obtained by pinning the twist component to zero -- which splits the singular system into two
// definite tridiagonal blocks, one Thomas sweep each -- and projecting the null vector z out before and after.  No other
// eigenpair is needed, so nothing of the forward's work is repeated.  All of it in fp64 on values the whole row shares.
#pragma once
#include "w8pt16_body.h"

struct W8BwdArgs {
  const float* pts1;
  const float* pts2;
  const float* wts;
  int B, Bm, N;
  float hw_sx, hw_sy, clamp_at;
  const float* save;
  const float* F_out;
  const float* g_F;
  const float* g_res;
  const float* g_epi;
  const float* g_w_extra;
  const float* g_scale;  // one float scaling g_F, g_res, g_epi (nullptr = 1)
  float* g_w;
  float* g_p1;
  float* g_p2;
  int logits_mode;
  unsigned variant;  // DFEPE_W8PT_NO_ROWNORM (Fit(normalize_SVD=False), DeepFNet.py:211): rows enter X un-normalised; 0 otherwise
  const void* pending_head;  // host side only: a deferred loss head to run beside this launch (dfepe_w8pt_bwd), or nullptr
  bool row_per_pair;         // host side only: never the cooperative workgroup (DFEPE_W8PT_ROW_PER_PAIR)
};

__device__ __forceinline__ double guard_den16(double d) {
  // keep the sign, floor the magnitude: repeated eigen/singular values give a large-but-finite gradient, never NaN
  const double lim = 1e-30;
  return (fabs(d) < lim) ? ((d < 0.0) ? -lim : lim) : d;
}

// y = (T - lam I)^+ g for the unit-trace tridiagonal (td, te), its eigenpair (lam, z) and the twist index.
// Uniform over the row.  g is overwritten.
__device__ __forceinline__ void tri_pinv_apply(const double* td, const double* te, double lam, const double* z, int twist,
                                               double* g, double* y) {
  double zg = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) zg = fma(z[k], g[k], zg);
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = fma(-zg, z[k], g[k]);
  // Thomas elimination from the top (rows above the twist) and from the bottom (rows below it); with y_twist = 0 the
  // two blocks decouple and each is a principal sub-matrix of the positive semi-definite T - lam I that excludes the
  // largest component of its null vector: definite, pivots away from zero.
  double rp[9], gp[9], rm[9], gm[9];  // reciprocal pivots, eliminated right-hand sides
  rp[0] = rcp_nr<1, false>(guard_den16(td[0] - lam));
  gp[0] = g[0];
#pragma unroll
  for (int k = 1; k < 9; ++k) {
    const double m = te[k - 1] * rp[k - 1];
    rp[k] = rcp_nr<1, false>(guard_den16((td[k] - lam) - m * te[k - 1]));
    gp[k] = g[k] - m * gp[k - 1];
  }
  rm[8] = rcp_nr<1, false>(guard_den16(td[8] - lam));
  gm[8] = g[8];
#pragma unroll
  for (int k = 7; k >= 0; --k) {
    const double m = te[k] * rm[k + 1];
    rm[k] = rcp_nr<1, false>(guard_den16((td[k] - lam) - m * te[k]));
    gm[k] = g[k] - m * gm[k + 1];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) y[k] = 0.0;
#pragma unroll
  for (int k = 7; k >= 0; --k) {
    const double yk = (gp[k] - te[k] * y[k + 1]) * rp[k];
    y[k] = (k < twist) ? yk : y[k];
  }
#pragma unroll
  for (int k = 1; k < 9; ++k) {
    const double yk = (gm[k] - te[k - 1] * y[k - 1]) * rm[k];
    y[k] = (k > twist) ? yk : y[k];
  }
  double zy = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) zy = fma(z[k], y[k], zy);
#pragma unroll
  for (int k = 0; k < 9; ++k) y[k] = fma(-zy, z[k], y[k]);
}

// PLAIN: no variant flag is set (the hot instantiations carry no test for them); otherwise A.variant may hold NO_ROWNORM
// (p^ = p: the row factor `inv` is 1 and constant), without point gradients.
// LDS block of the cooperative variant (ROWS = 16: one workgroup per pair, see W8Coop in w8pt16_body.h)
struct W8BwdCoop {
  float red[20][16];  // pass-A sums and the softmax-adjoint dot product: [value][row]
  double u[9];        // the eigenvector adjoint, published by row 0
};

// UP = false: the caller has no upstream gradient besides g_F (g_residual, g_epi and g_weights_extra are all absent -- the
// backward of a step whose weights are inputs, e.g. the benchmark's): pass A, the 3 loads per correspondence that feed it and the
// registers that hold them are compiled out.
template <int IT, bool RAW, bool PGRAD, bool PLAIN = true, int ROWS = 1, bool UP = true>
__device__ __forceinline__ void w8pt16_bwd_pair_impl(const W8BwdArgs& A0, const int pair, double* /*xch*/, W8BwdCoop* co = nullptr,
                                                     const int rowid = 0) {
  W8BwdArgs A = A0;
  if constexpr (!UP) { A.g_res = nullptr; A.g_epi = nullptr; A.g_w_extra = nullptr; }
  static_assert(PLAIN || !PGRAD, "the un-normalised-rows variant has no point gradients");
  static_assert(ROWS == 1 || (ROWS == 16 && IT > 0 && !PGRAD), "cooperative variant: correspondences in registers, weight gradients only");
  constexpr int S = 16 * ROWS;
  const bool norow = PLAIN ? false : (A.variant & DFEPE_W8PT_NO_ROWNORM) != 0;
  const int l = rg_lane();
  const int L = rowid * 16 + l;  // lane within the pair
  const int N = A.N;
  const size_t mp = (size_t)(pair % A.Bm);
  const float* sv = A.save + (size_t)pair * DFEPE_SAVE_FLOATS;

  DFEPE_MARK("B0_load");
  // ---- the forward's record (row-uniform loads) and the pair's correspondences -------------------------------
  const double s1 = sv[S16_T1], c1x = sv[S16_T1 + 1], c1y = sv[S16_T1 + 2];
  const double s2 = sv[S16_T2], c2x = sv[S16_T2 + 1], c2y = sv[S16_T2 + 2];
  double f[9], z[9], td[9], te[8], hb[7], hv[7];
#pragma unroll
  for (int c = 0; c < 9; ++c) { f[c] = sv[S16_F + c]; z[c] = sv[S16_Z + c]; td[c] = reinterpret_cast<const double*>(sv + S16_TD)[c]; }
#pragma unroll
  for (int c = 0; c < 8; ++c) te[c] = reinterpret_cast<const double*>(sv + S16_TE)[c];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    hb[k] = sv[S16_HB + k];
    // unconditional load from a clamped slot, then a select: a load under `?:` compiles to branch + load + wait, seven
    // dependent memory round trips in a row
    const bool own = l > k && l < 9;
    const float hvk = sv[own ? S16_HV + s16_hv_off(k) + (l - k - 1) : S16_SCRATCH];
    hv[k] = own ? (double)hvk : 0.0;
  }
  const double lam = reinterpret_cast<const double*>(sv + S16_LAM)[0];
  // d loss / d F_out and its scale: loaded here (null-safe addresses) with everything else, not where they are first used
  float gF_in[9];
  {
    const float* gp = (A.g_F != nullptr) ? A.g_F + (size_t)pair * 9 : sv + S16_F;
#pragma unroll
    for (int c = 0; c < 9; ++c) gF_in[c] = gp[c];
  }
  const float gs_in = ((A.g_scale != nullptr) ? A.g_scale : sv + S16_TAG)[l & 0];  // a vector load: the scalar cache would miss
  const int twist = (int)sv[S16_TWIST];
  const double inv_tr = sv[S16_INVTR];
  const bool good = sv[S16_TAG] == S16_TAG_VALUE;  // a record of another layout (an older library) would be misread: poison instead

  // same unconditional loads and the same drop rule as the forward (w8pt16_fwd_pair, phase 0); IT = 0: any N, re-read per pass
  constexpr int ITR = (IT > 0) ? IT : 1;
  const int nit = (IT > 0) ? IT : (N + S - 1) / S;
  Pt pt[ITR];
  float wv[ITR];   // the weight (softmax output in logits mode); 0 on padding lanes
  bool kept[ITR];  // false: the forward dropped this correspondence from X
  const float* wsrc = A.wts + (size_t)pair * N;
  constexpr int kWi = RAW ? 4 : 6;  // slot of the weight in a RawRec
  auto point_load = [&](int it) {
    RawRec r;
    const int i = it * S + L;
    load_point_raw<RAW>(A.pts1, A.pts2, mp, N, i, r);
    r.v[kWi] = wsrc[(i < N) ? i : N - 1];
    return r;
  };
  auto point_decode = [&](int it, const RawRec& raw) {
    PRec r;
    const int i = it * S + L;
    decode_point<RAW>(raw, N, i, A.hw_sx, A.hw_sy, r.p, r.valid, r.keep);
    const float wr = raw.v[kWi];
    const bool wfin = fabsf(wr) < 3e38f;
    r.keep = r.keep && wfin;
    r.w = (r.valid && wfin) ? wr : 0.0f;
    r.ws = r.w;
    return r;
  };
  // Upstream per-correspondence gradients: null-safe pointers (an absent tensor reads the weights instead and is masked), so
  // that every load is unconditional and can be issued with the correspondences -- one memory round trip, not one per use.
  const bool has_res = A.g_res != nullptr, has_epi = A.g_epi != nullptr, has_wx = A.g_w_extra != nullptr;
  const float* gres_p = has_res ? A.g_res + (size_t)pair * N : wsrc;
  const float* gepi_p = has_epi ? A.g_epi + (size_t)pair * N : wsrc;
  const float* gwx_p = has_wx ? A.g_w_extra + (size_t)pair * N : wsrc;
  float gres_r[ITR], gepi_r[ITR], gwx_r[ITR];  // IT > 0: in registers from the start
  // (round 6, scripts/ubench/bwd_phases.hip: this block -- ~50 load instructions, no wait inside it -- is 23 % of the g_F-only kernel:
  // a lone wavefront pays ~60 cycles per VMEM instruction it ISSUES while every CU is loading.  Keeping the correspondences raw until
  // pass B, so that the uniform part runs under their arrival, changed nothing: 3059 vs 3064 cycles, same 250 registers -- the data
  // is back long before the rank-2 adjoint needs it; what costs is the instruction issue itself.)
  if constexpr (IT > 0) {
    static_for<0, IT>([&](auto c) {
      constexpr int it = decltype(c)::value;
      const int i = it * S + L;
      const int ic = (i < N) ? i : N - 1;
      const RawRec raw = point_load(it);
      const float a = gres_p[ic], b = gepi_p[ic], cxt = gwx_p[ic];
      const PRec r = point_decode(it, raw);
      pt[it] = r.p; wv[it] = r.w; kept[it] = r.keep;
      gres_r[it] = has_res ? a : 0.0f; gepi_r[it] = has_epi ? b : 0.0f; gwx_r[it] = has_wx ? cxt : 0.0f;
    });
  }
  // upstream gradients of correspondence (it, i): registers (IT > 0) or unconditional loads (IT = 0); 0 when absent
  auto up_res = [&](int it, int i) { if constexpr (IT > 0) return gres_r[it]; else return has_res ? gres_p[(i < N) ? i : N - 1] : 0.0f; };
  auto up_epi = [&](int it, int i) { if constexpr (IT > 0) return gepi_r[it]; else return has_epi ? gepi_p[(i < N) ? i : N - 1] : 0.0f; };
  auto up_wx = [&](int it, int i) { if constexpr (IT > 0) return gwx_r[it]; else return has_wx ? gwx_p[(i < N) ? i : N - 1] : 0.0f; };
  auto cached_load = [&](int it) {
    if constexpr (IT > 0) return RawRec{};
    else return point_load(it);
  };
  auto point = [&](int it, const RawRec& raw) {
    if constexpr (IT > 0) {
      PRec r;
      r.p = pt[it]; r.w = wv[it]; r.ws = wv[it]; r.keep = kept[it];
      r.valid = it * S + L < N;
      return r;
    } else {
      return point_decode(it, raw);
    }
  };

  DFEPE_MARK("B1_passA");
  // ---- pass A: X^T g_r  and  sum_i g_epi_i d(d_i)/d(out) ---------------------------------------------------------
  // partial sums in fp32 (the reference's whole backward is fp32); everything uniform downstream is fp64
  double gx[9], go[9], o[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) { gx[c] = 0.0; go[c] = 0.0; o[c] = 0.0; }
  if (A.g_epi != nullptr) {
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c] = (double)A.F_out[(size_t)pair * 9 + c];
  }
  if (A.g_res != nullptr || A.g_epi != nullptr) {
    float gxf[9], gof[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) { gxf[c] = 0.0f; gof[c] = 0.0f; }
    for_points<IT>(nit, cached_load, point, [&](int it, const PRec& rec) {
      const int i = it * S + L;
      const Pt& p = rec.p;
      const float wf = rec.w;
      const bool keep = rec.keep;
      if (!rec.valid) return;
      if (A.g_res != nullptr) {
        const double w = (double)wf;
        double ra[3], rb[2], inv;
        row_factors(p, s1, c1x, c1y, s2, c2x, c2y, ra, rb, inv);
        if (!PLAIN) inv = norow ? 1.0 : inv;
        const double gw = keep ? (double)up_res(it, i) * w * inv : 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double ga = gw * ra[c];
          gxf[c] += (float)(ga * rb[0]); gxf[3 + c] += (float)(ga * rb[1]); gxf[6 + c] += (float)ga;
        }
      }
      if (A.g_epi != nullptr) {
        const double x1[3] = {p.x1, p.y1, p.z1}, x2[3] = {p.x2, p.y2, p.z2};
        double l1[3], l2[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) l1[c] = x2[0] * o[c] + x2[1] * o[3 + c] + x2[2] * o[6 + c];
#pragma unroll
        for (int r = 0; r < 3; ++r) l2[r] = o[3 * r] * x1[0] + o[3 * r + 1] * x1[1] + o[3 * r + 2] * x1[2];
        const double dd = x1[0] * l1[0] + x1[1] * l1[1] + x1[2] * l1[2];
        const double n1 = sqrt_nr<1>(l1[0] * l1[0] + l1[1] * l1[1]), n2 = sqrt_nr<1>(l2[0] * l2[0] + l2[1] * l2[1]);
        const double i1 = rcp_nr<1, false>(n1 + 1e-6), i2 = rcp_nr<1, false>(n2 + 1e-6);
        const double S = i1 + i2, ad = fabs(dd);
        const double d = ad * S;
        // clamp(max=) passes the gradient up to and including the bound
        const double g = (d <= (double)A.clamp_at) ? (double)up_epi(it, i) : 0.0;
        const double sg = (dd > 0.0) ? 1.0 : ((dd < 0.0) ? -1.0 : 0.0);
        const double k1 = (n1 > 0.0) ? ad * i1 * i1 * rcp_nr<1, false>(n1) : 0.0;
        const double k2 = (n2 > 0.0) ? ad * i2 * i2 * rcp_nr<1, false>(n2) : 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            double t = sg * S * x2[r] * x1[c];
            if (c < 2) t -= k1 * l1[c] * x2[r];
            if (r < 2) t -= k2 * l2[r] * x1[c];
            gof[3 * r + c] += (float)(g * t);
          }
      }
    });
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      if (A.g_res != nullptr) gxf[c] = rg_sum(gxf[c]);
      if (A.g_epi != nullptr) gof[c] = rg_sum(gof[c]);
    }
    if constexpr (ROWS > 1) {  // the 16 rows' sums through LDS: one barrier for all 18 values
      if (l == 0) {
#pragma unroll
        for (int c = 0; c < 9; ++c) { co->red[c][rowid] = gxf[c]; co->red[9 + c][rowid] = gof[c]; }
      }
      DFEPE_BLOCK_SYNC();
#pragma unroll
      for (int c = 0; c < 9; ++c) { gxf[c] = rg_sum(co->red[c][l]); gof[c] = rg_sum(co->red[9 + c][l]); }
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      if (A.g_res != nullptr) gx[c] = (double)gxf[c];
      if (A.g_epi != nullptr) go[c] = (double)gof[c];
    }
  }
  if (A.g_F != nullptr) {
#pragma unroll
    for (int c = 0; c < 9; ++c) go[c] += (double)gF_in[c];
  }
  const double gsc = (A.g_scale != nullptr) ? (double)gs_in : 1.0;  // everything downstream is linear in the three gradients
  if (A.g_scale != nullptr) {
#pragma unroll
    for (int c = 0; c < 9; ++c) { go[c] *= gsc; gx[c] *= gsc; }
  }

  DFEPE_MARK("B2_uniform");
  double u[9];
  double u3[3], v3[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { u3[c] = sv[S16_U3 + c]; v3[c] = sv[S16_V3 + c]; }
  const double s3 = sv[S16_S3];
  if (ROWS == 1 || rowid == 0) {
  // ---- uniform part (one row per pair) ------------------------------------------------------------------------
  // g_F' = T2 g_out T1^T with T = [[s,0,-s cx],[0,s,-s cy],[0,0,1]] written out
  double tmp[9], G[9];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    tmp[c] = s2 * (go[c] - c2x * go[6 + c]);
    tmp[3 + c] = s2 * (go[3 + c] - c2y * go[6 + c]);
    tmp[6 + c] = go[6 + c];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    G[3 * r] = s1 * (tmp[3 * r] - c1x * tmp[3 * r + 2]);
    G[3 * r + 1] = s1 * (tmp[3 * r + 1] - c1y * tmp[3 * r + 2]);
    G[3 * r + 2] = tmp[3 * r + 2];
  }
  DFEPE_MARK("B3_rank2");
  // rank-2 projection adjoint with nothing but the dropped triplet (s3, u3, v3) and F = reshape(f).  With the
  // pseudo-inverses A_u = (s3^2 I - F F^T)^+ (null vector u3) and A_v = (s3^2 I - F^T F)^+ (null vector v3), the first-order
  // perturbation of the triplet is  d s3 = u3^T dF v3,  d u3 = A_u (s3 dF v3 + F dF^T u3),  d v3 = A_v (s3 dF^T u3 + F^T dF v3),
  // so with p = A_u G v3, q = A_v G^T u3, a33 = u3^T G v3:
  //   g_F = G - (a33 u3 + s3^2 p + s3 F q) v3^T - u3 (s3 F^T p + s3^2 q)^T
  // (the same five-term expression the full-SVD form sum_k coef_k (...) u_k / v_k collapses to).
  double gf[9];
  {
    const double* Fm = f;
    double Gv3[3], Gtu3[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      Gv3[r] = G[3 * r] * v3[0] + G[3 * r + 1] * v3[1] + G[3 * r + 2] * v3[2];
      Gtu3[r] = G[r] * u3[0] + G[3 + r] * u3[1] + G[6 + r] * u3[2];
    }
    const double a33 = u3[0] * Gv3[0] + u3[1] * Gv3[1] + u3[2] * Gv3[2];
    const double q2 = s3 * s3;
    // F F^T and F^T F (6 distinct entries each)
    const double d00 = Fm[0] * Fm[0] + Fm[1] * Fm[1] + Fm[2] * Fm[2], d01 = Fm[0] * Fm[3] + Fm[1] * Fm[4] + Fm[2] * Fm[5];
    const double d02 = Fm[0] * Fm[6] + Fm[1] * Fm[7] + Fm[2] * Fm[8], d11 = Fm[3] * Fm[3] + Fm[4] * Fm[4] + Fm[5] * Fm[5];
    const double d12 = Fm[3] * Fm[6] + Fm[4] * Fm[7] + Fm[5] * Fm[8], d22 = Fm[6] * Fm[6] + Fm[7] * Fm[7] + Fm[8] * Fm[8];
    const double b00 = Fm[0] * Fm[0] + Fm[3] * Fm[3] + Fm[6] * Fm[6], b01 = Fm[0] * Fm[1] + Fm[3] * Fm[4] + Fm[6] * Fm[7];
    const double b02 = Fm[0] * Fm[2] + Fm[3] * Fm[5] + Fm[6] * Fm[8], b11 = Fm[1] * Fm[1] + Fm[4] * Fm[4] + Fm[7] * Fm[7];
    const double b12 = Fm[1] * Fm[2] + Fm[4] * Fm[5] + Fm[7] * Fm[8], b22 = Fm[2] * Fm[2] + Fm[5] * Fm[5] + Fm[8] * Fm[8];
    double pt3[3], qt3[3];
    sym3_pinv_apply(q2 - d00, -d01, -d02, q2 - d11, -d12, q2 - d22, u3, Gv3, pt3);
    sym3_pinv_apply(q2 - b00, -b01, -b02, q2 - b11, -b12, q2 - b22, v3, Gtu3, qt3);
    double pv[3], qv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double Fq = Fm[3 * r] * qt3[0] + Fm[3 * r + 1] * qt3[1] + Fm[3 * r + 2] * qt3[2];   // (F q)_r
      const double Ftp = Fm[r] * pt3[0] + Fm[3 + r] * pt3[1] + Fm[6 + r] * pt3[2];               // (F^T p)_r
      pv[r] = a33 * u3[r] + q2 * pt3[r] + s3 * Fq;
      qv[r] = s3 * Ftp + q2 * qt3[r];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) gf[3 * r + c] = G[3 * r + c] - pv[r] * v3[c] - u3[r] * qv[c] + gx[3 * r + c];
  }
  DFEPE_MARK("B4_eigadj");
  // eigenvector adjoint: u = -(1/trace) H (T - lam I)^+ H^T g_f.  H^T = H_6 ... H_0 (each H_k symmetric).
  {
    double gt[9], y[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) gt[c] = gf[c];
    static_for<0, 7>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const double s = -hb[k] * rg_dot_bcast<k + 1>(hv[k], gt);  // fused broadcast-FMA chains (rowgroup.h)
      rg_axpy_bcast<k + 1>(gt, hv[k], s);
    });
  DFEPE_MARK("B5_tripinv");
    tri_pinv_apply(td, te, lam, z, twist, gt, y);
    static_for<0, 7>([&](auto kc) {
      constexpr int k = 6 - decltype(kc)::value;
      const double s = -hb[k] * rg_dot_bcast<k + 1>(hv[k], y);
      rg_axpy_bcast<k + 1>(y, hv[k], s);
    });
    const double sc = good ? -inv_tr : (double)NAN;
#pragma unroll
    for (int c = 0; c < 9; ++c) u[c] = sc * y[c];
  }
  if constexpr (ROWS > 1) {
    if (l < 9) co->u[l] = u[l];
  }
  }  // solver row
  if constexpr (ROWS > 1) {
    DFEPE_BLOCK_SYNC();
#pragma unroll
    for (int c = 0; c < 9; ++c) u[c] = co->u[c];
  }

  DFEPE_MARK("B6_passB");
  // ---- pass B: g_w ---------------------------------------------------------------------------------------------
  float* dst = A.g_w + (size_t)pair * N;
  float gwv[ITR];
  float wg = 0.0f;
  for_points<IT>(nit, cached_load, point, [&](int it, const PRec& rec) {
    const int i = it * S + L;
    const Pt& p = rec.p;
    const float wf = rec.w;
    const bool valid = rec.valid, keep = rec.keep;
    double ra[3], rb[2], inv;
    row_factors(p, s1, c1x, c1y, s2, c2x, c2y, ra, rb, inv);
    if (!PLAIN) inv = norow ? 1.0 : inv;
    const double w = (double)wf;
    const double a = row_bilinear(ra, rb, f) * inv, b = row_bilinear(ra, rb, u) * inv;  // p^ . f, p^ . u
    const double gr = valid ? gsc * (double)up_res(it, i) : 0.0;
    float gwi = keep ? (float)(2.0 * w * a * b + gr * a) : 0.0f;
    gwi += valid ? up_wx(it, i) : 0.0f;
    gwi = valid ? gwi : 0.0f;
    wg = fmaf(gwi, wf, wg);
    if constexpr (IT > 0) gwv[it] = gwi;
    else if (valid) dst[i] = gwi;  // provisional when the softmax adjoint follows: the same lane re-reads it below
  });
  // softmax adjoint g_logit_i = w_i (g_w_i - sum_j w_j g_w_j)
  float sdot = A.logits_mode ? rg_sum(wg) : 0.0f;
  if constexpr (ROWS > 1) {
    if (l == 0) co->red[18][rowid] = sdot;
    DFEPE_BLOCK_SYNC();
    sdot = rg_sum(co->red[18][l]);
  }
  if constexpr (IT > 0) {
    static_for<0, IT>([&](auto c) {
      constexpr int it = decltype(c)::value;
      const int i = it * S + L;
      if (i < N) dst[i] = A.logits_mode ? wv[it] * (gwv[it] - sdot) : gwv[it];
    });
  } else if (A.logits_mode) {
    // (weight, provisional gradient) of correspondence `it`
    auto parked_load = [&](int it) {
      RawRec r;
      const int i = it * S + L;
      r.v[0] = wsrc[(i < N) ? i : N - 1];
      r.v[1] = dst[(i < N) ? i : N - 1];
      return r;
    };
    auto parked = [&](int it, const RawRec& raw) {
      PRec r;
      r.valid = it * S + L < N;
      r.w = raw.v[0];
      r.p.x1 = raw.v[1];
      return r;
    };
    for_points<0>(nit, parked_load, parked, [&](int it, const PRec& r) {
      if (r.valid) dst[it * S + L] = r.w * (r.p.x1 - sdot);
    });
  }

  if constexpr (PGRAD) {
    // ---- adjoint w.r.t. the point coordinates (derivation checked against autograd in scripts/proto_pts_grad.py) ----
    // rows -> (a, b) -> points, plus the dependence of the Hartley transforms (centroid c, scale s = k / mean distance)
    // on the points through the rows and through out = T2^T F' T1, plus the direct dependence of the epipolar residual.
    const double kH = 1.4142;
    const double t1[9] = {s1, 0.0, -s1 * c1x, 0.0, s1, -s1 * c1y, 0.0, 0.0, 1.0};
    const double t2[9] = {s2, 0.0, -s2 * c2x, 0.0, s2, -s2 * c2y, 0.0, 0.0, 1.0};
    double Fp[9], tA[9], gT1[9], gT2[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Fp[3 * r + c] = fma(-s3 * u3[r], v3[c], f[3 * r + c]);
    mat3_mul_tn(t2, Fp, tA);      // T2^T F'
    mat3_mul_tn(tA, go, gT1);     // d<G, T2^T F' T1>/dT1 = (T2^T F')^T G
    mat3_mul(Fp, t1, tA);         // F' T1
    mat3_mul_nt(tA, go, gT2);     // d/dT2 = F' T1 G^T
    float sums[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) sums[k] = 0.0f;
    float q1x[ITR], q1y[ITR], q2x[ITR], q2y[ITR];
    for_points<IT>(nit, cached_load, point, [&](int it, const PRec& rec) {
      const int i = it * S + L;
      const Pt& p = rec.p;
      const float wf = rec.w;
      const bool keep = rec.keep;
      if constexpr (IT > 0) q1x[it] = q1y[it] = q2x[it] = q2y[it] = 0.0f;
      if (!rec.valid) return;
      const double w = (double)wf;
      const double z1 = p.z1, z2 = p.z2;
      const double a[3] = {s1 * ((double)p.x1 - c1x * z1), s1 * ((double)p.y1 - c1y * z1), z1};
      const double b0 = s2 * ((double)p.x2 - c2x * z2), b1 = s2 * ((double)p.y2 - c2y * z2);
      const double n2 = (a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * (b0 * b0 + b1 * b1 + 1.0);
      const bool ok = keep && (n2 > 1e-24);
      const double inv = ok ? rsqrt_nr<1, false>(n2) : 0.0;
      double ph[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) { ph[k] = b0 * a[k] * inv; ph[3 + k] = b1 * a[k] * inv; ph[6 + k] = a[k] * inv; }
      double af = 0.0, bu = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) { af += ph[k] * f[k]; bu += ph[k] * u[k]; }
      const double gr = gsc * (double)up_res(it, i);
      const double cu = w * w * af, cf = w * w * bu + w * gr, dotp = 2.0 * w * w * af * bu + w * gr * af;
      double gp[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) gp[k] = (cu * u[k] + cf * f[k] - ph[k] * dotp) * inv;
      const double ga0 = b0 * gp[0] + b1 * gp[3] + gp[6], ga1 = b0 * gp[1] + b1 * gp[4] + gp[7], ga2 = b0 * gp[2] + b1 * gp[5] + gp[8];
      const double gb0 = a[0] * gp[0] + a[1] * gp[1] + a[2] * gp[2], gb1 = a[0] * gp[3] + a[1] * gp[4] + a[2] * gp[5];
      double e1[3] = {0.0, 0.0, 0.0}, e2[3] = {0.0, 0.0, 0.0};
      if (A.g_epi != nullptr) {  // direct dependence of d_i on x1_i, x2_i
        const double x1[3] = {p.x1, p.y1, p.z1}, x2[3] = {p.x2, p.y2, p.z2};
        double l1[3], l2[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) l1[c] = x2[0] * o[c] + x2[1] * o[3 + c] + x2[2] * o[6 + c];
#pragma unroll
        for (int r = 0; r < 3; ++r) l2[r] = o[3 * r] * x1[0] + o[3 * r + 1] * x1[1] + o[3 * r + 2] * x1[2];
        const double dd = x1[0] * l1[0] + x1[1] * l1[1] + x1[2] * l1[2];
        const double n1 = sqrt_nr<1>(l1[0] * l1[0] + l1[1] * l1[1]), nn2 = sqrt_nr<1>(l2[0] * l2[0] + l2[1] * l2[1]);
        const double i1 = rcp_nr<1, false>(n1 + 1e-6), i2 = rcp_nr<1, false>(nn2 + 1e-6);
        const double Ss = i1 + i2, ad = fabs(dd);
        const double g = (ad * Ss <= (double)A.clamp_at) ? gsc * (double)up_epi(it, i) : 0.0;
        const double sg = (dd > 0.0) ? 1.0 : ((dd < 0.0) ? -1.0 : 0.0);
        const double k1 = (n1 > 0.0) ? ad * i1 * i1 * rcp_nr<1, false>(n1) : 0.0;
        const double k2 = (nn2 > 0.0) ? ad * i2 * i2 * rcp_nr<1, false>(nn2) : 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          e1[c] = g * (sg * Ss * l1[c] - k2 * (l2[0] * o[c] + l2[1] * o[3 + c]));          // d n2 / d x1_c
          e2[c] = g * (sg * Ss * l2[c] - k1 * (l1[0] * o[3 * c] + l1[1] * o[3 * c + 1]));  // d n1 / d x2_c
        }
      }
      // provisional values; the Hartley terms are added below once their sums over the pair are known
      const float p1x = (float)(s1 * ga0 + e1[0]), p1y = (float)(s1 * ga1 + e1[1]);
      const float p2x = (float)(s2 * gb0 + e2[0]), p2y = (float)(s2 * gb1 + e2[1]);
      if constexpr (IT > 0) {
        q1x[it] = p1x; q1y[it] = p1y; q2x[it] = p2x; q2y[it] = p2y;
      } else if (RAW) {  // parked in the output: the same lane finishes them in the second pass
        float4 q; q.x = p1x; q.y = p1y; q.z = p2x; q.w = p2y;
        reinterpret_cast<float4*>(A.g_p1)[(size_t)pair * N + i] = q;
      } else {
        float* d1 = A.g_p1 + ((size_t)pair * N + i) * 3;
        float* d2 = A.g_p2 + ((size_t)pair * N + i) * 3;
        d1[0] = p1x; d1[1] = p1y; d2[0] = p2x; d2[1] = p2y;
      }
      if (!RAW) {
        A.g_p1[((size_t)pair * N + i) * 3 + 2] = (float)(ga2 - s1 * (c1x * ga0 + c1y * ga1) + e1[2]);
        A.g_p2[((size_t)pair * N + i) * 3 + 2] = (float)(-s2 * (c2x * gb0 + c2y * gb1) + e2[2]);
      }
      const double dx1 = (double)p.x1 - c1x, dy1 = (double)p.y1 - c1y, dx2 = (double)p.x2 - c2x, dy2 = (double)p.y2 - c2y;
      const double r1 = dx1 * dx1 + dy1 * dy1, r2 = dx2 * dx2 + dy2 * dy2;
      const double ir1 = (r1 > 0.0) ? rsqrt_nr<1, false>(r1) : 0.0, ir2 = (r2 > 0.0) ? rsqrt_nr<1, false>(r2) : 0.0;
      sums[0] += (float)(((double)p.x1 - c1x * z1) * ga0 + ((double)p.y1 - c1y * z1) * ga1);  // d/ds1 through the rows
      sums[1] += (float)(-s1 * z1 * ga0);
      sums[2] += (float)(-s1 * z1 * ga1);
      sums[3] += (float)(((double)p.x2 - c2x * z2) * gb0 + ((double)p.y2 - c2y * z2) * gb1);
      sums[4] += (float)(-s2 * z2 * gb0);
      sums[5] += (float)(-s2 * z2 * gb1);
      sums[6] += (float)(dx1 * ir1); sums[7] += (float)(dy1 * ir1);
      sums[8] += (float)(dx2 * ir2); sums[9] += (float)(dy2 * ir2);
    });
    double tot[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) tot[k] = (double)rg_sum(sums[k]);
    const double invN = 1.0 / (double)N;
    const double Gs1 = tot[0] + gT1[0] + gT1[4] - c1x * gT1[2] - c1y * gT1[5];
    const double Gs2 = tot[3] + gT2[0] + gT2[4] - c2x * gT2[2] - c2y * gT2[5];
    const double Gd1 = -Gs1 * s1 * s1 / kH, Gd2 = -Gs2 * s2 * s2 / kH;   // s = k / dbar
    const double Gc1x = (tot[1] - s1 * gT1[2] - Gd1 * invN * tot[6]) * invN, Gc1y = (tot[2] - s1 * gT1[5] - Gd1 * invN * tot[7]) * invN;
    const double Gc2x = (tot[4] - s2 * gT2[2] - Gd2 * invN * tot[8]) * invN, Gc2y = (tot[5] - s2 * gT2[5] - Gd2 * invN * tot[9]) * invN;
    for_points<IT>(nit, cached_load, point, [&](int it, const PRec& rec) {
      const int i = it * S + L;
      const Pt& p = rec.p;
      if (!rec.valid) return;
      const double dx1 = (double)p.x1 - c1x, dy1 = (double)p.y1 - c1y, dx2 = (double)p.x2 - c2x, dy2 = (double)p.y2 - c2y;
      const double r1 = dx1 * dx1 + dy1 * dy1, r2 = dx2 * dx2 + dy2 * dy2;
      const double ir1 = (r1 > 0.0) ? rsqrt_nr<1, false>(r1) : 0.0, ir2 = (r2 > 0.0) ? rsqrt_nr<1, false>(r2) : 0.0;
      const float a1x = (float)(Gd1 * invN * dx1 * ir1 + Gc1x), a1y = (float)(Gd1 * invN * dy1 * ir1 + Gc1y);
      const float a2x = (float)(Gd2 * invN * dx2 * ir2 + Gc2x), a2y = (float)(Gd2 * invN * dy2 * ir2 + Gc2y);
      float p1x, p1y, p2x, p2y;  // the provisional values of the first pass
      if constexpr (IT > 0) {
        p1x = q1x[it]; p1y = q1y[it]; p2x = q2x[it]; p2y = q2y[it];
      } else if (RAW) {
        const float4 q = reinterpret_cast<const float4*>(A.g_p1)[(size_t)pair * N + i];
        p1x = q.x; p1y = q.y; p2x = q.z; p2y = q.w;
      } else {
        p1x = A.g_p1[((size_t)pair * N + i) * 3]; p1y = A.g_p1[((size_t)pair * N + i) * 3 + 1];
        p2x = A.g_p2[((size_t)pair * N + i) * 3]; p2y = A.g_p2[((size_t)pair * N + i) * 3 + 1];
      }
      if (RAW) {
        float4 q;  // chain through x^ = 2x/W - 1
        q.x = (p1x + a1x) * A.hw_sx; q.y = (p1y + a1y) * A.hw_sy; q.z = (p2x + a2x) * A.hw_sx; q.w = (p2y + a2y) * A.hw_sy;
        reinterpret_cast<float4*>(A.g_p1)[(size_t)pair * N + i] = q;
      } else {
        float* d1 = A.g_p1 + ((size_t)pair * N + i) * 3;
        float* d2 = A.g_p2 + ((size_t)pair * N + i) * 3;
        d1[0] = p1x + a1x; d1[1] = p1y + a1y; d2[0] = p2x + a2x; d2[1] = p2y + a2y;
      }
    });
  }
  DFEPE_MARK("B7_end");
}

template <int IT, bool RAW>
__device__ __forceinline__ void w8pt16_bwd_pair(const W8BwdArgs& A, const int pair, double* xch) {
  if (A.g_p1 != nullptr) w8pt16_bwd_pair_impl<IT, RAW, true>(A, pair, xch);
  else w8pt16_bwd_pair_impl<IT, RAW, false>(A, pair, xch);
}
