// w8pt_bwd — analytic adjoint of w8pt_fwd with respect to the weights.  One wavefront per image pair.
//
// Replaces torch.autograd's replay of the per-sample torch.svd calls of Fit.weighted_svd
// (deepFEPE/models/DeepFNet.py:232-256) with closed forms:
//   epipolar residual  d_i(out)            -> g_out      (only when g_epi is given)
//   out = T2^T F' T1                        -> g_F' = T2 g_out T1^T
//   F' = F - s3 u3 v3^T  (rank-2 projection) -> g_F    (3x3 SVD adjoint restricted to the dropped triplet)
//   F = reshape(f), r = X f                 -> g_f = vec(g_F) + X^T g_r
//   f = eigenvector of X^T X                -> u = sum_k q_k (q_k . g_f) / (lam_f - lam_k)
//   X_i = w_i ph_i                          -> g_w_i = 2 w_i (ph_i.f)(ph_i.u) + g_r_i (ph_i.f)
// All nine eigenpairs come from the forward Jacobi (the `save` record), so the cost is O(9 N) per pair:
// two streaming passes over the correspondences (the second one hits L2) and a few hundred uniform flops.
#include "dfepe_common.h"
#include "w8pt16_bwd_body.h"  // W8BwdArgs

int dfepe_w8pt16_bwd_launch(const W8BwdArgs& A, bool raw, hipStream_t st);  // w8pt16.hip

namespace {

__device__ __forceinline__ double guard_den(double d) {
  // keep the sign, floor the magnitude: repeated eigen/singular values give a large-but-finite gradient, never NaN
  const double lim = 1e-30;
  return (fabs(d) < lim) ? ((d < 0.0) ? -lim : lim) : d;
}

// COOP: one 256-thread workgroup per pair instead of one wavefront (large N with a batch too small to fill the GPU, like
// w8pt_fwd's cooperative variant): the four wavefronts split the passes over the correspondences and each repeats the
// few hundred wave-uniform instructions; their partial sums meet in LDS.  Not combined with the point gradients.
template <bool RAW, bool PGRAD, bool COOP>
__global__ void __launch_bounds__(256, (PGRAD ? 2 : 4))  // the point-gradient variant trades occupancy for no spills
w8pt_bwd_kernel(const float* __restrict__ pts1, const float* __restrict__ pts2, const float* __restrict__ wts, int B,
                int Bm, int N, float hw_sx, float hw_sy, float clamp_at, const float* __restrict__ save,
                const float* __restrict__ F_out, const float* __restrict__ g_F, const float* __restrict__ g_res,
                const float* __restrict__ g_epi, const float* __restrict__ g_w_extra, const float* __restrict__ g_scale, int logits_mode,
                float* __restrict__ g_w, float* __restrict__ g_p1, float* __restrict__ g_p2) {
  static_assert(!(COOP && PGRAD), "the cooperative variant does not produce point gradients");
  __shared__ float red[4][20];  // COOP only: per-wavefront partial sums
  __shared__ float qlds[4][96];  // eigenvectors (81) and eigenvalues (9) of the pair, staged for the lane-parallel eigen adjoint
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform by construction
  const size_t pair = COOP ? (size_t)blockIdx.x : (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
  if (!COOP && pair >= (size_t)B) return;
  constexpr int NT = COOP ? 256 : WAVE;
  const int tid = COOP ? (int)threadIdx.x : lane;

  const size_t mp = pair % (size_t)Bm;  // correspondences may be shared by several weight sets
  {  // stage Q and Lambda of the save record in LDS (two coalesced loads; their latency hides behind the scalar loads below)
    const float* svq = save + pair * DFEPE_SAVE_FLOATS;
    float* ql = qlds[COOP ? 0 : wave];
    if (!COOP || wave == 0) {
      ql[lane] = svq[SV_Q + lane];
      if (lane < 17) ql[64 + lane] = svq[SV_Q + 64 + lane];
      if (lane < 9) ql[81 + lane] = svq[SV_LAM + lane];
    }
    if (COOP) __syncthreads();
  }
  const float* sv = save + pair * DFEPE_SAVE_FLOATS;
  // wave-uniform values are parked in scalar registers (to_sgpr) to keep the VGPR budget at 4 waves/SIMD
  const double s1 = to_sgpr((double)sv[SV_T1]), c1x = to_sgpr((double)sv[SV_T1 + 1]), c1y = to_sgpr((double)sv[SV_T1 + 2]);
  const double s2 = to_sgpr((double)sv[SV_T2]), c2x = to_sgpr((double)sv[SV_T2 + 1]), c2y = to_sgpr((double)sv[SV_T2 + 2]);
  const int ksel = (int)sv[SV_KMIN];
  const double sgn = sv[SV_SIGN];
  double f[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) f[c] = to_sgpr(sgn * (double)sv[SV_Q + ksel * 9 + c]);

  // ---- pass A: X^T g_r  and  sum_i g_epi_i d(d_i)/d(out) ---------------------------------------------
  // partial sums in fp32 (the reference's whole backward is fp32); everything uniform downstream is fp64
  float gxf[9], gof[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) { gxf[c] = 0.0f; gof[c] = 0.0f; }
  double o[9];
  if (g_epi != nullptr) {
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c] = to_sgpr((double)F_out[pair * 9 + c]);
  }
  const float* wsrc = wts + pair * N;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
  for (int i = tid; i < N; i += NT) {
    if (g_res == nullptr && g_epi == nullptr) break;  // nothing to accumulate (fused training step: the loss depends on F only)
    const Pt p = global_point<RAW>(pts1, pts2, mp, i, N, hw_sx, hw_sy);
    if (g_res != nullptr) {
      double ph[9];
      const double w = (double)wsrc[i];
      const bool ok = unit_row(p, s1, c1x, c1y, s2, c2x, c2y, ph) && (fabs(w) < 1e150);
      const double gw = ok ? (double)g_res[pair * N + i] * w : 0.0;
#pragma unroll
      for (int c = 0; c < 9; ++c) gxf[c] += (float)(gw * ph[c]);
    }
    if (g_epi != nullptr) {
      const double x1[3] = {p.x1, p.y1, p.z1}, x2[3] = {p.x2, p.y2, p.z2};
      double l1[3], l2[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) l1[c] = x2[0] * o[c] + x2[1] * o[3 + c] + x2[2] * o[6 + c];
#pragma unroll
      for (int r = 0; r < 3; ++r) l2[r] = o[3 * r] * x1[0] + o[3 * r + 1] * x1[1] + o[3 * r + 2] * x1[2];
      const double dd = x1[0] * l1[0] + x1[1] * l1[1] + x1[2] * l1[2];
      const double n1 = fast_sqrt(l1[0] * l1[0] + l1[1] * l1[1]), n2 = fast_sqrt(l2[0] * l2[0] + l2[1] * l2[1]);
      const double i1 = fast_rcp(n1 + 1e-6), i2 = fast_rcp(n2 + 1e-6);
      const double S = i1 + i2, ad = fabs(dd);
      const double d = ad * S;
      const double g = (d <= (double)clamp_at) ? (double)g_epi[pair * N + i] : 0.0;  // clamp(max=) passes the gradient up to and including the bound
      const double sg = (dd > 0.0) ? 1.0 : ((dd < 0.0) ? -1.0 : 0.0);
      const double k1 = (n1 > 0.0) ? ad * i1 * i1 * fast_rcp(n1) : 0.0;
      const double k2 = (n2 > 0.0) ? ad * i2 * i2 * fast_rcp(n2) : 0.0;
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
  }
  double gx[9], go[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    gx[c] = (g_res != nullptr) ? (double)wave_sum(gxf[c]) : 0.0;
    go[c] = (g_epi != nullptr) ? (double)wave_sum(gof[c]) : 0.0;
  }
  if (COOP && (g_res != nullptr || g_epi != nullptr)) {
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 9; ++c) { red[wave][c] = (float)gx[c]; red[wave][9 + c] = (float)go[c]; }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      gx[c] = (double)((red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
      go[c] = (double)((red[0][9 + c] + red[1][9 + c]) + (red[2][9 + c] + red[3][9 + c]));
    }
  }
  if (g_F != nullptr) {
#pragma unroll
    for (int c = 0; c < 9; ++c) go[c] += (double)g_F[pair * 9 + c];
  }
  const double gsc = (g_scale != nullptr) ? (double)g_scale[0] : 1.0;  // scales all three upstream gradients (linear downstream)
  if (g_scale != nullptr) {
#pragma unroll
    for (int c = 0; c < 9; ++c) { go[c] *= gsc; gx[c] *= gsc; }
  }

  // ---- uniform part ---------------------------------------------------------------------------------
  // g_F' = T2 g_out T1^T
  double t1[9] = {s1, 0.0, -s1 * c1x, 0.0, s1, -s1 * c1y, 0.0, 0.0, 1.0};
  double t2[9] = {s2, 0.0, -s2 * c2x, 0.0, s2, -s2 * c2y, 0.0, 0.0, 1.0};
  double tmp[9], G[9];
  // T = [[s,0,-s cx],[0,s,-s cy],[0,0,1]] written out (a generic 3x3 product would multiply by its zeros: no fast-math)
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
  // rank-2 projection adjoint
  double U[9], V[9], S[3];
#pragma unroll
  for (int c = 0; c < 9; ++c) { U[c] = sv[SV_U3 + c]; V[c] = sv[SV_V3 + c]; }
  S[0] = sv[SV_S3]; S[1] = sv[SV_S3 + 1]; S[2] = sv[SV_S3 + 2];
  // Only five entries of U^T G V enter: a_k3 = u_k^T G v_3 (k = 0,1,2) and a_3k = u_3^T G v_k (k = 0,1).  With
  // beta_k = coef_k (a_k3 S_3 + a_3k S_k), gamma_k = coef_k (a_k3 S_k + a_3k S_3), coef_k = S_3 / (S_3^2 - S_k^2):
  //   g_F = G - (a_33 u_3 + beta_0 u_0 + beta_1 u_1) v_3^T - u_3 (gamma_0 v_0 + gamma_1 v_1)^T
  double Gv3[3], Gtu3[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    Gv3[r] = G[3 * r] * V[2] + G[3 * r + 1] * V[5] + G[3 * r + 2] * V[8];
    Gtu3[r] = G[r] * U[2] + G[3 + r] * U[5] + G[6 + r] * U[8];
  }
  const double a33 = U[2] * Gv3[0] + U[5] * Gv3[1] + U[8] * Gv3[2];
  double gFm[9];
  {
    double pv[3], qv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { pv[r] = a33 * U[3 * r + 2]; qv[r] = 0.0; }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double coef = S[2] * fast_rcp(guard_den(S[2] * S[2] - S[k] * S[k]));
      const double ak3 = U[k] * Gv3[0] + U[3 + k] * Gv3[1] + U[6 + k] * Gv3[2];      // u_k^T G v_3
      const double a3k = Gtu3[0] * V[k] + Gtu3[1] * V[3 + k] + Gtu3[2] * V[6 + k];  // u_3^T G v_k
      const double beta = coef * (ak3 * S[2] + a3k * S[k]), gamma = coef * (ak3 * S[k] + a3k * S[2]);
#pragma unroll
      for (int r = 0; r < 3; ++r) { pv[r] += beta * U[3 * r + k]; qv[r] += gamma * V[3 * r + k]; }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) gFm[3 * r + c] = G[3 * r + c] - pv[r] * V[3 * c + 2] - U[3 * r + 2] * qv[c];
  }
  // g_f and the eigenvector adjoint
  double gf[9], u[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) { gf[c] = gFm[c] + gx[c]; u[c] = 0.0; }
  const double lsel = sv[SV_LAM + ksel];
  // lane k: coefficient c_k = (q_k . g_f) / (lam_f - lam_k); lane c: u_c = sum_k c_k q_k[c]; the nine u_c then go back to
  // scalar registers.  ~85 vector instructions instead of eight uniform (dot, reciprocal, axpy) rounds of ~35.
  wave_sync();
  const float* ql = qlds[COOP ? 0 : wave];
  double ck = 0.0, uc = 0.0;
  if (lane < 9) {
    double dot = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) dot += (double)ql[lane * 9 + c] * gf[c];
    ck = (lane == ksel) ? 0.0 : dot * fast_rcp(guard_den(lsel - (double)ql[81 + lane]));
  }
  {
    union { double d; int i[2]; } a, b;
    a.d = ck;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      b.i[0] = __builtin_amdgcn_readlane(a.i[0], k);
      b.i[1] = __builtin_amdgcn_readlane(a.i[1], k);
      if (lane < 9) uc += b.d * (double)ql[k * 9 + lane];
    }
    a.d = uc;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      b.i[0] = __builtin_amdgcn_readlane(a.i[0], c);
      b.i[1] = __builtin_amdgcn_readlane(a.i[1], c);
      u[c] = b.d;
    }
  }

  // ---- pass B: g_w --------------------------------------------------------------------------------------
  float* dst = g_w + pair * N;
  float wg = 0.0f;
  auto weight_grad = [&](int i, float& wf) -> float {
    const Pt p = global_point<RAW>(pts1, pts2, mp, i, N, hw_sx, hw_sy);
    double ph[9];
    wf = wsrc[i];
    const double w = (double)wf;
    const bool ok = unit_row(p, s1, c1x, c1y, s2, c2x, c2y, ph) && (fabs(w) < 1e150);
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) { a += ph[c] * f[c]; b += ph[c] * u[c]; }
    const double gr = (g_res != nullptr) ? gsc * (double)g_res[pair * N + i] : 0.0;
    float gwi = ok ? (float)(2.0 * w * a * b + gr * a) : 0.0f;
    if (g_w_extra != nullptr) gwi += g_w_extra[pair * N + i];
    return gwi;
  };
  if (!COOP && logits_mode && N <= 2 * WAVE) {
    // softmax adjoint g_logit_i = w_i (g_w_i - sum_j w_j g_w_j) with the (at most two) gradients of a lane kept in
    // registers: one store per correspondence instead of store, wave sum, load, store
    float g0 = 0.0f, g1 = 0.0f, w0 = 0.0f, w1 = 0.0f;
    const int i1 = lane + WAVE;
    if (lane < N) g0 = weight_grad(lane, w0);
    if (i1 < N) g1 = weight_grad(i1, w1);
    const float s = wave_sum(fmaf(g1, w1, g0 * w0));
    if (lane < N) dst[lane] = w0 * (g0 - s);
    if (i1 < N) dst[i1] = w1 * (g1 - s);
  } else {
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int i = tid; i < N; i += NT) {
      float wf;
      const float gwi = weight_grad(i, wf);
      dst[i] = gwi;
      wg += gwi * wf;
    }
    if (logits_mode) {
      // dst is re-read by the lane that wrote it
      float s = wave_sum(wg);
      if (COOP) {
        if (lane == 0) red[wave][18] = s;
        __syncthreads();
        s = (red[0][18] + red[1][18]) + (red[2][18] + red[3][18]);
      }
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
      for (int i = tid; i < N; i += NT) dst[i] = wsrc[i] * (dst[i] - s);
    }
  }

  if (PGRAD) {
    // ---- adjoint w.r.t. the point coordinates (derivation checked against autograd in scripts/proto_pts_grad.py) ----
    // rows -> (a, b) -> points, plus the dependence of the Hartley transforms (centroid c, scale s = k / mean distance)
    // on the points through the rows and through out = T2^T F' T1, plus the direct dependence of the epipolar residual.
    const double kH = 1.4142;
    double Fp[9], tA[9], gT1[9], gT2[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Fp[3 * r + c] = S[0] * U[3 * r] * V[3 * c] + S[1] * U[3 * r + 1] * V[3 * c + 1];
    mat3_mul_tn(t2, Fp, tA);      // T2^T F'
    mat3_mul_tn(tA, go, gT1);     // d<G, T2^T F' T1>/dT1 = (T2^T F')^T G
    mat3_mul(Fp, t1, tA);         // F' T1
    mat3_mul_nt(tA, go, gT2);     // d/dT2 = F' T1 G^T
    float sums[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) sums[k] = 0.0f;
    const int stride = RAW ? 4 : 3;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int i = lane; i < N; i += WAVE) {
      const Pt p = global_point<RAW>(pts1, pts2, mp, i, N, hw_sx, hw_sy);
      const double w = (double)wsrc[i];
      const double z1 = p.z1, z2 = p.z2;
      const double a[3] = {s1 * ((double)p.x1 - c1x * z1), s1 * ((double)p.y1 - c1y * z1), z1};
      const double b0 = s2 * ((double)p.x2 - c2x * z2), b1 = s2 * ((double)p.y2 - c2y * z2);
      const double n2 = (a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * (b0 * b0 + b1 * b1 + 1.0);
      const bool ok = (n2 < 1e300) && (n2 > 1e-24) && (fabs(w) < 1e150);
      const double inv = ok ? fast_rsqrt(n2) : 0.0;
      double ph[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) { ph[k] = b0 * a[k] * inv; ph[3 + k] = b1 * a[k] * inv; ph[6 + k] = a[k] * inv; }
      double af = 0.0, bu = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) { af += ph[k] * f[k]; bu += ph[k] * u[k]; }
      const double gr = (g_res != nullptr) ? gsc * (double)g_res[pair * N + i] : 0.0;
      const double cu = w * w * af, cf = w * w * bu + w * gr, dotp = 2.0 * w * w * af * bu + w * gr * af;
      double gp[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) gp[k] = (cu * u[k] + cf * f[k] - ph[k] * dotp) * inv;
      const double ga0 = b0 * gp[0] + b1 * gp[3] + gp[6], ga1 = b0 * gp[1] + b1 * gp[4] + gp[7], ga2 = b0 * gp[2] + b1 * gp[5] + gp[8];
      const double gb0 = a[0] * gp[0] + a[1] * gp[1] + a[2] * gp[2], gb1 = a[0] * gp[3] + a[1] * gp[4] + a[2] * gp[5];
      double e1[3] = {0.0, 0.0, 0.0}, e2[3] = {0.0, 0.0, 0.0};
      if (g_epi != nullptr) {  // direct dependence of d_i on x1_i, x2_i
        const double x1[3] = {p.x1, p.y1, p.z1}, x2[3] = {p.x2, p.y2, p.z2};
        double l1[3], l2[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) l1[c] = x2[0] * o[c] + x2[1] * o[3 + c] + x2[2] * o[6 + c];
#pragma unroll
        for (int r = 0; r < 3; ++r) l2[r] = o[3 * r] * x1[0] + o[3 * r + 1] * x1[1] + o[3 * r + 2] * x1[2];
        const double dd = x1[0] * l1[0] + x1[1] * l1[1] + x1[2] * l1[2];
        const double n1 = fast_sqrt(l1[0] * l1[0] + l1[1] * l1[1]), nn2 = fast_sqrt(l2[0] * l2[0] + l2[1] * l2[1]);
        const double i1 = fast_rcp(n1 + 1e-6), i2 = fast_rcp(nn2 + 1e-6);
        const double Ss = i1 + i2, ad = fabs(dd);
        const double g = (ad * Ss <= (double)clamp_at) ? gsc * (double)g_epi[pair * N + i] : 0.0;
        const double sg = (dd > 0.0) ? 1.0 : ((dd < 0.0) ? -1.0 : 0.0);
        const double k1 = (n1 > 0.0) ? ad * i1 * i1 * fast_rcp(n1) : 0.0;
        const double k2 = (nn2 > 0.0) ? ad * i2 * i2 * fast_rcp(nn2) : 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          e1[c] = g * (sg * Ss * l1[c] - k2 * (l2[0] * o[c] + l2[1] * o[3 + c]));          // d n2 / d x1_c
          e2[c] = g * (sg * Ss * l2[c] - k1 * (l1[0] * o[3 * c] + l1[1] * o[3 * c + 1]));  // d n1 / d x2_c
        }
      }
      // provisional values; the Hartley terms are added in the fix-up pass below (same lane re-reads what it wrote)
      const float q1x = (float)(s1 * ga0 + e1[0]), q1y = (float)(s1 * ga1 + e1[1]);
      const float q2x = (float)(s2 * gb0 + e2[0]), q2y = (float)(s2 * gb1 + e2[1]);
      if (RAW) {
        reinterpret_cast<float4*>(g_p1)[pair * N + i] = make_float4(q1x, q1y, q2x, q2y);
      } else {
        float* d1 = g_p1 + (pair * N + i) * 3;
        float* d2 = g_p2 + (pair * N + i) * 3;
        d1[0] = q1x; d1[1] = q1y; d1[2] = (float)(ga2 - s1 * (c1x * ga0 + c1y * ga1) + e1[2]);
        d2[0] = q2x; d2[1] = q2y; d2[2] = (float)(-s2 * (c2x * gb0 + c2y * gb1) + e2[2]);
      }
      const double dx1 = (double)p.x1 - c1x, dy1 = (double)p.y1 - c1y, dx2 = (double)p.x2 - c2x, dy2 = (double)p.y2 - c2y;
      const double r1 = dx1 * dx1 + dy1 * dy1, r2 = dx2 * dx2 + dy2 * dy2;
      const double ir1 = (r1 > 0.0) ? fast_rsqrt(r1) : 0.0, ir2 = (r2 > 0.0) ? fast_rsqrt(r2) : 0.0;
      sums[0] += (float)(((double)p.x1 - c1x * z1) * ga0 + ((double)p.y1 - c1y * z1) * ga1);  // d/ds1 through the rows
      sums[1] += (float)(-s1 * z1 * ga0);
      sums[2] += (float)(-s1 * z1 * ga1);
      sums[3] += (float)(((double)p.x2 - c2x * z2) * gb0 + ((double)p.y2 - c2y * z2) * gb1);
      sums[4] += (float)(-s2 * z2 * gb0);
      sums[5] += (float)(-s2 * z2 * gb1);
      sums[6] += (float)(dx1 * ir1); sums[7] += (float)(dy1 * ir1);
      sums[8] += (float)(dx2 * ir2); sums[9] += (float)(dy2 * ir2);
    }
    double tot[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) tot[k] = (double)wave_sum(sums[k]);
    const double invN = 1.0 / (double)N;
    const double Gs1 = tot[0] + gT1[0] + gT1[4] - c1x * gT1[2] - c1y * gT1[5];
    const double Gs2 = tot[3] + gT2[0] + gT2[4] - c2x * gT2[2] - c2y * gT2[5];
    const double Gd1 = -Gs1 * s1 * s1 / kH, Gd2 = -Gs2 * s2 * s2 / kH;   // s = k / dbar
    const double Gc1x = (tot[1] - s1 * gT1[2] - Gd1 * invN * tot[6]) * invN, Gc1y = (tot[2] - s1 * gT1[5] - Gd1 * invN * tot[7]) * invN;
    const double Gc2x = (tot[4] - s2 * gT2[2] - Gd2 * invN * tot[8]) * invN, Gc2y = (tot[5] - s2 * gT2[5] - Gd2 * invN * tot[9]) * invN;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int i = lane; i < N; i += WAVE) {
      const Pt p = global_point<RAW>(pts1, pts2, mp, i, N, hw_sx, hw_sy);
      const double dx1 = (double)p.x1 - c1x, dy1 = (double)p.y1 - c1y, dx2 = (double)p.x2 - c2x, dy2 = (double)p.y2 - c2y;
      const double r1 = dx1 * dx1 + dy1 * dy1, r2 = dx2 * dx2 + dy2 * dy2;
      const double ir1 = (r1 > 0.0) ? fast_rsqrt(r1) : 0.0, ir2 = (r2 > 0.0) ? fast_rsqrt(r2) : 0.0;
      const float a1x = (float)(Gd1 * invN * dx1 * ir1 + Gc1x), a1y = (float)(Gd1 * invN * dy1 * ir1 + Gc1y);
      const float a2x = (float)(Gd2 * invN * dx2 * ir2 + Gc2x), a2y = (float)(Gd2 * invN * dy2 * ir2 + Gc2y);
      if (RAW) {
        float4 q = reinterpret_cast<float4*>(g_p1)[pair * N + i];
        q.x = (q.x + a1x) * hw_sx; q.y = (q.y + a1y) * hw_sy; q.z = (q.z + a2x) * hw_sx; q.w = (q.w + a2y) * hw_sy;
        reinterpret_cast<float4*>(g_p1)[pair * N + i] = q;  // chain through x^ = 2x/W - 1
      } else {
        float* d1 = g_p1 + (pair * N + i) * 3;
        float* d2 = g_p2 + (pair * N + i) * 3;
        d1[0] += a1x; d1[1] += a1y; d2[0] += a2x; d2[1] += a2y;
      }
    }
  }
}

}  // namespace

int dfepe_loss_head_from_workspace(const void* workspace_desc, hipStream_t st);  // loss_tail.hip

extern "C" int dfepe_w8pt_bwd(const float* pts1, const float* pts2, const float* weights, int B, int N, int n_weight_sets,
                              unsigned flags,
                              float image_w, float image_h, float clamp_at, const float* save, const float* F_out,
                              const float* g_F, const float* g_residual, const float* g_epi,
                              const float* g_weights_extra, const float* g_scale, float* g_weights, float* g_pts1, float* g_pts2,
                              const void* pending_loss_head, void* stream) {
  const bool raw = (flags & DFEPE_W8PT_RAW_MATCHES) != 0;
  const int logits_mode = (flags & DFEPE_W8PT_LOGITS) ? 1 : 0;
  if (B < 0 || N <= 0 || n_weight_sets < 1) return DFEPE_ERR_INVALID_ARG;
  if (flags & (DFEPE_W8PT_SQRT2 | DFEPE_W8PT_FORCE_110 | DFEPE_W8PT_NO_HARTLEY)) return DFEPE_ERR_UNSUPPORTED;
  const unsigned variant = flags & DFEPE_W8PT_NO_ROWNORM;  // the one variant with an adjoint (row kernels, weight gradients)
  if (variant && (g_pts1 != nullptr || !dfepe_w8pt_use_rows(N, (long long)B * n_weight_sets, flags))) return DFEPE_ERR_UNSUPPORTED;
  if (B == 0) return DFEPE_OK;
  if (!pts1 || (!raw && !pts2) || !weights || !save || !g_weights) return DFEPE_ERR_INVALID_ARG;
  if (g_epi && !F_out) return DFEPE_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(pending_loss_head) & 15u) return DFEPE_ERR_INVALID_ARG;
  const bool pgrad = g_pts1 != nullptr;
  if (pgrad && n_weight_sets != 1) return DFEPE_ERR_UNSUPPORTED;  // point gradients of shared correspondences would need a sum over the sets
  const int Bm = B;
  B *= n_weight_sets;
  if (!raw && ((g_pts1 == nullptr) != (g_pts2 == nullptr))) return DFEPE_ERR_INVALID_ARG;
  if (raw && pgrad && (reinterpret_cast<uintptr_t>(g_pts1) & 15u)) return DFEPE_ERR_INVALID_ARG;
  if (raw && !(image_w > 0.f && image_h > 0.f)) return DFEPE_ERR_INVALID_ARG;
  if (raw && (reinterpret_cast<uintptr_t>(pts1) & 15u)) return DFEPE_ERR_INVALID_ARG;
  if (flags & ~DFEPE_W8PT_ALL_FLAGS) return DFEPE_ERR_INVALID_ARG;
  if (dfepe_w8pt_use_rows(N, (long long)B, flags)) {  // same rule as dfepe_w8pt_fwd (B already counts the weight sets): the record formats differ
    W8BwdArgs A;
    A.pts1 = pts1; A.pts2 = pts2; A.wts = weights;
    A.Bm = Bm; A.B = B; A.N = N;
    A.hw_sx = raw ? 2.0f / image_w : 0.f; A.hw_sy = raw ? 2.0f / image_h : 0.f; A.clamp_at = clamp_at;
    A.save = save; A.F_out = F_out; A.g_F = g_F; A.g_res = g_residual; A.g_epi = g_epi; A.g_w_extra = g_weights_extra; A.g_scale = g_scale;
    A.g_w = g_weights; A.g_p1 = g_pts1; A.g_p2 = g_pts2; A.logits_mode = logits_mode; A.variant = variant; A.pending_head = pending_loss_head;
    return dfepe_w8pt16_bwd_launch(A, raw, static_cast<hipStream_t>(stream));
  }
  const int waves = 4;
  // large N, batch small enough to be resident at once: one workgroup per pair (N = 1000, B = 512: see DESIGN.md)
  const bool coop = !pgrad && N >= 256 && B <= 1024;
  const dim3 grid(coop ? B : (B + waves - 1) / waves), block(64 * waves);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float hw_sx = raw ? 2.0f / image_w : 0.f, hw_sy = raw ? 2.0f / image_h : 0.f;
#define DFEPE_LAUNCH_BWD(R, P, C)                                                                                           \
  hipLaunchKernelGGL((w8pt_bwd_kernel<R, P, C>), grid, block, 0, st, pts1, pts2, weights, B, Bm, N, hw_sx, hw_sy, clamp_at, save, \
                     F_out, g_F, g_residual, g_epi, g_weights_extra, g_scale, logits_mode, g_weights, g_pts1, g_pts2)
  if (raw) {
    if (pgrad) DFEPE_LAUNCH_BWD(true, true, false); else if (coop) DFEPE_LAUNCH_BWD(true, false, true); else DFEPE_LAUNCH_BWD(true, false, false);
  } else {
    if (pgrad) DFEPE_LAUNCH_BWD(false, true, false); else if (coop) DFEPE_LAUNCH_BWD(false, false, true); else DFEPE_LAUNCH_BWD(false, false, false);
  }
#undef DFEPE_LAUNCH_BWD
  if (hipGetLastError() != hipSuccess) return DFEPE_ERR_HIP;
  return (pending_loss_head != nullptr) ? dfepe_loss_head_from_workspace(pending_loss_head, st) : DFEPE_OK;  // a launch of its own here
}
