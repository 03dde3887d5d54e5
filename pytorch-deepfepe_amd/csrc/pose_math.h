// Pose decomposition math shared by pose.hip (one lane per (layer, pair)) and loss_tail (the fused loss tail): the
// four-fold decomposition of E^T, quaternion / translation L2 errors, candidate selection, angular metrics and the analytic
// adjoint w.r.t. E.  Restates the per-layer, per-sample loop of get_Rt_loss (deepFEPE/train_good_utils.py:96-239):
//   _get_M2s      dsac_tools/utils_F.py:478-498   (U W V^T, W negated when det < 0, t = u3/|u3|)
//   _R_to_q       dsac_tools/utils_geo.py:58-86   (trace method on m = R^T, 4 branches, q0 >= 0)
//   _l2_error     dsac_tools/utils_geo.py:165-167
//   selection by strict '<'                       train_good_utils.py:160-168 (R and t picked independently)
//   rot12_to_angle_error / vector_angle           dsac_tools/utils_geo.py:150-155, 175-179
// Everything is 3x3 work in fp64 registers of ONE lane (closed-form 3x3 SVD, dfepe_math.h svd3_closed).  The adjoint of the SVD uses the
// combined (1,2)-block form Z12/(s1+s2): for R = U W V^T the generic 1/(s1^2-s2^2) terms cancel analytically, so true
// essential matrices (s1 == s2) stay finite.  cv2.Rodrigues is replaced by atan2(|axis|, trace-1) (same angle; OpenCV
// arithmetic is unpinned).  Pure per-lane C++ (needs dfepe_math.h).
#pragma once
#include "dfepe_math.h"

struct Quat {
  double q[4];
  int branch;
  double tr, sg;
};

__device__ __forceinline__ Quat rot_to_quat(const double* R) {
  // m = R^T : m[i][j] = R[j][i]
#define M_(i, j) R[3 * (j) + (i)]
  Quat o;
  double v[4], t;
  if (M_(2, 2) < 0.0) {
    if (M_(0, 0) > M_(1, 1)) {
      t = 1.0 + M_(0, 0) - M_(1, 1) - M_(2, 2);
      v[0] = M_(1, 2) - M_(2, 1); v[1] = t; v[2] = M_(0, 1) + M_(1, 0); v[3] = M_(2, 0) + M_(0, 2);
      o.branch = 0;
    } else {
      t = 1.0 - M_(0, 0) + M_(1, 1) - M_(2, 2);
      v[0] = M_(2, 0) - M_(0, 2); v[1] = M_(0, 1) + M_(1, 0); v[2] = t; v[3] = M_(1, 2) + M_(2, 1);
      o.branch = 1;
    }
  } else {
    if (M_(0, 0) < -M_(1, 1)) {
      t = 1.0 - M_(0, 0) - M_(1, 1) + M_(2, 2);
      v[0] = M_(0, 1) - M_(1, 0); v[1] = M_(2, 0) + M_(0, 2); v[2] = M_(1, 2) + M_(2, 1); v[3] = t;
      o.branch = 2;
    } else {
      t = 1.0 + M_(0, 0) + M_(1, 1) + M_(2, 2);
      v[0] = t; v[1] = M_(1, 2) - M_(2, 1); v[2] = M_(2, 0) - M_(0, 2); v[3] = M_(0, 1) - M_(1, 0);
      o.branch = 3;
    }
  }
#undef M_
  const double sc = 0.5 * rsqrt_nr<2>(t);  // t >= 1 in the branch taken
  o.sg = (v[0] * sc < 0.0) ? -1.0 : 1.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) o.q[k] = o.sg * sc * v[k];
  o.tr = t;
  return o;
}

// gradient w.r.t. R (row-major) of <gq, q(R)> for the branch recorded in `qq`
__device__ __forceinline__ void rot_to_quat_bwd(const Quat& qq, const double* gq, double* gR) {
  double gm[9];  // w.r.t. m = R^T, row-major
#pragma unroll
  for (int k = 0; k < 9; ++k) gm[k] = 0.0;
  const double itr = rcp_nr<2>(qq.tr);
  const double sc = qq.sg * 0.5 * rsqrt_nr<2>(qq.tr);
  const double gv0 = sc * gq[0], gv1 = sc * gq[1], gv2 = sc * gq[2], gv3 = sc * gq[3];
  const double gt = -0.5 * (qq.q[0] * gq[0] + qq.q[1] * gq[1] + qq.q[2] * gq[2] + qq.q[3] * gq[3]) * itr;
#define ADD(i, j, c) gm[3 * (i) + (j)] += (c)
  if (qq.branch == 0) {
    const double T = gv1 + gt;
    ADD(0, 0, T); ADD(1, 1, -T); ADD(2, 2, -T);
    ADD(1, 2, gv0); ADD(2, 1, -gv0); ADD(0, 1, gv2); ADD(1, 0, gv2); ADD(2, 0, gv3); ADD(0, 2, gv3);
  } else if (qq.branch == 1) {
    const double T = gv2 + gt;
    ADD(0, 0, -T); ADD(1, 1, T); ADD(2, 2, -T);
    ADD(2, 0, gv0); ADD(0, 2, -gv0); ADD(0, 1, gv1); ADD(1, 0, gv1); ADD(1, 2, gv3); ADD(2, 1, gv3);
  } else if (qq.branch == 2) {
    const double T = gv3 + gt;
    ADD(0, 0, -T); ADD(1, 1, -T); ADD(2, 2, T);
    ADD(0, 1, gv0); ADD(1, 0, -gv0); ADD(2, 0, gv1); ADD(0, 2, gv1); ADD(1, 2, gv2); ADD(2, 1, gv2);
  } else {
    const double T = gv0 + gt;
    ADD(0, 0, T); ADD(1, 1, T); ADD(2, 2, T);
    ADD(1, 2, gv1); ADD(2, 1, -gv1); ADD(2, 0, gv2); ADD(0, 2, -gv2); ADD(0, 1, gv3); ADD(1, 0, -gv3);
  }
#undef ADD
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) gR[3 * r + c] = gm[3 * c + r];
}

struct Pose {
  double U[9], S[3], V[9];
  double R[2][9];
  double t[3];
  double sd;  // sign applied to U W V^T so that det > 0
  Quat q[2];
  double tg[3];
  double qe[2], te[2];
  int qi, ti;
};

__device__ inline void pose_forward(const float* E, const float* q_gt, const float* t_gt, Pose& P) {
  double Ec[9];  // E^T: get_Rt_loss decomposes the transposed matrix (train_good_utils.py:106)
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Ec[3 * r + c] = (double)E[3 * c + r];
  svd3_closed(Ec, P.U, P.S, P.V);
  const double* U = P.U;
  const double* V = P.V;
  // U W = [u2, -u1, u3],  U W^T = [-u2, u1, u3]   (columns);   R = (U W) V^T
  double UW[9], UWt[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    UW[3 * r + 0] = U[3 * r + 1]; UW[3 * r + 1] = -U[3 * r + 0]; UW[3 * r + 2] = U[3 * r + 2];
    UWt[3 * r + 0] = -U[3 * r + 1]; UWt[3 * r + 1] = U[3 * r + 0]; UWt[3 * r + 2] = U[3 * r + 2];
  }
  mat3_mul_nt(UW, V, P.R[0]);
  mat3_mul_nt(UWt, V, P.R[1]);
  const double* R = P.R[0];
  const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
  P.sd = (det < 0.0) ? -1.0 : 1.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) { P.R[0][k] *= P.sd; P.R[1][k] *= P.sd; }
  const double iun = rsqrt_nr<2>(U[2] * U[2] + U[5] * U[5] + U[8] * U[8]);
  P.t[0] = U[2] * iun; P.t[1] = U[5] * iun; P.t[2] = U[8] * iun;
  const double ign = rcp_nr<2>(fmax(sqrt_nr<2>((double)t_gt[0] * t_gt[0] + (double)t_gt[1] * t_gt[1] + (double)t_gt[2] * t_gt[2]), 1e-12));
#pragma unroll
  for (int k = 0; k < 3; ++k) P.tg[k] = (double)t_gt[k] * ign;  // F.normalize(p=2, dim=0) (:151)
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    P.q[c] = rot_to_quat(P.R[c]);
    double e = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const double d = P.q[c].q[k] - (double)q_gt[k]; e += d * d; }
    P.qe[c] = sqrt_nr<2>(e);
    const double sgn = (c == 0) ? 1.0 : -1.0;
    double f = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double d = sgn * P.t[k] - P.tg[k]; f += d * d; }
    P.te[c] = sqrt_nr<2>(f);
  }
  P.qi = (P.qe[0] < P.qe[1]) ? 0 : 1;  // strict '<' (:160-161)
  P.ti = (P.te[0] < P.te[1]) ? 0 : 1;
}

// adjoint: gql = d loss / d q_l2, gtl = d loss / d t_l2 of the selected candidates -> gE (row-major, w.r.t. E itself)
__device__ inline void pose_backward(const Pose& P, const float* q_gt, double gql, double gtl, double* gE) {
  // d|q - q_gt| / dq
  double gq[4], gR[9];
  const Quat& qq = P.q[P.qi];
  const double iqe = (P.qe[P.qi] > 0.0) ? gql * rcp_nr<2>(P.qe[P.qi]) : 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) gq[k] = iqe * (qq.q[k] - (double)q_gt[k]);
  rot_to_quat_bwd(qq, gq, gR);
  // d|+-t - t_gt| / du3   (t = u3 / |u3|, |u3| = 1)
  const double sgn = (P.ti == 0) ? 1.0 : -1.0;
  const double ite = (P.te[P.ti] > 0.0) ? gtl * rcp_nr<2>(P.te[P.ti]) : 0.0;
  double gt[3], gu3[3], tdot = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    gt[k] = ite * (sgn * P.t[k] - P.tg[k]);
    tdot += P.t[k] * gt[k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) gu3[k] = sgn * (gt[k] - P.t[k] * tdot);
  // Pm = sd U^T gR V ;  U^T gU = Pm Wk^T (+ U^T gu3 in column 3),  V^T gV = Pm^T Wk
  double tmp[9], Pm[9];
  mat3_mul_tn(P.U, gR, tmp);
  mat3_mul(tmp, P.V, Pm);
#pragma unroll
  for (int k = 0; k < 9; ++k) Pm[k] *= P.sd;
  // W = [[0,-1,0],[1,0,0],[0,0,1]];  candidate 0 uses W, candidate 1 uses W^T
  const double w01 = (P.qi == 0) ? -1.0 : 1.0;  // Wk[0][1];  Wk[1][0] = -w01
  double A[9], Bm[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    // (Pm Wk^T)[r][c] = sum_k Pm[r][k] Wk[c][k]
    A[3 * r + 0] = Pm[3 * r + 1] * w01;
    A[3 * r + 1] = Pm[3 * r + 0] * (-w01);
    A[3 * r + 2] = Pm[3 * r + 2];
    // (Pm^T Wk)[r][c] = sum_k Pm[k][r] Wk[k][c]
    Bm[3 * r + 0] = Pm[3 + r] * (-w01);
    Bm[3 * r + 1] = Pm[r] * w01;
    Bm[3 * r + 2] = Pm[6 + r];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) A[3 * r + 2] += P.U[r] * gu3[0] + P.U[3 + r] * gu3[1] + P.U[6 + r] * gu3[2];
  double Mid[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (i == j) { Mid[3 * i + j] = 0.0; continue; }
      const double Z = A[3 * i + j] - A[3 * j + i];
      const double Y = Bm[3 * i + j] - Bm[3 * j + i];
      if (i < 2 && j < 2) {
        Mid[3 * i + j] = Z * rcp_nr<2>(fmax(P.S[0] + P.S[1], 1e-30));
      } else {
        double den = P.S[j] * P.S[j] - P.S[i] * P.S[i];
        if (fabs(den) < 1e-30) den = (den < 0.0) ? -1e-30 : 1e-30;
        Mid[3 * i + j] = (Z * P.S[j] + P.S[i] * Y) * rcp_nr<2>(den);
      }
    }
  double gEc[9];
  mat3_mul(P.U, Mid, tmp);
  mat3_mul_nt(tmp, P.V, gEc);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) gE[3 * r + c] = gEc[3 * c + r];  // E = Ec^T
}

// angular metrics of the selected candidates (degrees): rotation angle of R_est R_gt^T, angle between +-t and t_gt
__device__ inline double pose_R_deg(const Pose& P, const float* R_gt) {
  const double* Re = P.R[P.qi];
  double D[9], Rg[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rg[k] = (double)R_gt[k];
  mat3_mul_nt(Re, Rg, D);
  const double ax = D[7] - D[5], ay = D[2] - D[6], az = D[3] - D[1];
  return atan2(sqrt(ax * ax + ay * ay + az * az), D[0] + D[4] + D[8] - 1.0) * 57.29577951308232;
}
__device__ inline double pose_t_deg(const Pose& P) {
  const double sgn = (P.ti == 0) ? 1.0 : -1.0;
  double dot = 0.0, n1 = 0.0, n2 = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double a = sgn * P.t[k];
    dot += a * P.tg[k]; n1 += a * a; n2 += P.tg[k] * P.tg[k];
  }
  const double den = (sqrt(n1) + 1e-10) * (sqrt(n2) + 1e-10) + 1e-10;  // utils_geo.py:172-179
  return acos(fmin(fmax(dot / den, -1.0), 1.0)) * 57.29577951308232;
}
