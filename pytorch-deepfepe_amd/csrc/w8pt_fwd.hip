// w8pt_fwd — weighted normalised 8-point fit, forward.  One wavefront (64 lanes) per image pair.
//
// Restates (not from the reference source) the arithmetic of
//   NormalizeAndExpand_HW   deepFEPE/models/DeepFNet.py:93-120   (fused when RAW)
//   Fit.normalize           deepFEPE/models/DeepFNet.py:148-179  (Hartley, unit weights, literal 1.4142)
//   Fit.weighted_svd        deepFEPE/models/DeepFNet.py:181-257
//   compute_epi_residual    deepFEPE/dsac_tools/utils_F.py:400-413
//
// Phases (all wave-synchronous, no block barrier; the 4 waves of a block are independent pairs):
//   0  coalesced global -> LDS copy of the pair's correspondences (+ weights), coordinate sums
//   1  mean distance to the centroid  -> Hartley scale                      (wave reductions, fp64)
//   2  rows p/|p| * w, the 36 distinct sums of X^T X per lane in fp64        (Kronecker structure of the rows)
//   3  recursive-halving reduce-scatter of the 36 distinct sums across the wave -> 9x9 M in LDS (fp64)
//   4  all nine eigenpairs of M/trace(M) by a parallel-ordering two-sided Jacobi in fp32 (systolic form: the
//      four rotation pairs always sit at positions (0,1)(2,3)(4,5)(6,7), the data is permuted between rounds,
//      so every LDS address is loop-invariant); then the one eigenvector the solver needs is polished in fp64
//      against the fp64 M by residual correction in the Jacobi basis (converges to fp64 accuracy)
//   5  f = that eigenvector; 3x3 SVD; F' = F - s3 u3 v3^T; out = T2^T F' T1
//   6  residual_i = X_i . f and the symmetric epipolar residual per correspondence (coalesced stores)
// No MFMA: the only contraction (X^T X, 9xN by Nx9) is far too skinny; the rest is eigen work.
#include "dfepe_common.h"
#include "w8pt16_body.h"  // W8Args

int dfepe_w8pt16_fwd_launch(const W8Args& A, bool raw, hipStream_t st);  // w8pt16.hip

namespace {

// X^T X = sum_i k_i^2 (b b^T) (x) (a a^T) with a = (x1~,y1~,z1), b = (x2~,y2~,1): only 6 x 6 = 36 distinct sums.
// pair tables for the 6 distinct entries of a symmetric 3x3 outer product
__constant__ unsigned char kSymR[6] = {0, 0, 0, 1, 1, 2};
__constant__ unsigned char kSymC[6] = {0, 1, 2, 1, 2, 2};

// Loop-invariant addresses of a lane's two Jacobi work items (see phase 4a), one 16-byte record per lane, built at
// compile time: one global_load_dwordx4 replaces ~150 integer instructions and four dependent table look-ups.
struct JacLane {
  unsigned short rd_own, rd_par, wr2, wr2t;  // slot 2 (A element or V element 64..80): float offsets from A32
  unsigned short rd_v, wr_v;                 // slot 1 (V element = lane): float offsets from V32
  unsigned char ci, cj, c0, flags;           // CS slots of the row / column rotation (slot 2) and of slot 1
};
constexpr unsigned kJacIsA = 1, kJacIsV2 = 2, kJacOdd2 = 4, kJacOffDiag = 16;
struct JacTable { JacLane l[64]; };
constexpr JacTable make_jac_table() {
  constexpr int perm[9] = {8, 3, 0, 5, 2, 7, 4, 6, 1};  // seat permutation of the round-robin tournament: position p moves to perm[p] after every round
  JacTable t{};
  int ai = 0, aj = 0;  // walks the upper triangle row by row
  for (int lane = 0; lane < 64; ++lane) {
    const bool a_item = lane < 45, v_item = (lane >= 45 && lane < 62);
    const int ti = a_item ? ai : (v_item ? (lane + 19) / 9 : 0);  // V element e = 64 + (lane - 45) = lane + 19
    const int tj = a_item ? aj : (v_item ? (lane + 19) % 9 : 0);
    if (a_item) { if (++aj == 9) { ++ai; aj = ai; } }
    const int tip = (ti < 8) ? (ti ^ 1) : 8;
    const int mat2 = a_item ? 0 : 90;                      // slot-2 matrix base (A32 or V32), in floats from A32
    const int rd_own = mat2 + ti * 10 + (tj & ~1);         // float2 {even column, odd column} of my row
    const int vi0 = lane / 9, vj0 = lane % 9;
    t.l[lane].rd_own = (unsigned short)rd_own;
    t.l[lane].rd_par = (unsigned short)(a_item ? tip * 10 + (tj & ~1) : rd_own);  // same columns, partner row (A only)
    t.l[lane].wr2 = (unsigned short)(a_item ? perm[ti] * 10 + perm[tj] : mat2 + ti * 10 + perm[tj]);
    t.l[lane].wr2t = (unsigned short)(perm[tj] * 10 + perm[ti]);                   // mirror (A items only)
    t.l[lane].rd_v = (unsigned short)(vi0 * 10 + (vj0 & ~1));
    t.l[lane].wr_v = (unsigned short)(vi0 * 10 + perm[vj0]);
    t.l[lane].ci = (unsigned char)(4 * (a_item ? ti : 8));                      // V items: identity row rotation
    t.l[lane].cj = (unsigned char)(4 * tj + ((tj & 1) ? 2 : 0));               // odd column: swapped coefficients
    t.l[lane].c0 = (unsigned char)(4 * vj0 + ((vj0 & 1) ? 2 : 0));
    t.l[lane].flags = (unsigned char)((a_item ? kJacIsA : 0) | (v_item ? kJacIsV2 : 0) | ((tj & 1) ? kJacOdd2 : 0) |
                                      ((a_item && ti != tj) ? kJacOffDiag : 0));
  }
  return t;
}
__constant__ JacTable kJac = make_jac_table();

constexpr int kWsDoubles = 232;  // per-wave workspace in LDS (1856 B, 16-B multiple): see the carve in the kernel
constexpr float kClusterTol = 4e-6f;  // fp32 Jacobi eigenvalues (unit trace) closer than this to the selected one are re-resolved in fp64
constexpr int kMaxSweeps = 8;  // safety bound only: the sweeps run until off(A)^2 <= kJacobiTol (4-5 sweeps, a 6th for ~1 % of the pairs); stopping
                               // early leaves eigenpairs too rough for the polish when the gap is a few 1e-6 (scripts/stress_parity.py)
constexpr float kJacobiTol = 1e-13f;  // fp32 sweeps stop when off(A)^2 <= tol (A is scaled to unit trace)
constexpr int kRefineIters = 12;  // upper bound; the loop leaves as soon as the fp64 residual is at round-off level

template <bool RAW>
__device__ __forceinline__ Pt lds_point(const float* P, int i, int npad) {
  Pt p;
  if (RAW) {
    const float4 m = reinterpret_cast<const float4*>(P)[i];
    p.x1 = m.x; p.y1 = m.y; p.z1 = 1.0f; p.x2 = m.z; p.y2 = m.w; p.z2 = 1.0f;
  } else {
    const float* a = P + 3 * i;
    const float* b = P + 3 * npad + 3 * i;
    p.x1 = a[0]; p.y1 = a[1]; p.z1 = a[2]; p.x2 = b[0]; p.y2 = b[1]; p.z2 = b[2];
  }
  return p;
}

// One halving step of the reduce-scatter: CNT live values per lane -> (CNT+1)/2.  The partner is lane^MASK through
// ds_bpermute for the two cross-row steps (MASK 32, 16) and a DPP reflection / quad permute inside a 16-lane row
// (row_mirror pairs k with 15-k, row_half_mirror k with 7-k, quad_perm for xor 2 and xor 1): any pairing works as long
// as the two partners sit on opposite sides of the decision bit, which all of these do.
template <int CNT, int DPP_CTRL>
__device__ __forceinline__ void halve(double* a, bool upper, int mask) {
  constexpr int H = (CNT + 1) / 2;
#pragma unroll
  for (int k = 0; k < H; ++k) {
    const double lo = a[k];
    const double hi = (k + H < CNT) ? a[k + H] : 0.0;
    const double send = upper ? lo : hi;
    const double keep = upper ? hi : lo;
    const double got = (DPP_CTRL == 0) ? __shfl_xor(send, mask, WAVE) : dpp_f64<(DPP_CTRL == 0) ? 0xB1 : DPP_CTRL>(send);
    a[k] = keep + got;
  }
}

// WPP (wavefronts per pair) = 1: one wavefront per pair (blocks of 4, 2 or 1 independent wavefronts, no block barrier).
// WPP = 2 or 4: one cooperative workgroup per pair, for N so large that the pair's staging area leaves a CU with few
//   wavefronts: the wavefronts share the per-correspondence phases (0, 1, 2, 6), combine their partial sums through LDS,
//   and wavefront 0 alone runs the eigen phases (3-5) while the others wait at a barrier.
constexpr int kCoopBytes = 1536;  // COOP: [16] centroid, [8] Hartley, [144] moment partials, [9] f, then 17 floats (max, sum, F)
template <bool RAW, int WPP>
__global__ void __launch_bounds__(256, 4)
w8pt_fwd_kernel(const float* __restrict__ pts1, const float* __restrict__ pts2, const float* __restrict__ wts,
                int B, int Bm, int N, int npad, int wave_bytes, float hw_sx, float hw_sy, float clamp_at,
                float* __restrict__ F_out, float* __restrict__ residual, float* __restrict__ epi_res,
                float* __restrict__ save, float* __restrict__ weights_out, int logits_mode, unsigned variant) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform by construction
  constexpr bool COOP = WPP > 1;
  const int pair = COOP ? (int)blockIdx.x : (int)(blockIdx.x * (blockDim.x >> 6) + wave);
  if (!COOP && pair >= B) return;  // whole wave leaves; the per-wavefront variant has no block-level barrier
  constexpr int NT = WPP * WAVE;                 // threads that stride over the pair's correspondences
  const int tid = COOP ? (int)threadIdx.x : lane;

  unsigned char* base = COOP ? smem : smem + (size_t)wave * wave_bytes;
  double* M64 = reinterpret_cast<double*>(base);            // [81] X^T X, fp64, natural index order      0..648
  double* SCR = M64 + 81;                                   // [32] exchange scratch for the refinement  648..904
  float* A32 = reinterpret_cast<float*>(base + 904);        // [9][10] Jacobi iterate (position space, row stride 10) 904..1264
  float* V32 = A32 + 90;                                    // [9][10] accumulated rotations                        1264..1624
  float4* CS = reinterpret_cast<float4*>(base + 1632);      // [9]  (c, sh, sh, c) per position: .xy for an even, .zw for an odd column 1632..1776
  double* LAMC = reinterpret_cast<double*>(base + 1776);    // [9]  eigenvalues in fp64 (Ritz values of the stored vectors) 1776..1848
  float* P = reinterpret_cast<float*>(base + kWsDoubles * sizeof(double));
  float* W = P + (RAW ? 4 : 6) * npad;
  double* RED = reinterpret_cast<double*>(W + npad);         // COOP only: cross-wavefront exchange (kCoopBytes)
  float* REDF = reinterpret_cast<float*>(RED + 177);
  auto pair_sync = [&]() { if (COOP) __syncthreads(); else wave_sync(); };

  // ---- phase 0: stage the pair in LDS, coordinate sums ------------------------------------------
  double sx1 = 0, sy1 = 0, sx2 = 0, sy2 = 0;
  const float* wsrc = wts + (size_t)pair * N;
  const size_t mp = (size_t)(pair % Bm);  // several weight sets may share one set of correspondences (n_weight_sets > 1)
  if (RAW) {
    const float4* src = reinterpret_cast<const float4*>(pts1) + mp * N;
  #pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
  for (int i = tid; i < N; i += NT) {
      float4 m = src[i];
      m.x = fmaf(m.x, hw_sx, -1.0f);
      m.y = fmaf(m.y, hw_sy, -1.0f);
      m.z = fmaf(m.z, hw_sx, -1.0f);
      m.w = fmaf(m.w, hw_sy, -1.0f);
      reinterpret_cast<float4*>(P)[i] = m;
      W[i] = wsrc[i];
      sx1 += m.x; sy1 += m.y; sx2 += m.z; sy2 += m.w;
    }
  } else {
    const float* s1p = pts1 + mp * N * 3;
    const float* s2p = pts2 + mp * N * 3;
    for (int t = tid; t < 3 * N; t += NT) {
      P[t] = s1p[t];
      P[3 * npad + t] = s2p[t];
    }
    for (int i = tid; i < N; i += NT) W[i] = wsrc[i];
    pair_sync();
  #pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
  for (int i = tid; i < N; i += NT) {
      const Pt p = lds_point<false>(P, i, npad);
      sx1 += p.x1; sy1 += p.y1; sx2 += p.x2; sy2 += p.y2;
    }
  }
  if (logits_mode) {
    // fused F.softmax(logits, dim=N) (DeepFNet.py:443,512): W holds the logits at this point
    wave_sync();
    if (!COOP && N <= 2 * WAVE) {
      // the (at most two) logits of a lane stay in registers: one LDS round trip instead of three passes over W
      const int i1 = lane + WAVE;
      const float l0 = (lane < N) ? W[lane] : -INFINITY, l1 = (i1 < N) ? W[i1] : -INFINITY;
      const float mxr = wave_max(fmaxf(l0, l1));
      const float e0 = (lane < N) ? expf(l0 - mxr) : 0.0f, e1 = (i1 < N) ? expf(l1 - mxr) : 0.0f;
      const float invr = 1.0f / wave_sum(e0 + e1);
      if (lane < N) {
        W[lane] = e0 * invr;
        if (weights_out != nullptr) weights_out[(size_t)pair * N + lane] = e0 * invr;
      }
      if (i1 < N) {
        W[i1] = e1 * invr;
        if (weights_out != nullptr) weights_out[(size_t)pair * N + i1] = e1 * invr;
      }
    } else {
    float mx = -INFINITY;
    for (int i = tid; i < N; i += NT) mx = fmaxf(mx, W[i]);
    mx = wave_max(mx);
    if (COOP) {
      if (lane == 0) REDF[wave] = mx;
      __syncthreads();
      mx = REDF[0];
#pragma unroll
      for (int k = 1; k < WPP; ++k) mx = fmaxf(mx, REDF[k]);
    }
    float sm = 0.0f;
    for (int i = tid; i < N; i += NT) {
      const float e = expf(W[i] - mx);
      W[i] = e;
      sm += e;
    }
    sm = wave_sum(sm);
    if (COOP) {
      if (lane == 0) REDF[4 + wave] = sm;
      __syncthreads();
      sm = REDF[4];
#pragma unroll
      for (int k = 1; k < WPP; ++k) sm += REDF[4 + k];
    }
    const float inv = 1.0f / sm;
    for (int i = tid; i < N; i += NT) {
      const float w = W[i] * inv;
      W[i] = w;
      if (weights_out != nullptr) weights_out[(size_t)pair * N + i] = w;
    }
    }
  }
  const double invN = 1.0 / (double)N;
  const bool hartley = (variant & DFEPE_W8PT_NO_HARTLEY) == 0;  // wave-uniform
  sx1 = wave_sum(sx1); sy1 = wave_sum(sy1); sx2 = wave_sum(sx2); sy2 = wave_sum(sy2);
  if (COOP) {
    if (lane == 0) { RED[4 * wave] = sx1; RED[4 * wave + 1] = sy1; RED[4 * wave + 2] = sx2; RED[4 * wave + 3] = sy2; }
    __syncthreads();
    sx1 = RED[0]; sy1 = RED[1]; sx2 = RED[2]; sy2 = RED[3];
#pragma unroll
    for (int k = 1; k < WPP; ++k) { sx1 += RED[4 * k]; sy1 += RED[4 * k + 1]; sx2 += RED[4 * k + 2]; sy2 += RED[4 * k + 3]; }
  }
  const double c1x = hartley ? to_sgpr(sx1 * invN) : 0.0, c1y = hartley ? to_sgpr(sy1 * invN) : 0.0;
  const double c2x = hartley ? to_sgpr(sx2 * invN) : 0.0, c2y = hartley ? to_sgpr(sy2 * invN) : 0.0;
  wave_sync();

  // ---- phase 1: Hartley scale -------------------------------------------------------------------
  double d1 = 0, d2 = 0;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
  for (int i = tid; i < N; i += NT) {
    const Pt p = lds_point<RAW>(P, i, npad);
    const double ax = (double)p.x1 - c1x, ay = (double)p.y1 - c1y;
    const double bx = (double)p.x2 - c2x, by = (double)p.y2 - c2y;
    d1 += fast_sqrt(ax * ax + ay * ay);
    d2 += fast_sqrt(bx * bx + by * by);
  }
  // Fit.normalize uses the literal 1.4142, not sqrt(2) (DeepFNet.py:168); utils_F._normalize_XY uses np.sqrt(2)
  const double hscale = (variant & DFEPE_W8PT_SQRT2) ? 1.4142135623730951 : 1.4142;
  d1 = wave_sum(d1); d2 = wave_sum(d2);
  if (COOP) {
    if (lane == 0) { RED[16 + 2 * wave] = d1; RED[17 + 2 * wave] = d2; }
    __syncthreads();
    d1 = RED[16]; d2 = RED[17];
#pragma unroll
    for (int k = 1; k < WPP; ++k) { d1 += RED[16 + 2 * k]; d2 += RED[17 + 2 * k]; }
  }
  const double s1 = hartley ? to_sgpr(hscale * fast_rcp(d1 * invN)) : 1.0;
  const double s2 = hartley ? to_sgpr(hscale * fast_rcp(d2 * invN)) : 1.0;

  // ---- phase 2: X^T X as 36 distinct fp64 sums per lane (exact products of fp32-derived factors) ---------
  double acc[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = 0.0;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
  for (int i = tid; i < N; i += NT) {
    const Pt p = lds_point<RAW>(P, i, npad);
    const double w = (double)W[i];
    const double z1 = p.z1, z2 = p.z2;
    const double a0 = s1 * ((double)p.x1 - c1x * z1), a1 = s1 * ((double)p.y1 - c1y * z1), a2 = z1;
    const double b0 = s2 * ((double)p.x2 - c2x * z2), b1 = s2 * ((double)p.y2 - c2y * z2);
    const double n2 = (a0 * a0 + a1 * a1 + a2 * a2) * (b0 * b0 + b1 * b1 + 1.0);  // |p|^2
    const bool ok = (n2 < 1e300) && (fabs(w) < 1e150);
    const double k2 = ok ? ((variant & DFEPE_W8PT_NO_ROWNORM) ? (w * w) : (w * w) * fast_rcp(fmax(n2, 1e-24))) : 0.0;                        // (w / max(|p|,1e-12))^2
    const double aa[6] = {a0 * a0, a0 * a1, a0 * a2, a1 * a1, a1 * a2, a2 * a2};
    const double bb[6] = {k2 * b0 * b0, k2 * b0 * b1, k2 * b0, k2 * b1 * b1, k2 * b1, k2};
    if (ok) {
#pragma unroll
      for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int v = 0; v < 6; ++v) acc[6 * u + v] += bb[u] * aa[v];
    }
  }

  // ---- phase 3: reduce-scatter across the wave; lane ends up owning (at most) one distinct sum -------------
  halve<36, 0>(acc, lane & 32, 32);
  halve<18, 0>(acc, lane & 16, 16);
  halve<9, 0x140>(acc, lane & 8, 8);   // row_mirror
  halve<5, 0x141>(acc, lane & 4, 4);   // row_half_mirror
  halve<3, 0x4E>(acc, lane & 2, 2);    // quad_perm [2,3,0,1]
  halve<2, 0xB1>(acc, lane & 1, 1);    // quad_perm [1,0,3,2]
  {
    // mirror the fixed halving schedule 36 -> 18 -> 9 -> 5 -> 3 -> 2 -> 1: `idx` is the distinct sum this lane
    // ends up owning, `cnt` how many of its slots were real data (<= 0: the lane holds padding)
    int cnt = 36, idx = 0, width = 36;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const int h = (width + 1) / 2;
      if (lane & m) { idx += h; cnt -= h; } else { cnt = (cnt < h) ? cnt : h; }
      width = h;
    }
    if (COOP) {  // the four wavefronts' partial sums meet in LDS; wavefront 0 adds them up
      if (cnt >= 1) RED[24 + 36 * wave + idx] = acc[0];
      __syncthreads();
      if (cnt >= 1) {
        double tot = RED[24 + idx];
#pragma unroll
        for (int k = 1; k < WPP; ++k) tot += RED[24 + 36 * k + idx];
        acc[0] = tot;
      }
    }
    if (cnt >= 1 && (!COOP || wave == 0)) {
      // sum (u,v) is M[3r+c][3r'+c'] for (r,r') = sym pair u, (c,c') = sym pair v, and its 3 index swaps
      const int u = idx / 6, v = idx % 6;
      const int r0 = kSymR[u], r1 = kSymC[u], q0 = kSymR[v], q1 = kSymC[v];
      const double val = acc[0];
      const int i00 = 3 * r0 + q0, i01 = 3 * r0 + q1, i10 = 3 * r1 + q0, i11 = 3 * r1 + q1;
      M64[i00 * 9 + i11] = val; M64[i11 * 9 + i00] = val;
      M64[i01 * 9 + i10] = val; M64[i10 * 9 + i01] = val;
    }
  }
  wave_sync();

  double f[9];   // the solver's eigenvector (unit norm, oriented) and the de-normalised rank-2 F: produced by phases 4-5,
  float of[9];   // consumed by the per-correspondence outputs of phase 6
  if (!COOP || wave == 0) {
  // ---- phase 4a: fp32 Jacobi on M / trace(M) ----------------------------------------------------------
  // A' = J^T A J, V' = V J with J_pp = J_qq = c, J_pq = s, J_qp = -s for the pairs (p,q) = (0,1),(2,3),(4,5),(6,7)
  // of *positions*; position 8 sits out.  Per position k we keep (c_k, sh_k), sh_p = -s, sh_q = +s, so that
  // col_k' = c_k col_k + sh_k col_(k^1).  The result of the round is stored through the seat permutation (make_jac_table),
  // which realises the round-robin schedule (9 rounds = all 36 pairs once).  Eigenpairs come out in seat order,
  // which is irrelevant: (A32[k][k], V32[:,k]) is a consistent pair for every k.
  double tr = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) tr += M64[k * 10];
  const double inv_tr = (tr > 0.0) ? fast_rcp(tr) : 1.0;
  // A32/V32 are stored with a row stride of 10 floats so that the column pair (2m, 2m+1) of any row is one aligned
  // 8-byte word: one ds_read_b64 fetches an element together with its rotation partner.
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = lane + 64 * k;
    if (e < 90) {
      const int i = e / 10, j = e % 10;
      A32[e] = (j < 9) ? (float)(M64[i * 9 + j] * inv_tr) : 0.0f;
      V32[e] = (i == j) ? 1.0f : 0.0f;
    }
  }
  if (lane == 8) CS[8] = make_float4(1.0f, 0.0f, 0.0f, 1.0f);  // position 8 sits out: identity rotation
  // Work items of a round.  Slot 1: V elements 0..63 (one per lane).  Slot 2: the 45 upper-triangular A elements on
  // lanes 0..44 and the 17 remaining V elements on lanes 45..61 -- a V element is the special case (c_i, s_i) = (1, 0)
  // of the two-sided update, so both kinds run the same instruction stream.  All addresses are loop-invariant.
  const uint4 jw = *reinterpret_cast<const uint4*>(&kJac.l[lane]);
  const int rd_own = jw.x & 0xffff, rd_par = jw.x >> 16, wr2 = jw.y & 0xffff, wr2t = jw.y >> 16;
  const int rd_v = jw.z & 0xffff, wr_v = jw.z >> 16;
  const int ci_off = jw.w & 0xff, cj_off = (jw.w >> 8) & 0xff, c0_off = (jw.w >> 16) & 0xff;  // float offsets into CS
  const unsigned jfl = jw.w >> 24;
  const bool is_a = (jfl & kJacIsA) != 0, is_v2 = (jfl & kJacIsV2) != 0, odd2 = (jfl & kJacOdd2) != 0;
  const bool offdiag = (jfl & kJacOffDiag) != 0;
  const int off_idx = rd_own + (odd2 ? 1 : 0);           // A32[ti][tj] for the A items
  const int pp = (lane < 16) ? (lane & 6) : 0;  // lanes 0..15: my pair is positions (pp, pp+1); lanes 8..15 repeat 0..7 and
                                                // store the swapped copy (one ds_write_b64 per lane instead of a ds_write_b128)
  wave_sync();
  for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
    float off = 0.0f;
    if (offdiag) {
      const float a = A32[off_idx];
      off = a * a;
    }
    off = wave_sum(off);
    if (!(off > kJacobiTol)) break;  // wave-uniform (also leaves on NaN)
    for (int r = 0; r < 9; ++r) {
      // every read of the round is issued up front (none depends on this round's rotations): 3 x ds_read_b64 + the
      // rotation inputs of lanes 0..15, then the (c, sh) exchange through CS (3 x ds_read_b64), then 2-3 ds_write_b32
      const float2 own = *reinterpret_cast<const float2*>(A32 + rd_own);
      const float2 par = *reinterpret_cast<const float2*>(A32 + rd_par);
      const float2 vv = *reinterpret_cast<const float2*>(V32 + rd_v);
      if (lane < 16) {
        const float2 pq = *reinterpret_cast<const float2*>(A32 + pp * 11);  // {A[p][p], A[p][p+1]}
        const float aqq = A32[pp * 11 + 11];
        // small-angle Jacobi rotation from two reciprocal square roots: cos 2t = |d| / h, c = sqrt((1 + cos 2t) / 2),
        // s = sgn(d) b / (2 h c)   (same rotation as t = b / (d + sgn(d) h), |t| <= 1, five instructions shorter)
        const float d = aqq - pq.x, b = 2.0f * pq.y;
        const float r = __builtin_amdgcn_rsqf(fmaf(d, d, b * b));
        const float x = fmaf(0.5f * fabsf(d), r, 0.5f);
        const float y = __builtin_amdgcn_rsqf(x);
        float c = x * y;
        float sn = copysignf(0.5f, d) * b * r * y;
        if (pq.y == 0.0f) { c = 1.0f; sn = 0.0f; }  // also catches 0/0
        const float sh = (lane & 1) ? sn : -sn;
        reinterpret_cast<float2*>(CS)[2 * (lane & 7) + (lane >> 3)] = (lane < 8) ? make_float2(c, sh) : make_float2(sh, c);
      }
      wave_sync();
      // col' = c col + sh col_partner.  A lane whose column is the odd one of its pair holds (partner, self) in its
      // float2, so it reads the coefficients in swapped order (.zw): no per-element selects in the loop.
      const float* CSf = reinterpret_cast<const float*>(CS);
      const float2 ci = *reinterpret_cast<const float2*>(CSf + ci_off);
      const float2 kj = *reinterpret_cast<const float2*>(CSf + cj_off);
      const float2 k0 = *reinterpret_cast<const float2*>(CSf + c0_off);
      const float new2 = ci.x * fmaf(kj.x, own.x, kj.y * own.y) + ci.y * fmaf(kj.x, par.x, kj.y * par.y);
      const float v0 = fmaf(k0.x, vv.x, k0.y * vv.y);
      if (is_a || is_v2) A32[wr2] = new2;
      if (is_a) A32[wr2t] = new2;
      V32[wr_v] = v0;
      wave_sync();
    }
  }

  // ---- phase 4b: pick the eigenpair the reference picks, polish it in fp64 ----------------------------
  // torch.svd(X)[2][:, -1] is the right singular vector of the smallest of the min(N,9) singular values
  // (DeepFNet.py:232-233): for N >= 9 the smallest eigenvalue of X^T X; for N < 9 the reduced SVD has only N
  // columns, so the reference takes the smallest of the N *non-null* directions.  `skip` = 9 - min(N,9)
  // eigenvalues are passed over (ascending order, index as tie-break) to mirror that.
  float lam32[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) lam32[k] = A32[k * 11];
  if (lane < 9) LAMC[lane] = (double)A32[lane * 11] * tr;
  const int skip = (N >= 9) ? 0 : 9 - N;
  int kmin = 0;
  // provisional choice from the fp32 eigenvalues: rank `skip` in ascending order (index as tie-break)
  if (N >= 9) {  // the usual case: plain arg-min (first index on ties)
    float lmin = lam32[0];
#pragma unroll
    for (int k = 1; k < 9; ++k)
      if (lam32[k] < lmin) { lmin = lam32[k]; kmin = k; }
  } else {       // wave-uniform branch: rank selection, skipping the 9 - N null directions
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      int rank = 0;
#pragma unroll
      for (int j = 0; j < 9; ++j) rank += (lam32[j] < lam32[k] || (lam32[j] == lam32[k] && j < k)) ? 1 : 0;
      if (rank == skip) kmin = k;
    }
  }
  // The fp32 sweeps ran on X^T X, so eigenvalues closer than ~1e-6 trace -- singular values of X below ~1e-3 sigma_1,
  // which LAPACK's SVD of X (the reference) still tells apart -- come out in arbitrary order with arbitrarily mixed
  // vectors.  Members of such a cluster around the selected eigenvalue are re-resolved by Rayleigh-Ritz in fp64 against the
  // exact M64: cyclic Jacobi on H = Qc^T M Qc, rotating the stored (fp32) vectors; their Ritz values are good to
  // ~1e-14 trace, i.e. the resolution of an fp32 SVD of X.  Rare (peaked weights, near-minimal or near-planar sets);
  // the usual case is a cluster of one and costs two dozen instructions.
  float lsel = lam32[0];
#pragma unroll
  for (int k = 1; k < 9; ++k) lsel = (k == kmin) ? lam32[k] : lsel;
  unsigned cmask = 0;
  int below = 0;  // eigenvalues clearly below the cluster (only possible for N < 9)
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const bool in = fabsf(lam32[k] - lsel) < kClusterTol;
    cmask |= in ? (1u << k) : 0u;
    below += (!in && lam32[k] < lsel) ? 1 : 0;
  }
  wave_sync();
  if (__popc(cmask) > 1) {  // wave-uniform
    const double floor_h = 1e-16 * tr;
    // purify the cluster vectors first: remove what the fp32 sweeps left in them of the directions OUTSIDE the cluster
    // (one residual-correction step against the well-separated eigenpairs; afterwards only fp32 storage error remains)
    for (int i = 0; i < 9; ++i) {
      if (!((cmask >> i) & 1u)) continue;
      if (lane < 9) {
        double yi = 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c) yi += M64[lane * 9 + c] * (double)V32[c * 10 + i];
        SCR[lane] = yi;
      }
      wave_sync();
      double hii = 0.0, rr[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) hii += (double)V32[c * 10 + i] * SCR[c];
#pragma unroll
      for (int c = 0; c < 9; ++c) rr[c] = SCR[c] - hii * (double)V32[c * 10 + i];
      if (lane < 9) {
        double dot = 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c) dot += (double)V32[c * 10 + lane] * rr[c];
        const double den = hii - LAMC[lane];
        SCR[16 + lane] = (((cmask >> lane) & 1u) || !(fabs(den) > 1e-7 * tr)) ? 0.0 : dot * fast_rcp(den);
      }
      wave_sync();
      if (lane < 9) {
        double dsum = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) dsum += SCR[16 + k] * (double)V32[lane * 10 + k];
        SCR[lane] = (double)V32[lane * 10 + i] + dsum;
      }
      wave_sync();
      double nn = 0.0;
#pragma unroll
      for (int c = 0; c < 9; ++c) nn += SCR[c] * SCR[c];
      const double inn = fast_rsqrt(nn);
      if (lane < 9) V32[lane * 10 + i] = (float)(SCR[lane] * inn);
      wave_sync();
    }
    for (int sw = 0; sw < 6; ++sw) {
      bool rotated = false;
      for (int i = 0; i < 8; ++i) {
        if (!((cmask >> i) & 1u)) continue;
        for (int j = i + 1; j < 9; ++j) {
          if (!((cmask >> j) & 1u)) continue;
          // y_i = M q_i, y_j = M q_j: one row per lane
          if (lane < 9) {
            double yi = 0.0, yj = 0.0;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
              const double mc = M64[lane * 9 + c];
              yi += mc * (double)V32[c * 10 + i];
              yj += mc * (double)V32[c * 10 + j];
            }
            SCR[lane] = yi;
            SCR[16 + lane] = yj;
          }
          wave_sync();
          double hii = 0.0, hjj = 0.0, hij = 0.0;
#pragma unroll
          for (int c = 0; c < 9; ++c) {
            const double qi = (double)V32[c * 10 + i], qj = (double)V32[c * 10 + j];
            hii += qi * SCR[c];
            hjj += qj * SCR[16 + c];
            hij += qi * SCR[16 + c];
          }
          const double dlt = hjj - hii;
          if (fabs(hij) > floor_h && fabs(hij) > 1e-7 * fabs(dlt)) {
            const double tau = 0.5 * dlt * fast_rcp(hij);
            const double t = ((tau >= 0.0) ? 1.0 : -1.0) * fast_rcp(fabs(tau) + fast_sqrt(1.0 + tau * tau));
            const double cs = fast_rsqrt(1.0 + t * t), sn = t * cs;
            if (lane < 9) {
              const double qi = (double)V32[lane * 10 + i], qj = (double)V32[lane * 10 + j];
              V32[lane * 10 + i] = (float)(cs * qi - sn * qj);
              V32[lane * 10 + j] = (float)(sn * qi + cs * qj);
            }
            hii -= t * hij;
            hjj += t * hij;
            rotated = true;
          }
          if (lane == 0) { LAMC[i] = hii; LAMC[j] = hjj; }
          wave_sync();
        }
      }
      if (!rotated || __popc(cmask) == 2) break;  // one rotation diagonalises a 2 x 2 exactly
    }
    // final choice inside the cluster: rank (skip - below) among its members by the fp64 Ritz values
    const int want = skip - below;
    int pick = kmin;
    for (int k = 0; k < 9; ++k) {
      if (!((cmask >> k) & 1u)) continue;
      int rank = 0;
      const double lk = LAMC[k];
      for (int j = 0; j < 9; ++j) {
        if (!((cmask >> j) & 1u)) continue;
        const double lj = LAMC[j];
        rank += (lj < lk || (lj == lk && j < k)) ? 1 : 0;
      }
      if (rank == want) pick = k;
    }
    kmin = pick;
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) f[c] = (double)V32[c * 10 + kmin];
  double rho = LAMC[kmin];
  // residual correction: r = M f - rho f;  f += sum_{k != kmin} q_k (q_k . r) / (rho - lam_k);  renormalise.
  // The Jacobi basis (fp32-accurate) acts as an approximate inverse of (M - rho); the fixed point is the exact
  // fp64 eigenvector, reached at a linear rate ~ eps32 |M| / gap per iteration.
  double rn2_prev = 0.0;
  bool last_pass = false;
  for (int it = 0; it < kRefineIters; ++it) {
    double fn2 = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) fn2 += f[c] * f[c];
    const double fin = fast_rsqrt(fn2);
#pragma unroll
    for (int c = 0; c < 9; ++c) f[c] *= fin;
    // y = M f, one row per lane
    if (lane < 9) {
      double y = 0.0;
#pragma unroll
      for (int c = 0; c < 9; ++c) y += M64[lane * 9 + c] * f[c];
      SCR[lane] = y;
    }
    wave_sync();
    double r[9];
    rho = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) { r[c] = SCR[c]; rho += r[c] * f[c]; }
    double rn2 = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) { r[c] -= rho * f[c]; rn2 += r[c] * r[c]; }
    const double rn2_tol = 1e-28 * tr * tr;
    if (!(rn2 > rn2_tol)) break;  // |M f - rho f| <= 1e-14 trace(M): converged (wave-uniform)
    // the iteration contracts linearly: when the last step's factor, applied once more, lands 100x below the
    // tolerance, this correction is the final one and the verification pass after it is skipped
    last_pass = (it > 0) && (rn2 * rn2 < 1e-2 * rn2_tol * rn2_prev);
    rn2_prev = rn2;
    // a_k = (q_k . r) / (rho - lam_k) for k != kmin, one k per lane
    if (lane < 9) {
      double dot = 0.0;
#pragma unroll
      for (int c = 0; c < 9; ++c) dot += (double)V32[c * 10 + lane] * r[c];
      double den = rho - LAMC[lane];
      const double lim = 1e-12 * tr;
      if (fabs(den) < lim) den = (den < 0.0) ? -lim : lim;
      SCR[16 + lane] = (lane == kmin) ? 0.0 : dot * fast_rcp(den);
    }
    wave_sync();
    // d_c = sum_k a_k q_k[c], one component per lane; then everybody reads the new f
    if (lane < 9) {
      double dsum = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) dsum += SCR[16 + k] * (double)V32[lane * 10 + k];
      SCR[lane] = dsum;
    }
    wave_sync();
#pragma unroll
    for (int c = 0; c < 9; ++c) f[c] += SCR[c];
    wave_sync();
    if (last_pass) break;
  }

  // ---- phase 5: rank-2 projection, de-normalisation (wave-uniform arithmetic) ------------------------------
  double fn2 = 0.0;
#pragma unroll
  for (int c = 0; c < 9; ++c) fn2 += f[c] * f[c];
  // orientation: largest-magnitude component positive (first one on ties)
  double big = f[0];
#pragma unroll
  for (int c = 1; c < 9; ++c)
    if (fabs(f[c]) > fabs(big)) big = f[c];
  const double sgn = (big < 0.0) ? -1.0 : 1.0;
  const double fscale = sgn * fast_rsqrt(fn2);
#pragma unroll
  for (int c = 0; c < 9; ++c) f[c] *= fscale;

  float Ff[9], U3[9], S3[3], V3[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) Ff[c] = (float)f[c];
  svd3_fast(Ff, U3, S3, V3);
  // s3 = u3^T F v3 in fp64 (stationary w.r.t. first-order errors of u3, v3), F' = F - s3 u3 v3^T
  double s3 = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) s3 += (double)U3[3 * r + 2] * f[3 * r + c] * (double)V3[3 * c + 2];
  double Fp[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Fp[3 * r + c] = f[3 * r + c] - s3 * (double)U3[3 * r + 2] * (double)V3[3 * c + 2];
  if (variant & DFEPE_W8PT_FORCE_110) {  // E' = U diag(1,1,0) V^T = u1 v1^T + u2 v2^T  (utils_F.py:148-149)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        Fp[3 * r + c] = (double)U3[3 * r] * (double)V3[3 * c] + (double)U3[3 * r + 1] * (double)V3[3 * c + 1];
  }
  // out = T2^T F' T1,  T = [[s,0,-s cx],[0,s,-s cy],[0,0,1]]
  double Mx[9], out[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    Mx[3 * r + 0] = s1 * Fp[3 * r + 0];
    Mx[3 * r + 1] = s1 * Fp[3 * r + 1];
    Mx[3 * r + 2] = Fp[3 * r + 2] - s1 * (c1x * Fp[3 * r + 0] + c1y * Fp[3 * r + 1]);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    out[c] = s2 * Mx[c];
    out[3 + c] = s2 * Mx[3 + c];
    out[6 + c] = Mx[6 + c] - s2 * (c2x * Mx[c] + c2y * Mx[3 + c]);
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) of[c] = (float)out[c];
  if (lane == 0) {
    float* dst = F_out + (size_t)pair * 9;
#pragma unroll
    for (int c = 0; c < 9; ++c) dst[c] = of[c];
  }
  if (save != nullptr) {
    float* sv = save + (size_t)pair * DFEPE_SAVE_FLOATS;
    // the polished, oriented f replaces its Jacobi column in the record; staged through LDS so that no register
    // array is indexed dynamically (that would push it to scratch)
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 9; ++c) SCR[c] = sgn * f[c];
    }
    wave_sync();
    {
      const int k = lane / 9, c = lane % 9;
      sv[SV_Q + lane] = (k == kmin) ? (float)SCR[c] : V32[c * 10 + k];
    }
    if (lane < 17) {
      const int e = lane + 64, k = e / 9, c = e % 9;
      sv[SV_Q + e] = (k == kmin) ? (float)SCR[c] : V32[c * 10 + k];
    }
    if (lane < 9) sv[SV_LAM + lane] = (lane == kmin) ? (float)rho : (float)LAMC[lane];
    if (lane == 0) {
      sv[SV_T1 + 0] = (float)s1; sv[SV_T1 + 1] = (float)c1x; sv[SV_T1 + 2] = (float)c1y;
      sv[SV_T2 + 0] = (float)s2; sv[SV_T2 + 1] = (float)c2x; sv[SV_T2 + 2] = (float)c2y;
      sv[SV_KMIN] = (float)kmin;
      sv[SV_SIGN] = (float)sgn;
#pragma unroll
      for (int c = 0; c < 9; ++c) { sv[SV_U3 + c] = U3[c]; sv[SV_V3 + c] = V3[c]; }
      sv[SV_S3 + 0] = S3[0]; sv[SV_S3 + 1] = S3[1]; sv[SV_S3 + 2] = (float)s3;
      sv[127] = 64.0f;  // record format tag: wavefront-per-pair kernel (the row-per-pair kernels write 16)
    }
  }

  if (COOP && lane == 0) {
#pragma unroll
    for (int c = 0; c < 9; ++c) { RED[168 + c] = f[c]; REDF[8 + c] = of[c]; }
  }
  }  // eigen phases (wavefront 0 of a cooperative workgroup)
  if (COOP) {
    __syncthreads();
    if (wave != 0) {
#pragma unroll
      for (int c = 0; c < 9; ++c) { f[c] = RED[168 + c]; of[c] = REDF[8 + c]; }
    }
  }

  // ---- phase 6: per-correspondence outputs ----------------------------------------------------------
  float* rdst = residual + (size_t)pair * N;
  float* edst = (epi_res != nullptr) ? epi_res + (size_t)pair * N : nullptr;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
  for (int i = tid; i < N; i += NT) {
    const Pt p = lds_point<RAW>(P, i, npad);
    double ph[9];
    const double w = (double)W[i];
    const bool ok = unit_row(p, s1, c1x, c1y, s2, c2x, c2y, ph) && (fabs(w) < 1e150);
    double r = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) r += ph[c] * f[c];
    r = ok ? r * w : 0.0;
    rdst[i] = (float)r;
    if (edst != nullptr) {
      // l1 = F^T x2 (row form x2 F), l2 = F x1, dd = x2^T F x1 = x1 . l1     (utils_F.py:402-411), fp32 like the reference
      const float l1x = fmaf(p.x2, of[0], fmaf(p.y2, of[3], p.z2 * of[6]));
      const float l1y = fmaf(p.x2, of[1], fmaf(p.y2, of[4], p.z2 * of[7]));
      const float l1z = fmaf(p.x2, of[2], fmaf(p.y2, of[5], p.z2 * of[8]));
      const float l2x = fmaf(p.x1, of[0], fmaf(p.y1, of[1], p.z1 * of[2]));
      const float l2y = fmaf(p.x1, of[3], fmaf(p.y1, of[4], p.z1 * of[5]));
      const float dd = fmaf(p.x1, l1x, fmaf(p.y1, l1y, p.z1 * l1z));
      const float n1 = sqrtf(fmaf(l1x, l1x, l1y * l1y)) + 1e-6f;
      const float n2 = sqrtf(fmaf(l2x, l2x, l2y * l2y)) + 1e-6f;
      const float d = fabsf(dd) * (1.0f / n1 + 1.0f / n2);
      edst[i] = fminf(d, clamp_at);
    }
  }
}

}  // namespace

// host-side launcher ------------------------------------------------------------------------------------
extern "C" int dfepe_w8pt_fwd(const float* pts1, const float* pts2, const float* weights, int B, int N,
                              int n_weight_sets, unsigned flags, float image_w, float image_h, float clamp_at, float* F_out,
                              float* residual, float* epi_res, float* save, float* weights_out, void* stream) {
  const bool raw = (flags & DFEPE_W8PT_RAW_MATCHES) != 0;
  const int logits_mode = (flags & DFEPE_W8PT_LOGITS) ? 1 : 0;
  const unsigned variant = flags & (DFEPE_W8PT_SQRT2 | DFEPE_W8PT_NO_ROWNORM | DFEPE_W8PT_FORCE_110 | DFEPE_W8PT_NO_HARTLEY);
  if (B < 0 || N <= 0 || n_weight_sets < 1) return DFEPE_ERR_INVALID_ARG;
  // the textbook variants are forward-only; un-normalised rows alone (Fit(normalize_SVD=False)) have a backward in the row kernels
  if (variant && save && !(variant == DFEPE_W8PT_NO_ROWNORM && dfepe_w8pt_use_rows(N, (long long)B * n_weight_sets, flags)))
    return DFEPE_ERR_UNSUPPORTED;
  if (B == 0) return DFEPE_OK;
  if (!pts1 || (!raw && !pts2) || !weights || !F_out || !residual) return DFEPE_ERR_INVALID_ARG;
  if (raw && !(image_w > 0.f && image_h > 0.f)) return DFEPE_ERR_INVALID_ARG;
  if (raw && (reinterpret_cast<uintptr_t>(pts1) & 15u)) return DFEPE_ERR_INVALID_ARG;  // float4 loads
  if (reinterpret_cast<uintptr_t>(save) & 15u) return DFEPE_ERR_INVALID_ARG;          // wide stores of the record

  if (flags & ~DFEPE_W8PT_ALL_FLAGS) return DFEPE_ERR_INVALID_ARG;  // unknown flag bits are rejected, not ignored
  if (dfepe_w8pt_use_rows(N, (long long)B * n_weight_sets, flags)) {
    // one 16-lane row per pair, fp64 tridiagonal eigen-solver (w8pt16.hip)
    W8Args A;
    A.pts1 = pts1; A.pts2 = pts2; A.wts = weights;
    A.Bm = B; A.B = B * n_weight_sets; A.N = N;
    A.hw_sx = raw ? 2.0f / image_w : 0.f; A.hw_sy = raw ? 2.0f / image_h : 0.f; A.clamp_at = clamp_at;
    A.F_out = F_out; A.residual = residual; A.epi_res = epi_res; A.save = save; A.weights_out = weights_out;
    A.logits_mode = logits_mode; A.variant = variant;
    return dfepe_w8pt16_fwd_launch(A, raw, static_cast<hipStream_t>(stream));
  }
  const int npad = (N + 3) & ~3;
  const int wave_bytes = kWsDoubles * (int)sizeof(double) + (raw ? 5 : 7) * npad * (int)sizeof(float);
  // waves per block: 4 when at least 16 wavefronts fit a CU's 160 KiB anyway, otherwise whichever of {4,2,1} keeps the
  // most wavefronts resident (at N = 1000 a pair needs 21.7 KB: 1-wave blocks give 7 per CU, 4-wave blocks only 4)
  const int lds_cap = 160 * 1024;
  if (wave_bytes > lds_cap) return DFEPE_ERR_UNSUPPORTED;  // N > ~8000 (raw) / ~5800 (pts): not staged in LDS yet
  int waves = 4, best = 0;
  for (int wv = 4; wv >= 1; wv >>= 1) {
    const int resident = (lds_cap / (wv * wave_bytes)) * wv;
    if (resident >= 16) { waves = wv; break; }
    if (resident > best) { best = resident; waves = wv; }
  }
  // Large N: the staging area limits the per-wavefront variant to few wavefronts per CU (7 at N = 1000, 3 at N = 2000),
  // each of which walks all N correspondences four times.  A cooperative workgroup per pair shares those walks between
  // 2 or 4 wavefronts: the smallest count that puts >= 12 wavefronts on a CU (measured, B = 4096: N = 768 96 -> 89 us,
  // N = 1000 126 -> 110 us, N = 2000 336 -> 182 us), and 4 whenever the batch fits in one residency round of cooperative
  // workgroups (1024 = 256 CUs x 4; only latency matters then: N = 1000, B = 512: 37 -> 28 us).
  const unsigned force_wpp = (flags & DFEPE_W8PT_WAVE_PER_PAIR) ? 1u : 0u;  // one wavefront per pair, never a cooperative workgroup
  const bool can_coop = (N >= 256) && (wave_bytes + kCoopBytes <= lds_cap);
  const int blocks_coop = can_coop ? lds_cap / (wave_bytes + kCoopBytes) : 0;
  int wpp = 1;
  if (can_coop && lds_cap / wave_bytes < 12) wpp = (2 * blocks_coop >= 12) ? 2 : 4;
  if (can_coop && lds_cap / wave_bytes < 16 && (long long)B * n_weight_sets <= 1024) wpp = 4;
  if (force_wpp) wpp = (force_wpp == 1) ? 1 : ((force_wpp == 2) ? 2 : 4);
  if (wpp > 1 && wave_bytes + kCoopBytes > lds_cap) wpp = 1;
  const bool coop = wpp > 1;
  if (coop) waves = wpp;
  const size_t lds = coop ? (size_t)wave_bytes + kCoopBytes : (size_t)waves * wave_bytes;
  const int Bm = B;
  B *= n_weight_sets;  // one wavefront (or cooperative workgroup) per (weight set, pair)
  const dim3 grid(coop ? B : (B + waves - 1) / waves), block(64 * waves);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float hw_sx = raw ? 2.0f / image_w : 0.f, hw_sy = raw ? 2.0f / image_h : 0.f;
#define DFEPE_LAUNCH_FWD(R, C)                                                                                              \
  do {                                                                                                                      \
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&w8pt_fwd_kernel<R, C>),                      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)          \
      return DFEPE_ERR_HIP;                                                                                                 \
    hipLaunchKernelGGL((w8pt_fwd_kernel<R, C>), grid, block, lds, st, pts1, pts2, weights, B, Bm, N, npad, wave_bytes, hw_sx, \
                       hw_sy, clamp_at, F_out, residual, epi_res, save, weights_out, logits_mode, variant);            \
  } while (0)
  if (raw) {
    if (wpp == 4) DFEPE_LAUNCH_FWD(true, 4); else if (wpp == 2) DFEPE_LAUNCH_FWD(true, 2); else DFEPE_LAUNCH_FWD(true, 1);
  } else {
    if (wpp == 4) DFEPE_LAUNCH_FWD(false, 4); else if (wpp == 2) DFEPE_LAUNCH_FWD(false, 2); else DFEPE_LAUNCH_FWD(false, 1);
  }
#undef DFEPE_LAUNCH_FWD
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
