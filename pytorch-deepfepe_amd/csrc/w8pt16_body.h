// w8pt16 -- weighted normalised 8-point fit with ONE 16-LANE ROW PER IMAGE PAIR (four pairs per wavefront), any N.
//
// Arithmetic contract:
//   NormalizeAndExpand_HW   deepFEPE/models/DeepFNet.py:93-120   (fused when RAW)
//   Fit.normalize           deepFEPE/models/DeepFNet.py:148-179  (Hartley, unit weights, literal 1.4142)
//   Fit.weighted_svd        deepFEPE/models/DeepFNet.py:181-257
//   compute_epi_residual    deepFEPE/dsac_tools/utils_F.py:400-413
// restated from SURVEY.md Appendix A, not from the reference source.
//
// Why a row: with a wavefront per pair, everything that is uniform over the pair (selection, rank-2 step, Hartley
// bookkeeping, every scalar recurrence) was issued for 64 lanes, and the 9x9 eigenproblem went through ~41 dependent LDS
// round trips of a systolic Jacobi.  Here
//   * the pair's (at most 128) correspondences stay in the registers of its 16 lanes for the whole kernel: one HBM read,
//     no LDS staging;
//   * the 36 distinct fp64 moment sums are reduce-scattered inside the row with DPP (35 exchanges);
//   * the solver needs ONE eigenpair of M = X^T X, so the Jacobi is replaced by an fp64 tridiagonal route with no
//     iteration that depends on the data's conditioning:
//       Householder tridiagonalisation (lane i holds row i of M; broadcasts are row_newbcast DPP moves),
//       the (skip+1)-th smallest eigenvalue by 16-way multisection on the division-free Sturm sequence (every lane of the
//       row probes its own shift: a log-spaced round finds the magnitude, ~9 linear rounds of 17x each finish it),
//       its eigenvector by twisted factorisation, back-transformed through the seven reflectors;
//     everything is fp64 (full-rate on gfx950), so there is no fp32 sweep to polish and no cluster repair;
//   * the backward (w8pt16_bwd_pair) applies (M - lam I)^+ through the same tridiagonal form saved here;
//   * every global load is issued at the top, unconditionally (clamped indices), and every store at the bottom: a load under
//     a condition costs a dependent memory round trip, and a consumer scheduled after a store waits for the STORE (in-order
//     memory counter).  The kernel is bound by instruction issue (~5 cycles per vector instruction), not by latency.
// The bodies are written against rowgroup.h only, so tests/emu/ runs them on the host against the oracle.
#pragma once
#include <type_traits>

#include "dfepe.h"
#include <rowgroup.h>  // angle brackets on purpose: tests/emu/ substitutes its host emulation through the include path
#include "dfepe_math.h"

// ---- layout of the per-pair `save` record written by w8pt16_fwd_pair (floats; DFEPE_SAVE_FLOATS = 128) ----------
#define S16_T1 0       // Hartley transform of image 1: s, cx, cy
#define S16_T2 3       // Hartley transform of image 2
#define S16_F 6        // 9: the oriented unit eigenvector f (largest-magnitude component positive)
#define S16_Z 15       // 9: z = H^T f, the eigenvector of the tridiagonal T (same orientation)
#define S16_HV 26      // 35: reflector k = 0..6, components k+1..8, at offset 8k - k(k-1)/2  (floats 24..63: nothing else, see below)
#define S16_TD 68      // 9 doubles: diagonal of T  (T = H^T (M / trace M) H)
#define S16_TE 86      // 8 doubles: off-diagonal of T
#define S16_LAM 102    // 1 double: the selected eigenvalue of M / trace M
#define S16_U3 104     // smallest singular triplet of F = reshape(f): u3 (3 floats)
#define S16_V3 107     //                                              v3 (3 floats)
#define S16_S3 110     //                                              s3 >= 0
#define S16_TWIST 111  // twist index of the factorisation (largest-residual-free row), as float
#define S16_HB 112     // 7: beta_k  (H_k = I - beta_k v_k v_k^T)
#define S16_INVTR 125  // 1 / trace(M)
#define S16_SCRATCH 126 // unused since round 3 (kept zero); float 24 takes the branch-free stray stores of the reflector lanes
#define S16_TAG 127    // record tag: the backward poisons its output when handed anything but a record of this layout
#define S16_TAG_VALUE 17.0f  // 17: the round-3 layout (TWIST and HB behind S3)

__host__ __device__ constexpr int s16_hv_off(int k) { return 8 * k - (k * (k - 1)) / 2; }

struct W8Args {
  const float* pts1;
  const float* pts2;
  const float* wts;
  int B, Bm, N;
  float hw_sx, hw_sy, clamp_at;
  float* F_out;
  float* residual;
  float* epi_res;
  float* save;
  float* weights_out;
  int logits_mode;
  unsigned variant;
  bool row_per_pair;  // host side only: never the cooperative workgroup (DFEPE_W8PT_ROW_PER_PAIR)
};

// phase markers for scripts/isa_phases.py (hipcc -DDFEPE_ISA_MARKS -S): comments in the assembly, nothing otherwise;
// -DDFEPE_PHASE_CLOCKS (scripts/ubench/fit_phases.hip only): lane 0 of every wavefront stamps the shader clock at each marker
#if defined(DFEPE_ISA_MARKS)
#define DFEPE_MARK(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; MARK " name); __builtin_amdgcn_sched_barrier(0); } while (0)
#elif defined(DFEPE_PHASE_CLOCKS)
constexpr int dfepe_phase_id(const char* s) {  // P0 P1 P2 P3 P4 P4b P4c P4d P5 P5b P5s P6 Pend -> 0..12; the backward's B0 .. B7 -> 0..7
  return (s[0] == 'B') ? (s[1] - '0') : (s[1] == '0') ? 0 : (s[1] == '1') ? 1 : (s[1] == '2') ? 2 : (s[1] == '3') ? 3
       : (s[1] == '4') ? ((s[2] == 0) ? 4 : (s[2] == 'b') ? 5 : (s[2] == 'c') ? 6 : 7)
       : (s[1] == '5') ? ((s[2] == 0) ? 8 : (s[2] == 'b') ? 9 : 10) : (s[1] == '6') ? 11 : 12;
}
extern __device__ unsigned long long* g_dfepe_phase_clk;  // [wavefront][16]
#define DFEPE_MARK(name)                                                                                              \
  do {                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    const unsigned long long t_ = __builtin_readcyclecounter();                                                       \
    if ((threadIdx.x & 63u) == 0u) g_dfepe_phase_clk[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + dfepe_phase_id(name)] = t_; \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  } while (0)
#else
#define DFEPE_MARK(name)
#endif

#ifndef DFEPE_P6_FAST
#define DFEPE_P6_FAST 1  // the output phase of the registers-resident kernels: unguarded stores where the guards are known to hold (round 6)
#endif

template <int I, int E, class Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
  if constexpr (I < E) {
    fn(std::integral_constant<int, I>{});
    static_for<I + 1, E>(fn);
  }
}

// One correspondence as a phase sees it: coordinates (zeros when dropped or padding), weight, existence, membership in X.
struct PRec {
  Pt p;
  float w;   // weight in X (0 when dropped or padding)
  float ws;  // softmax weight before the drop mask (logits mode, looped kernel)
  bool valid, keep;
};

// What one correspondence costs in global loads: RAW: (x1,y1,x2,y2) as one float4 + one weight-like float; otherwise two
// homogeneous points + the weight.  v[7] is spare (the backward parks a provisional gradient there).
struct RawRec {
  float v[8];
};

// Loop over a lane's correspondences.  IT > 0 (N <= 128, they live in registers): fully unrolled.  IT = 0 (any N): chunks of
// eight, double-buffered on the RAW loads -- load(it) only issues global loads (index clamped: it may lie past the end),
// decode(it, raw) does all dependent arithmetic.  The next chunk's loads are issued before the current chunk is decoded and
// consumed, so their latency (a wavefront alone on its SIMD has nothing else to hide it behind) overlaps a chunk of work.
constexpr int kPointChunk = 8;
template <int IT, class Load, class Decode, class Fn>
__device__ __forceinline__ void for_points(int nit, Load&& load, Decode&& decode, Fn&& fn) {
  if constexpr (IT > 0) {
    static_for<0, IT>([&](auto c) {
      constexpr int it = decltype(c)::value;
      const PRec r = decode(it, load(it));
      fn(it, r);
    });
  } else {
    RawRec cur[kPointChunk], nxt[kPointChunk];
#pragma unroll
    for (int j = 0; j < kPointChunk; ++j) cur[j] = load(j);
    for (int base = 0; base < nit; base += kPointChunk) {
#pragma unroll
      for (int j = 0; j < kPointChunk; ++j) nxt[j] = load(base + kPointChunk + j);
#pragma unroll
      for (int j = 0; j < kPointChunk; ++j) fn(base + j, decode(base + j, cur[j]));
#pragma unroll
      for (int j = 0; j < kPointChunk; ++j) cur[j] = nxt[j];
    }
  }
}

// The coordinates of correspondence i (clamped into the pair: no branch per correspondence) as loaded ...
template <bool RAW>
__device__ __forceinline__ void load_point_raw(const float* __restrict__ pts1, const float* __restrict__ pts2, size_t mp, int N, int i,
                                               RawRec& r) {
  const int ic = (i < N) ? i : N - 1;
  if (RAW) {
    const float4 m = reinterpret_cast<const float4*>(pts1)[mp * N + ic];
    r.v[0] = m.x; r.v[1] = m.y; r.v[2] = m.z; r.v[3] = m.w;
  } else {
    const float* a = pts1 + (mp * N + ic) * 3;
    const float* b = pts2 + (mp * N + ic) * 3;
    r.v[0] = a[0]; r.v[1] = a[1]; r.v[2] = a[2]; r.v[3] = b[0]; r.v[4] = b[1]; r.v[5] = b[2];
  }
}
// ... and decoded: image-size normalised (RAW) and sanitised, coordinates zeroed when not finite or past N.
// keep = it enters X; valid = it exists.
template <bool RAW>
__device__ __forceinline__ void decode_point(const RawRec& r, int N, int i, float hw_sx, float hw_sy, Pt& p, bool& valid, bool& keep) {
  valid = i < N;
  p.z1 = p.z2 = 1.0f;
  if (RAW) {
    p.x1 = fmaf(r.v[0], hw_sx, -1.0f);
    p.y1 = fmaf(r.v[1], hw_sy, -1.0f);
    p.x2 = fmaf(r.v[2], hw_sx, -1.0f);
    p.y2 = fmaf(r.v[3], hw_sy, -1.0f);
  } else {
    p.x1 = r.v[0]; p.y1 = r.v[1]; p.z1 = r.v[2];
    p.x2 = r.v[3]; p.y2 = r.v[4]; p.z2 = r.v[5];
  }
  // one comparison: the sum of magnitudes is below the bound only if every coordinate is finite and of sane size
  float mag = (fabsf(p.x1) + fabsf(p.y1)) + (fabsf(p.x2) + fabsf(p.y2));
  if (!RAW) mag += fabsf(p.z1) + fabsf(p.z2);
  keep = valid && (mag < 1e18f);  // false for NaN
  p.x1 = keep ? p.x1 : 0.0f; p.y1 = keep ? p.y1 : 0.0f; p.x2 = keep ? p.x2 : 0.0f; p.y2 = keep ? p.y2 : 0.0f;
  if (!RAW) { p.z1 = keep ? p.z1 : 1.0f; p.z2 = keep ? p.z2 : 1.0f; }
}
template <bool RAW>
__device__ __forceinline__ void load_point(const float* __restrict__ pts1, const float* __restrict__ pts2, size_t mp, int N, int i,
                                           float hw_sx, float hw_sy, Pt& p, bool& valid, bool& keep) {
  RawRec r;
  load_point_raw<RAW>(pts1, pts2, mp, N, i, r);
  decode_point<RAW>(r, N, i, hw_sx, hw_sy, p, valid, keep);
}

// One halving step of the in-row reduce-scatter: CNT live values per lane -> (CNT+1)/2.
template <int CNT, int STEP>
__device__ __forceinline__ void rg_halve(double* a, bool upper) {
  constexpr int H = (CNT + 1) / 2;
#pragma unroll
  for (int k = 0; k < H; ++k) {
    const double lo = a[k];
    const double hi = (k + H < CNT) ? a[k + H] : 0.0;
    const double send = upper ? lo : hi;
    const double keep = upper ? hi : lo;
    a[k] = keep + rg_xchg<STEP>(send);
  }
}

// Number of eigenvalues below x of the symmetric tridiagonal (td, te), te2 = te^2: sign changes of the three-term
// recurrence p_k = (d_k - x) p_{k-1} - e_{k-1}^2 p_{k-2}.  Division-free; with a unit-trace matrix |p_k| stays within
// [1e-150, 1].  An exact zero counts as positive: its neighbours have opposite signs, so the count is the same.
// "x lies at or below the smallest eigenvalue" = the count is zero = no term of the sequence is negative: one compare per term
// and mask arithmetic on the scalar unit instead of counting sign changes (the case of every N >= 9: kth = 0)
// (the sign bits of the nine terms are OR-ed as integers, two per v_or3_b32, instead of nine compares or a chain of fp64 minima;
// a term that is exactly -0.0 cannot occur: the recurrence only produces a zero by cancellation, which rounds to +0.0)
__device__ __forceinline__ bool sturm_none_below(const double* td, const double* te2, double x) {
  typedef unsigned u32x2 __attribute__((vector_size(8)));
  auto hi = [](double v) { return __builtin_bit_cast(u32x2, v)[1]; };  // the dword with the sign bit
  double pm2 = 1.0, pm1 = td[0] - x;
  unsigned sg = hi(pm1);
#pragma unroll
  for (int k = 1; k < 9; ++k) {
    const double p = fma(td[k] - x, pm1, -(te2[k - 1] * pm2));
    sg |= hi(p);
    pm2 = pm1;
    pm1 = p;
  }
  return (int)sg >= 0;
}
__device__ __forceinline__ int sturm_count(const double* td, const double* te2, double x) {
  double pm2 = 1.0, pm1 = td[0] - x;
  int cnt = (pm1 < 0.0) ? 1 : 0;
#pragma unroll
  for (int k = 1; k < 9; ++k) {
    const double p = fma(td[k] - x, pm1, -(te2[k - 1] * pm2));
    cnt += ((p < 0.0) != (pm1 < 0.0)) ? 1 : 0;
    pm2 = pm1;
    pm1 = p;
  }
  return cnt;
}

// Factors of one row of the design matrix: p = b (x) a with a = T1 x1, b = T2 x2 (b[2] = 1), p^ = inv * p,
// inv = 1 / max(|p|, 1e-12) = 1 / max(|a| |b|, 1e-12)  (DeepFNet.py:203-212).  Branch-free; bilinear forms p^ . g = inv * b^T G a replace the explicit 9-vector wherever only dot products
// of the row are needed.
__device__ __forceinline__ void row_ab(const Pt& p, double s1, double c1x, double c1y, double s2, double c2x, double c2y,
                                       double* a, double* b) {
  const double z1 = p.z1, z2 = p.z2;
  a[0] = s1 * ((double)p.x1 - c1x * z1); a[1] = s1 * ((double)p.y1 - c1y * z1); a[2] = z1;
  b[0] = s2 * ((double)p.x2 - c2x * z2); b[1] = s2 * ((double)p.y2 - c2y * z2);
}
__device__ __forceinline__ bool row_factors(const Pt& p, double s1, double c1x, double c1y, double s2, double c2x, double c2y,
                                            double* a, double* b, double& inv) {
  row_ab(p, s1, c1x, c1y, s2, c2x, c2y, a, b);
  const double n2 = (a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * (b[0] * b[0] + b[1] * b[1] + 1.0);
  inv = fmin(rsqrt_nr<1>(n2), 1e12);  // n2 is finite: the callers drop non-finite correspondences when they load them
  return true;
}
__device__ __forceinline__ double row_bilinear(const double* a, const double* b, const double* g) {  // b^T reshape(g) a
  const double t0 = fma(g[0], a[0], fma(g[1], a[1], g[2] * a[2])), t1 = fma(g[3], a[0], fma(g[4], a[1], g[5] * a[2]));
  const double t2 = fma(g[6], a[0], fma(g[7], a[1], g[8] * a[2]));
  return fma(b[0], t0, fma(b[1], t1, t2));
}

__device__ __forceinline__ double pivot_guard(double q) { return (fabs(q) < 1e-30) ? -1e-30 : q; }

// ---- the eigen phase shared by nothing but this kernel, kept separate for readability -----------------------------
// In: Ar = row `l` of M / trace(M) (lanes 0..8; zeros elsewhere), kth = number of eigenvalues to pass over.
// Out (uniform over the row): f[9] unit eigenvector (unoriented), z[9] its tridiagonal-space image, twist, lam, td, te;
//      per lane: hv[k] = component l of reflector k; hb[k] uniform.
// FUSED_SWEEPS (the lean build): the twisted factorisation keeps ~18 instead of 36 doubles live, same operations in the same order.
template <bool FUSED_SWEEPS = false>
__device__ __forceinline__ void eig9_select(double* Ar, const int l, const int kth, double* f, double* z, int& twist,
                                            double& lam, double* td, double* te, double* hv, double* hb) {
  // Householder tridiagonalisation, lower form: step k annihilates column k below the sub-diagonal
  static_for<0, 7>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    td[k] = rg_bcast<k>(Ar[k]);
    const double xk = (l > k) ? Ar[k] : 0.0;  // A[l][k] = A[k][l] (symmetric): the column below the diagonal
    const double x1 = rg_bcast<k + 1>(xk);
    const double sig = rg_sum_to8<k + 2>(xk * xk);
    const double nrm = sqrt_nr<2>(fma(x1, x1, sig));
    const bool ok = sig > 0.0;  // nothing to annihilate otherwise: H_k = I
    const double alpha = (x1 > 0.0) ? -nrm : nrm;
    const double vk1 = x1 - alpha;
    const double vtv = fma(vk1, vk1, sig);
    const double beta = ok ? 2.0 * rcp_nr<2, false>(vtv) : 0.0;
    te[k] = ok ? alpha : x1;
    hb[k] = beta;
    const double v = ok ? ((l == k + 1) ? vk1 : ((l > k + 1) ? xk : 0.0)) : 0.0;
    hv[k] = v;
    // p = beta A v (rows > k), K = beta/2 v^T p, w = p - K v, A -= v w^T + w v^T
    double p = rg_dot_bcast<k + 1>(v, Ar);  // one fused broadcast-FMA per term (rowgroup.h)
    p = (l > k) ? p * beta : 0.0;
    const double K = 0.5 * beta * rg_sum_to8<k + 1>(v * p);
    const double w = fma(-K, v, p);
    rg_axpy2_bcast<k + 1>(Ar, w, -v, v, -w);
  });
  td[7] = rg_bcast<7>(Ar[7]);
  td[8] = rg_bcast<8>(Ar[8]);
  te[7] = rg_bcast<8>(Ar[7]);

  DFEPE_MARK("P4b_multisection");
  // the (kth+1)-th smallest eigenvalue: 16-way multisection, lane l probes lo + (hi - lo)(l + 1)/17
  double te2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) te2[k] = te[k] * te[k];
  // unit trace, positive semi-definite: the spectrum lies in [0, 1].  Round 0 probes the powers 17^-16 ... 17^-1 (the wanted
  // eigenvalue is usually orders of magnitude below the trace: one round finds its magnitude instead of three or four rounds
  // each keeping the lowest seventeenth); the following rounds split the bracket linearly until it is 2e-16 wide (absolute:
  // that is the accuracy of M itself), 9-12 rounds instead of a fixed 13.  Counts are monotone in the probe, so
  // m = number of probes still below the wanted eigenvalue locates the sub-interval.
  const float kLog2_17 = 4.087462841250339f;
  auto pow17 = [&](int e) { return (double)exp2f((float)e * kLog2_17); };  // probe positions need no accuracy
  double lo = -1e-3, hi = 1.0 + 1e-3;
  const double frac = (double)(l + 1) * (1.0 / 17.0);
  // probe still below the wanted eigenvalue?  kth = 0 (uniform over the launch: N >= 9) takes the cheaper test; the search is
  // instantiated once per test so that the choice is made once, not in every round
  auto search = [&](auto below) {
    {
      const int m = rg_count(below(pow17(l - 16)));
      lo = (m > 0) ? pow17(m - 17) : lo;
      hi = (m < 16) ? pow17(m - 16) : hi;
    }
    // the bracket shrinks 17x per round: at most 14 rounds from width 1 to 2e-16, so the width test alone ends the loop (a NaN
    // leaves through the negated comparison)
    for (;;) {
      const double wdt = hi - lo;
      if (!(wdt > 2e-16)) break;  // uniform over the row
      const int m = rg_count(below(fma(wdt, frac, lo)));  // probes still below the wanted eigenvalue (monotone in l)
      const double step = wdt * (1.0 / 17.0);
      lo = fma(step, (double)m, lo);
      hi = lo + step;
    }
  };
  if (kth == 0) search([&](double x) { return sturm_none_below(td, te2, x); });
  else search([&](double x) { return sturm_count(td, te2, x) <= kth; });
  lam = 0.5 * (lo + hi);

  DFEPE_MARK("P4c_twisted");
  // eigenvector of T by twisted factorisation: pivots from the top (dp) and from the bottom (dm), twist where
  // gamma_k = dp_k + dm_k - (d_k - lam) is smallest in magnitude
  if constexpr (!FUSED_SWEEPS) {
  double dp[9], dm[9], rp[9], rm[9];
  dp[0] = td[0] - lam;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    dp[k] = pivot_guard(dp[k]);
    rp[k] = rcp_nr<1, false>(dp[k]);  // one Newton step from the fp32 seed: ~2e-14, far inside what the eigenvector needs
    dp[k + 1] = (td[k + 1] - lam) - te2[k] * rp[k];
  }
  dm[8] = td[8] - lam;
#pragma unroll
  for (int k = 7; k >= 0; --k) {
    dm[k + 1] = pivot_guard(dm[k + 1]);
    rm[k + 1] = rcp_nr<1, false>(dm[k + 1]);
    dm[k] = (td[k] - lam) - te2[k] * rm[k + 1];
  }
  twist = 0;
  double gbest = fabs(dp[0] + dm[0] - (td[0] - lam));
#pragma unroll
  for (int k = 1; k < 9; ++k) {
    const double g = fabs(dp[k] + dm[k] - (td[k] - lam));
    if (g < gbest) { gbest = g; twist = k; }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) z[k] = (k == twist) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 7; k >= 0; --k) z[k] = (k < twist) ? -(te[k] * rp[k]) * z[k + 1] : z[k];
#pragma unroll
  for (int k = 1; k < 9; ++k) z[k] = (k > twist) ? -(te[k - 1] * rm[k]) * z[k - 1] : z[k];
  } else {
  // The bottom-up sweep first, keeping only what the rest needs of it: its pivots (for the twist search) and the products
  // te[k-1] / dm[k] (for z below the twist).  The top-down sweep then runs FUSED with the twist search -- gamma_k is formed the
  // moment dp_k exists, after which that pivot of the bottom-up sweep is dead -- and keeps te[k] / dp[k] only: ~18 live doubles at
  // the peak instead of the 36 of two stored sweeps (round 5: what a <= 256-register build of the N <= 112 kernel needed).
  // Same operations in the same order as two stored sweeps: bit-identical z.
  double dmv[9], cm[9], cp[8];
  {
    double d = td[8] - lam;  // dm[8]
#pragma unroll
    for (int k = 7; k >= 0; --k) {
      d = pivot_guard(d);
      dmv[k + 1] = d;
      const double r = rcp_nr<1, false>(d);
      cm[k + 1] = te[k] * r;
      d = (td[k] - lam) - te2[k] * r;
    }
    dmv[0] = d;
  }
  twist = 0;
  double gbest;
  {
    double d = td[0] - lam;  // dp[0]
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (k < 8) d = pivot_guard(d);
      const double g = fabs(d + dmv[k] - (td[k] - lam));
      if (k == 0) gbest = g;
      else if (g < gbest) { gbest = g; twist = k; }
      if (k < 8) {
        const double r = rcp_nr<1, false>(d);  // one Newton step from the fp64 seed: ~2e-14, far inside what the eigenvector needs
        cp[k] = te[k] * r;
        d = (td[k + 1] - lam) - te2[k] * r;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) z[k] = (k == twist) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 7; k >= 0; --k) z[k] = (k < twist) ? -cp[k] * z[k + 1] : z[k];
#pragma unroll
  for (int k = 1; k < 9; ++k) z[k] = (k > twist) ? -cm[k] * z[k - 1] : z[k];
  }
  double zn = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) zn = fma(z[k], z[k], zn);
  zn = rsqrt_nr<2, false>(zn);  // >= 1: z[twist] = 1
#pragma unroll
  for (int k = 0; k < 9; ++k) { z[k] *= zn; f[k] = z[k]; }

  DFEPE_MARK("P4d_backtransform");
  // f = H_0 H_1 ... H_6 z, every lane keeps the whole vector (the reflector components come in by broadcast)
  static_for<0, 7>([&](auto kc) {
    constexpr int k = 6 - decltype(kc)::value;
    const double s = -hb[k] * rg_dot_bcast<k + 1>(hv[k], f);
    rg_axpy_bcast<k + 1>(f, hv[k], s);
  });
}

// ---- cooperative variant: ROWS = 16 rows (one 256-thread workgroup) share ONE pair ------------------------------------
// For N > 128 a single row would walk N / 16 correspondences per lane four times; here lane L = 16 row + l of the workgroup owns
// correspondences L, L + 256, ... (IT = ceil(N / 256) of them, in registers), the per-correspondence phases run on all 16 rows,
// pair-wide sums go through this LDS block (one barrier each), row 0 alone runs the eigen / rank-2 phases and publishes f and
// F_out for the output phase of all rows.  Same arithmetic, same `save` record as the row-per-pair kernel.
struct W8Coop {
  double red[12][16];   // one slot per pair-wide reduction: [slot][row]
  double part[16][36];  // per-row moment sums (after the in-row reduce-scatter)
  double xch[36];       // their sum over the rows
  double f[9];          // the oriented unit eigenvector
  float of[9];          // F_out
};
#ifndef DFEPE_BLOCK_SYNC
#define DFEPE_BLOCK_SYNC() __syncthreads()
#endif

// ---- forward, one pair ---------------------------------------------------------------------------------------
// IT = ceil(N / 16) correspondences per lane, kept in registers (N <= 128); IT = 0: any N, correspondences re-read per phase.
// (Round 4 measured what the re-reading costs: with the first 512 correspondences of every pair LDS-resident after the first pass --
// 128 KB per workgroup, bit-identical outputs, fabric traffic of the N = 1000 launch down by a third -- the launch took 68.9 us at
// 4096 pairs against 66-72 us without, and 131 against 126 us at 8192, where the LDS now admits one workgroup per CU.  The looped
// kernel is bound by the issue of its 21 000 instructions per wavefront, not by the 4.6 TB/s of re-reads it generates.  Reverted.)
// xch: 36 doubles of LDS owned by this pair.
// PLAIN: none of the textbook-solver variant flags is set (the hot instantiation carries no test for them).
// LEAN (IT > 0, one row per pair): the <= 256-register build for batches of >= 8192 pairs, where a SIMD holds two wavefronts if they
// fit (round 5).  The coordinates of the lane's correspondences and their 1 / |p| are NOT kept from the moments phase to the output
// phase -- the stretch that holds the eigen solve and the rank-2 step, the kernel's register peak -- but fetched again (an L2 hit:
// this very wavefront read them at its start) and re-derived by the same expressions: bit-identical outputs, ~120 more instructions,
// 45 registers fewer.  (A register cap alone makes the compiler spill exactly these values to scratch memory and back.)
template <int IT, bool RAW, bool PLAIN, int ROWS = 1, bool LEAN = false>
__device__ __forceinline__ void w8pt16_fwd_pair(const W8Args& A, const int pair, double* xch, W8Coop* co = nullptr, const int rowid = 0) {
  static_assert(ROWS == 1 || (ROWS == 16 && IT > 0) || (ROWS == 2 && IT == 0),
                "one row per pair, the 16 rows of a workgroup with the correspondences in registers, or two rows of one wavefront (looped)");
  static_assert(!LEAN || (IT > 0 && ROWS == 1), "the lean build is a variant of the registers-resident row kernel");
  constexpr int S = 16 * ROWS;  // lanes per pair
  const int l = rg_lane();
  const int L = rowid * 16 + l;  // lane within the pair
  const int N = A.N;
  // pair-wide reductions: inside the row by DPP; across the 16 rows of a cooperative workgroup through LDS (row r's value in
  // slot[r], re-read as lane r's operand of a second row reduction).  Every call site has its own slot: one barrier per reduction.
  auto psum = [&](double v, int slot) {
    v = rg_sum(v);
    if constexpr (ROWS == 2) {
      v += rg_xrow(v);  // the pair's other row sits 16 lanes away in the same wavefront: one exchange, no LDS, no barrier
    } else if constexpr (ROWS > 1) {
      if (l == 0) co->red[slot][rowid] = v;
      DFEPE_BLOCK_SYNC();
      v = rg_sum(co->red[slot][l]);
    }
    return v;
  };
  // several sums behind ONE barrier (round 6; scripts/ubench/bwd_phases.hip <pairs>: at one pair per workgroup the cooperative kernel is a
  // chain of eleven block barriers of ~600 cycles each around 4 correspondences per lane of work): the same additions in the same order
  // as K separate psum calls on slots slot0 .. slot0 + K - 1, so the results are bit-identical
  auto psumk = [&](auto& vals, int slot0) {
    constexpr int K = sizeof(vals) / sizeof(vals[0]);
#pragma unroll
    for (int k = 0; k < K; ++k) vals[k] = rg_sum(vals[k]);
    if constexpr (ROWS == 2) {
#pragma unroll
      for (int k = 0; k < K; ++k) vals[k] += rg_xrow(vals[k]);
    } else if constexpr (ROWS > 1) {
      if (l == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) co->red[slot0 + k][rowid] = vals[k];
      }
      DFEPE_BLOCK_SYNC();
#pragma unroll
      for (int k = 0; k < K; ++k) vals[k] = rg_sum(co->red[slot0 + k][l]);
    }
  };
  auto psumf = [&](float v) {  // looped kernels only (ROWS <= 2): the sum of the exponentials, in fp32 like the row kernel's
    v = rg_sum(v);
    if constexpr (ROWS == 2) v += rg_xrow(v);
    return v;
  };
  auto pmaxf = [&](float v, int slot) {
    v = rg_max(v);
    if constexpr (ROWS == 2) {
      v = fmaxf(v, rg_xrow(v));
    } else if constexpr (ROWS > 1) {
      if (l == 0) co->red[slot][rowid] = (double)v;
      DFEPE_BLOCK_SYNC();
      v = rg_max((float)co->red[slot][l]);
    }
    return v;
  };
  const unsigned variant = PLAIN ? 0u : A.variant;
  const size_t mp = (size_t)(pair % A.Bm);  // several weight sets may share one set of correspondences

  DFEPE_MARK("P0");
  // ---- phase 0: the pair's correspondences -> registers; softmax of the logits; coordinate sums ----------------
  // Loads are unconditional (index clamped into the pair, value masked afterwards): no branch per correspondence, all of a
  // lane's loads are in flight together.  A correspondence with a non-finite coordinate or weight is dropped here, once
  // (zero row of X, like the reference's NaN scrub, models/model_utils.py:5-15), so that no later phase needs a guard.
  // IT > 0: the lane's correspondences and weights stay in registers for the whole kernel (one HBM read).  IT = 0 (N > 128):
  // every phase re-reads them (16 B per correspondence, L2 hits after the first pass) and re-derives the weight.
  constexpr int ITR = (IT > 0) ? IT : 1;
  constexpr int kWi = RAW ? 4 : 6;  // slot of the weight-like value in a RawRec
  const int nit = (IT > 0) ? IT : (N + S - 1) / S;
  Pt pt[ITR];
  float wv[ITR];
  float wsm[ITR];  // softmax weights as weights_out wants them (logits mode, IT > 0)
  bool kept[ITR];
  const float* wsrc = A.wts + (size_t)pair * N;
  float lmax = 0.0f, linv = 1.0f;  // softmax of the logits: w = exp(logit - lmax) * linv
  if constexpr (IT > 0) {
    static_for<0, IT>([&](auto c) {
      constexpr int it = decltype(c)::value;
      const int i = it * S + L;
      bool valid, keep;
      load_point<RAW>(A.pts1, A.pts2, mp, N, i, A.hw_sx, A.hw_sy, pt[it], valid, keep);
      float w = wsrc[valid ? i : N - 1];
      if (A.logits_mode) w = valid ? w : -INFINITY;
      else w = (keep && fabsf(w) < 3e38f) ? w : 0.0f;
      wv[it] = w;
      kept[it] = keep;
    });
  }
  if (A.logits_mode) {
    // fused F.softmax(logits, dim=N) (DeepFNet.py:443,512)
    // the raw logit of correspondence `it` of this lane, -inf past the end
    auto logit_load = [&](int it) {
      RawRec r;
      if constexpr (IT == 0) {
        const int i = it * S + L;
        r.v[kWi] = wsrc[(i < N) ? i : N - 1];
      }
      return r;
    };
    auto logit = [&](int it, const RawRec& raw) {
      PRec r;
      if constexpr (IT > 0) r.w = wv[it];
      else r.w = (it * S + L < N) ? raw.v[kWi] : -INFINITY;
      return r;
    };
    float mx = -INFINITY;
    for_points<IT>(nit, logit_load, logit, [&](int it, const PRec& r) { mx = fmaxf(mx, r.w); });
    lmax = pmaxf(mx, 0);
    if constexpr (IT > 0) {
      float sm = 0.0f;
      static_for<0, IT>([&](auto c) {
        constexpr int it = decltype(c)::value;
        const float e = expf(wv[it] - lmax);  // padding lanes hold -inf: exactly 0
        wv[it] = e;
        sm += e;
      });
      linv = 1.0f / (float)psum((double)sm, 1);
      static_for<0, IT>([&](auto c) {
        constexpr int it = decltype(c)::value;
        const float wgt = wv[it] * linv;
        // weights_out is stored with the other per-correspondence outputs in phase 6: a store here is still in flight at the
        // join with the plain-weights path, where the compiler has to wait for it (~1 us of store latency, nothing to hide it)
        wsm[it] = wgt;
        wv[it] = kept[it] ? wgt : 0.0f;  // a dropped correspondence keeps its softmax weight in weights_out, not in X
      });
    }
    // IT = 0: the sum of exponentials rides on the centroid pass below, the weights are written by the moments pass
  }
  // the correspondence `it` of this lane as every later phase sees it: coordinates, weight in X, existence
  // Looped kernel (IT = 0), logits mode: pass 1 needs exp(logit - max) of every existing correspondence (stage 0), the moments
  // pass the normalised weight (stage 1, which also writes weights_out), the output pass reads weights_out back when the
  // caller provided it (stage 2; written by this very lane) instead of a third exponential.
  int wstage = 0;
  auto point_load = [&](int it) {
    RawRec r;
    if constexpr (IT == 0) {
      const int i = it * S + L;
      load_point_raw<RAW>(A.pts1, A.pts2, mp, N, i, r);
      const float* wp = (A.logits_mode && wstage == 2 && A.weights_out != nullptr) ? A.weights_out + (size_t)pair * N : wsrc;
      r.v[kWi] = wp[(i < N) ? i : N - 1];
    }
    return r;
  };
  auto point = [&](int it, const RawRec& raw) {
    PRec r;
    if constexpr (IT > 0) {
      r.p = pt[it];
      r.w = wv[it];
      r.ws = wv[it];
      r.valid = it * S + L < N;
      r.keep = kept[it];
    } else {
      const int i = it * S + L;
      decode_point<RAW>(raw, N, i, A.hw_sx, A.hw_sy, r.p, r.valid, r.keep);
      const float wr = raw.v[kWi];
      if (A.logits_mode) {
        r.ws = (wstage == 2 && A.weights_out != nullptr) ? wr : expf(wr - lmax) * linv;  // linv = 1 until the sum is known
        r.ws = r.valid ? r.ws : 0.0f;
        r.w = r.keep ? r.ws : 0.0f;
      } else {
        r.w = (r.keep && fabsf(wr) < 3e38f) ? wr : 0.0f;
        r.ws = r.w;
      }
    }
    return r;
  };
  const bool hartley = (variant & DFEPE_W8PT_NO_HARTLEY) == 0;
  const double invN = 1.0 / (double)N;
  double c1x = 0.0, c1y = 0.0, c2x = 0.0, c2y = 0.0, s1 = 1.0, s2 = 1.0;
  if (hartley) {
    double sx1 = 0, sy1 = 0, sx2 = 0, sy2 = 0;
    float sme = 0.0f;
    for_points<IT>(nit, point_load, point, [&](int it, const PRec& r) {  // padding / dropped correspondences hold zeros
      const Pt& p = r.p;
      sx1 += (double)p.x1; sy1 += (double)p.y1; sx2 += (double)p.x2; sy2 += (double)p.y2;
      sme += r.ws;
    });
    if (IT == 0 && A.logits_mode) linv = 1.0f / psumf(sme);
    double cs[4] = {sx1, sy1, sx2, sy2};
    psumk(cs, 2);
    c1x = cs[0] * invN; c1y = cs[1] * invN; c2x = cs[2] * invN; c2y = cs[3] * invN;
  DFEPE_MARK("P1");
    // ---- phase 1: Hartley scale (mean distance to the centroid) -------------------------------------------------
    double d1 = 0, d2 = 0;
    for_points<IT>(nit, point_load, point, [&](int it, const PRec& r) {
      const Pt& p = r.p;
      const double vm = r.valid ? 1.0 : 0.0;  // arithmetic mask: no branch around the square roots
      const double ax = (double)p.x1 - c1x, ay = (double)p.y1 - c1y;
      const double bx = (double)p.x2 - c2x, by = (double)p.y2 - c2y;
      d1 = fma(vm, sqrt_nr<1>(ax * ax + ay * ay), d1);
      d2 = fma(vm, sqrt_nr<1>(bx * bx + by * by), d2);
    });
    // Fit.normalize uses the literal 1.4142, not sqrt(2) (DeepFNet.py:168); utils_F._normalize_XY uses np.sqrt(2)
    const double hscale = (variant & DFEPE_W8PT_SQRT2) ? 1.4142135623730951 : 1.4142;
    double ds[2] = {d1, d2};
    psumk(ds, 6);
    s1 = hscale * rcp_nr<2>(ds[0] * invN);
    s2 = hscale * rcp_nr<2>(ds[1] * invN);
  }

  if (IT == 0 && A.logits_mode && !hartley) {
    float sme = 0.0f;
    for_points<IT>(nit, point_load, point, [&](int it, const PRec& r) { sme += r.ws; });
    linv = 1.0f / psumf(sme);
  }
  wstage = 1;
  DFEPE_MARK("P2");
  // ---- phase 2: X^T X = sum_i k_i^2 (b b^T) (x) (a a^T): 36 distinct fp64 sums per lane ---------------------------
  double acc[36];
  double invs[IT > 0 ? IT : 1];
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = 0.0;
  for_points<IT>(nit, point_load, point, [&](int it, const PRec& r) {
    const Pt& p = r.p;
    const double w = (double)r.w;
    const double z1 = p.z1, z2 = p.z2;
    const double a0 = s1 * ((double)p.x1 - c1x * z1), a1 = s1 * ((double)p.y1 - c1y * z1), a2 = z1;
    const double b0 = s2 * ((double)p.x2 - c2x * z2), b1 = s2 * ((double)p.y2 - c2y * z2);
    const double n2 = (a0 * a0 + a1 * a1 + a2 * a2) * (b0 * b0 + b1 * b1 + 1.0);  // |p|^2, finite (phase 0 dropped the rest)
    // (w / max(|p|, 1e-12))^2; a dropped or padding correspondence has w = 0 and contributes exact zeros.  1 / |p| is kept
    // for the residual of phase 6 when the correspondences live in registers.
    const double inv = fmin(rsqrt_nr<1, !RAW>(n2), 1e12);  // ~2e-14: it only scales a weight; RAW: n2 >= 1
    if constexpr (IT > 0 && !LEAN) invs[it] = inv;
    const double wi = (variant & DFEPE_W8PT_NO_ROWNORM) ? w : w * inv;
    const double k2 = wi * wi;
    const double aa[6] = {a0 * a0, a0 * a1, a0 * a2, a1 * a1, a1 * a2, a2 * a2};
    const double bb[6] = {k2 * b0 * b0, k2 * b0 * b1, k2 * b0, k2 * b1 * b1, k2 * b1, k2};
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
      for (int v = 0; v < 6; ++v) acc[6 * u + v] = fma(bb[u], aa[v], acc[6 * u + v]);
    if constexpr (IT == 0) {
      if (A.logits_mode && A.weights_out != nullptr && r.valid) A.weights_out[(size_t)pair * N + it * S + L] = r.ws;
    }
  });
  wstage = 2;

  DFEPE_MARK("P3");
  // ---- phase 3: reduce-scatter inside the row (36 -> 18 -> 9 -> 5 -> 3 values per lane), M through LDS --------------
  rg_halve<36, 8>(acc, (l & 8) != 0);
  rg_halve<18, 4>(acc, (l & 4) != 0);
  rg_halve<9, 2>(acc, (l & 2) != 0);
  rg_halve<5, 1>(acc, (l & 1) != 0);
  {
    int cnt = 36, idx = 0, width = 36;  // mirror of the halving schedule: this lane ends up owning sums idx .. idx+cnt-1
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
      const int h = (width + 1) / 2;
      if (l & m) { idx += h; cnt -= h; } else { cnt = (cnt < h) ? cnt : h; }
      width = h;
    }
    if constexpr (ROWS == 2) {  // both rows end up with the pair's totals and write the same values
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[k] += rg_xrow(acc[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < cnt) xch[idx + k] = acc[k];
    } else if constexpr (ROWS > 1) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < cnt) co->part[rowid][idx + k] = acc[k];
      DFEPE_BLOCK_SYNC();
      if (L < 36) {  // sum the 16 rows' partial moments in a fixed order
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += co->part[r][L];
        co->xch[L] = t;
      }
      DFEPE_BLOCK_SYNC();
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < cnt) xch[idx + k] = acc[k];
    }
  }
  if constexpr (ROWS > 2) xch = co->xch;
  else rg_sync();
  double f[9];
  float of[9];
  // ROWS = 2: BOTH rows run the solver phases on the same numbers (they share the wavefront's instruction stream: nothing is saved
  // by masking one of them off); row 0 alone stores
  const bool storer = ROWS != 2 || rowid == 0;
  if (ROWS <= 2 || rowid == 0) {  // the solver phases: one row per pair
  // sum (u, v) is M[3r+c][3r'+c'] for (r, r') = symmetric pair u, (c, c') = symmetric pair v.  Lane i < 9 fetches row i.
  double Ar[9];
  double tr;
  {
    const int li = (l < 9) ? l : 0;
    const int r = li / 3, c = li - 3 * r;
    auto sym = [](int a, int b) { const int lo = (a < b) ? a : b, hi = (a < b) ? b : a; return (lo * (5 - lo)) / 2 + hi; };
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int rj = j / 3, cj = j % 3;
      const double m = xch[6 * sym(r, rj) + sym(c, cj)];
      Ar[j] = (l < 9) ? m : 0.0;
    }
    const double dg = xch[6 * sym(r, r) + sym(c, c)];
    tr = rg_sum_to8<0>(dg);
  }
  const double inv_tr = (tr > 0.0) ? rcp_nr<2, false>(tr) : 1.0;
#pragma unroll
  for (int j = 0; j < 9; ++j) Ar[j] *= inv_tr;

  DFEPE_MARK("P4");
  // ---- phase 4: the eigenpair the reference picks --------------------------------------------------------------
  // torch.svd(X)[2][:, -1] is the right singular vector of the smallest of the min(N,9) singular values
  // (DeepFNet.py:232-233): for N >= 9 the smallest eigenvalue of X^T X; for N < 9 the reduced SVD has only N columns,
  // so the reference takes the smallest of the N non-null directions: 9 - N eigenvalues are passed over.
  const int kth = (N >= 9) ? 0 : 9 - N;
  double z[9], td[9], te[8], hv[7], hb[7], lam;
  int twist;
  eig9_select<LEAN>(Ar, l, kth, f, z, twist, lam, td, te, hv, hb);
  // The `save` record in the LEAN build: everything but the reflector components is uniform over the row, and lane s of
  // {0, 1, 2, 8..15} assembles floats 8 s .. 8 s + 7 of it with one select per float (see the stores at the end of the solver phases).
  // Here the pieces are folded into the lane's slice AS SOON AS THEY ARE FINAL -- the tridiagonal, the eigenvalue and the reflector
  // scales right now, z and f after the orientation, the singular triplet after the rank-2 step -- so that 9 + 8 + 1 + 7 doubles stop
  // being live across the rank-2 step (part of what takes that build from 289 to <= 256 registers).  The resident build keeps
  // round 4's code, which assembles the whole record at the end: at one wavefront per SIMD the early selects measured 3-4 % slower
  // (12.5 -> 13.0 us at 4096 pairs, scripts/ab_fit_sizes.py).
  const bool saving = LEAN && A.save != nullptr;
  float slice[LEAN ? 8 : 1];
#pragma unroll
  for (int c = 0; c < (LEAN ? 8 : 1); ++c) slice[c] = 0.0f;
  auto put = [&](const int idx, const float v) {  // idx is a literal after unrolling: slice[] is indexed at compile time
    if constexpr (LEAN) slice[idx & 7] = ((idx >> 3) == l) ? v : slice[idx & 7];
  };
  auto put2 = [&](const int idx, const double v) {  // a double in two consecutive floats
    typedef float f32x2 __attribute__((vector_size(8)));
    const f32x2 h = __builtin_bit_cast(f32x2, v);
    put(idx, h[0]);
    put(idx + 1, h[1]);
  };
  float hvf[7];
  if constexpr (LEAN) {
    if (saving) {
#pragma unroll
      for (int c = 0; c < 9; ++c) put2(S16_TD + 2 * c, td[c]);
#pragma unroll
      for (int c = 0; c < 8; ++c) put2(S16_TE + 2 * c, te[c]);
      put2(S16_LAM, lam);
      put(S16_TWIST, (float)twist);
#pragma unroll
      for (int c = 0; c < 7; ++c) { put(S16_HB + c, (float)hb[c]); hvf[c] = (float)hv[c]; }
      put(S16_INVTR, (float)inv_tr);
      put(S16_TAG, S16_TAG_VALUE);
    }
  }

  DFEPE_MARK("P5");
  // ---- phase 5: orientation, rank-2 projection, de-normalisation (uniform over the row) -------------------------
  double fn2 = 0.0;
#pragma unroll
  for (int c = 0; c < 9; ++c) fn2 = fma(f[c], f[c], fn2);
  double big = f[0];  // orientation: largest-magnitude component positive (first one on ties)
#pragma unroll
  for (int c = 1; c < 9; ++c)
    if (fabs(f[c]) > fabs(big)) big = f[c];
  const double sgn = (big < 0.0) ? -1.0 : 1.0;
  const double fscale = sgn * rsqrt_nr<2, false>(fn2);  // a unit vector up to rounding
#pragma unroll
  for (int c = 0; c < 9; ++c) f[c] *= fscale;
  if constexpr (LEAN) {
    if (saving) {
#pragma unroll
      for (int c = 0; c < 9; ++c) { put(S16_F + c, (float)f[c]); put(S16_Z + c, (float)(sgn * z[c])); }
    }
  }

  DFEPE_MARK("P5b_rank2");
  // rank-2 step: F' = F - s3 u3 v3^T with the smallest singular triplet in closed form (fp64)
  double u3[3], v3[3], s3;
  double Fp[9];
  if (variant & DFEPE_W8PT_FORCE_110) {  // E' = U diag(1,1,0) V^T = u1 v1^T + u2 v2^T  (utils_F.py:148-149); forward-only variant
    float Ff[9], U3[9], S3[3], V3[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) Ff[c] = (float)f[c];
    svd3_fast(Ff, U3, S3, V3);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        Fp[3 * r + c] = (double)U3[3 * r] * (double)V3[3 * c] + (double)U3[3 * r + 1] * (double)V3[3 * c + 1];
    u3[0] = u3[1] = u3[2] = v3[0] = v3[1] = v3[2] = s3 = 0.0;
  } else {
    smallest_singular_triplet3(f, u3, v3, s3);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Fp[3 * r + c] = fma(-s3 * u3[r], v3[c], f[3 * r + c]);
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
  {
    // one store per lane: lane c writes out[c]
    float mine = of[0];
#pragma unroll
    for (int c = 1; c < 9; ++c) mine = (l == c) ? of[c] : mine;
    if (l < 9 && storer) A.F_out[(size_t)pair * 9 + l] = mine;
  }

  DFEPE_MARK("P5s_save");
  if constexpr (LEAN) {
  if (saving) {
    float* sv = static_cast<float*>(__builtin_assume_aligned(A.save, 16)) + (size_t)pair * DFEPE_SAVE_FLOATS;
    // Under load a global store INSTRUCTION costs this lone wavefront 50-70 cycles whatever its width or the number of lanes behind
    // it (scripts/ubench/lat2.hip), a select 5: so the uniform part (floats 0..23 and 64..127 of the record) leaves in TWO
    // instructions -- each lane's slice as two 16-byte pieces -- instead of the 26 stores a lane per piece needed (rounds 2-3).
    // Floats 24..63 hold only reflector components, which live in their lanes.
    put(S16_T1 + 0, (float)s1); put(S16_T1 + 1, (float)c1x); put(S16_T1 + 2, (float)c1y);
    put(S16_T2 + 0, (float)s2); put(S16_T2 + 1, (float)c2x); put(S16_T2 + 2, (float)c2y);
#pragma unroll
    for (int c = 0; c < 3; ++c) { put(S16_U3 + c, (float)u3[c]); put(S16_V3 + c, (float)v3[c]); }
    put(S16_S3, (float)s3);
    static_assert(S16_Z + 9 <= 24 && S16_HV >= 24 && S16_HV + 35 <= 64 && S16_HB >= 64 && S16_TD >= 64, "slices 3..7 of the record hold reflector components only");
    if (l < 3 || l >= 8) {
      float4* dst = reinterpret_cast<float4*>(sv + 8 * l);
      float4 q0, q1;
      q0.x = slice[0]; q0.y = slice[1]; q0.z = slice[2]; q0.w = slice[3];
      q1.x = slice[4]; q1.y = slice[5]; q1.z = slice[6]; q1.w = slice[7];
      dst[0] = q0;
      dst[1] = q1;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k)  // lanes that hold no component of reflector k write a scratch slot: no branch per reflector
      sv[(l > k && l < 9) ? S16_HV + s16_hv_off(k) + (l - k - 1) : 24] = hvf[k];
  }
  } else {
  if (A.save != nullptr && storer) {
    float* sv = static_cast<float*>(__builtin_assume_aligned(A.save, 16)) + (size_t)pair * DFEPE_SAVE_FLOATS;
    // Everything but the reflector components is uniform over the row.  Under load a global store INSTRUCTION costs this lone
    // wavefront 50-70 cycles whatever its width or the number of lanes behind it (scripts/ubench/lat2.hip), a select 5: so the
    // uniform part (floats 0..23 and 64..127 of the record) leaves in TWO instructions -- lane s of {0, 1, 2, 8..15} assembles
    // floats 8 s .. 8 s + 7 with one select per float (~95 v_cndmask) and stores them as two 16-byte pieces -- instead of the
    // 26 stores a lane per piece needed (rounds 2-3).  Floats 24..63 hold only reflector components, which live in their lanes.
    float slice[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) slice[c] = 0.0f;
    auto put = [&](const int idx, const float v) {  // idx is a literal after unrolling: slice[] is indexed at compile time
      slice[idx & 7] = ((idx >> 3) == l) ? v : slice[idx & 7];
    };
    auto put2 = [&](const int idx, const double v) {  // a double in two consecutive floats
      typedef float f32x2 __attribute__((vector_size(8)));
      const f32x2 h = __builtin_bit_cast(f32x2, v);
      put(idx, h[0]);
      put(idx + 1, h[1]);
    };
    put(S16_T1 + 0, (float)s1); put(S16_T1 + 1, (float)c1x); put(S16_T1 + 2, (float)c1y);
    put(S16_T2 + 0, (float)s2); put(S16_T2 + 1, (float)c2x); put(S16_T2 + 2, (float)c2y);
#pragma unroll
    for (int c = 0; c < 9; ++c) { put(S16_F + c, (float)f[c]); put(S16_Z + c, (float)(sgn * z[c])); }
#pragma unroll
    for (int c = 0; c < 9; ++c) put2(S16_TD + 2 * c, td[c]);
#pragma unroll
    for (int c = 0; c < 8; ++c) put2(S16_TE + 2 * c, te[c]);
    put2(S16_LAM, lam);
#pragma unroll
    for (int c = 0; c < 3; ++c) { put(S16_U3 + c, (float)u3[c]); put(S16_V3 + c, (float)v3[c]); }
    put(S16_S3, (float)s3);
    put(S16_TWIST, (float)twist);
#pragma unroll
    for (int c = 0; c < 7; ++c) put(S16_HB + c, (float)hb[c]);
    put(S16_INVTR, (float)inv_tr);
    put(S16_TAG, S16_TAG_VALUE);
    static_assert(S16_Z + 9 <= 24 && S16_HV >= 24 && S16_HV + 35 <= 64 && S16_HB >= 64 && S16_TD >= 64, "slices 3..7 of the record hold reflector components only");
    if (l < 3 || l >= 8) {
      float4* dst = reinterpret_cast<float4*>(sv + 8 * l);
      float4 q0, q1;
      q0.x = slice[0]; q0.y = slice[1]; q0.z = slice[2]; q0.w = slice[3];
      q1.x = slice[4]; q1.y = slice[5]; q1.z = slice[6]; q1.w = slice[7];
      dst[0] = q0;
      dst[1] = q1;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k)  // lanes that hold no component of reflector k write a scratch slot: no branch per reflector
      sv[(l > k && l < 9) ? S16_HV + s16_hv_off(k) + (l - k - 1) : 24] = (float)hv[k];
  }
  }
  if constexpr (ROWS > 2) {
    if (l < 9) { co->f[l] = f[l]; co->of[l] = of[l]; }
  }
  }  // solver row
  if constexpr (ROWS > 2) {
    DFEPE_BLOCK_SYNC();
#pragma unroll
    for (int c = 0; c < 9; ++c) { f[c] = co->f[c]; of[c] = co->of[c]; }
  }

  DFEPE_MARK("P6");
  // ---- phase 6: per-correspondence outputs ----------------------------------------------------------------------
  float* rdst = A.residual + (size_t)pair * N;
  float* edst = (A.epi_res != nullptr) ? A.epi_res + (size_t)pair * N : nullptr;
  float* wdst = A.weights_out + (size_t)pair * N;  // only dereferenced where weights_out was tested
  // LEAN: the correspondence is fetched and decoded again; its weight in X is still in the lane's registers
  auto reload = [&](int it) {
    RawRec r;
    if constexpr (LEAN) load_point_raw<RAW>(A.pts1, A.pts2, mp, N, it * S + L, r);
    return r;
  };
  auto point_out = [&](int it, const RawRec& raw) {
    PRec r;
    if constexpr (LEAN) {
      decode_point<RAW>(raw, N, it * S + L, A.hw_sx, A.hw_sy, r.p, r.valid, r.keep);
      r.w = r.ws = wv[it];
    }
    return r;
  };
  // the symmetric epipolar residual of TWO correspondences in packed fp32 (DFEPE_P6_FAST): each half is the scalar expression of
  // out_body below, operation for operation, so the outputs are bit-identical -- a lone wavefront just issues half as many instructions
  auto epi2 = [&](const pk2 x1, const pk2 y1, const pk2 z1, const pk2 x2, const pk2 y2, const pk2 z2, float& da, float& db) {
    auto zmul = [](pk2 z, pk2 v) { return RAW ? v : pk_mul(z, v); };  // pixel matches: z = 1, and 1 * v = v exactly
    const pk2 l1x = pk_fma(x2, pk_splat(of[0]), pk_fma(y2, pk_splat(of[3]), zmul(z2, pk_splat(of[6]))));
    const pk2 l1y = pk_fma(x2, pk_splat(of[1]), pk_fma(y2, pk_splat(of[4]), zmul(z2, pk_splat(of[7]))));
    const pk2 l1z = pk_fma(x2, pk_splat(of[2]), pk_fma(y2, pk_splat(of[5]), zmul(z2, pk_splat(of[8]))));
    const pk2 l2x = pk_fma(x1, pk_splat(of[0]), pk_fma(y1, pk_splat(of[1]), zmul(z1, pk_splat(of[2]))));
    const pk2 l2y = pk_fma(x1, pk_splat(of[3]), pk_fma(y1, pk_splat(of[4]), zmul(z1, pk_splat(of[5]))));
    const pk2 dd = pk_fma(x1, l1x, pk_fma(y1, l1y, zmul(z1, l1z)));
    const pk2 q1 = pk_fma(l1x, l1x, pk_mul(l1y, l1y)), q2 = pk_fma(l2x, l2x, pk_mul(l2y, l2y));
    const pk2 n1 = pk_add(pk_make(hw_sqrt(pk_lo(q1)), hw_sqrt(pk_hi(q1))), pk_splat(1e-6f));
    const pk2 m2 = pk_add(pk_make(hw_sqrt(pk_lo(q2)), hw_sqrt(pk_hi(q2))), pk_splat(1e-6f));
    const pk2 rr = pk_add(pk_make(hw_rcp(pk_lo(n1)), hw_rcp(pk_hi(n1))), pk_make(hw_rcp(pk_lo(m2)), hw_rcp(pk_hi(m2))));
    da = fminf(fabsf(pk_lo(dd)) * pk_lo(rr), A.clamp_at);
    db = fminf(fabsf(pk_hi(dd)) * pk_hi(rr), A.clamp_at);
  };
  // GUARD = false (DFEPE_P6_FAST, the registers-resident kernels): the caller has established that this correspondence exists and that
  // the training outputs (epi_res, weights_out) are wanted -- straight-line stores, no exec region and no pointer test per correspondence
  // d_given: the epipolar residual of this correspondence was computed by epi2 (d_in)
  auto out_body = [&](int it, const PRec& rec, auto guard_c, auto d_given, const float d_in) {
    constexpr bool GUARD = decltype(guard_c)::value;
    constexpr bool DGIVEN = decltype(d_given)::value;
    const int i = it * S + L;
    const Pt& p = rec.p;
    const float wf = rec.w;
    const bool valid = rec.valid;
    // residual_i = w_i p^_i . f  (DeepFNet.py:203-214,251); straight-line, only the stores are guarded
    double ra[3], rb[2], inv;
    if constexpr (IT > 0 && !LEAN) {
      row_ab(p, s1, c1x, c1y, s2, c2x, c2y, ra, rb);
      inv = invs[it];
    } else if constexpr (LEAN) {  // the expressions of phase 2, so that 1 / |p| comes out bit-identical
      row_ab(p, s1, c1x, c1y, s2, c2x, c2y, ra, rb);
      const double n2 = (ra[0] * ra[0] + ra[1] * ra[1] + ra[2] * ra[2]) * (rb[0] * rb[0] + rb[1] * rb[1] + 1.0);
      inv = fmin(rsqrt_nr<1, !RAW>(n2), 1e12);
    } else {
      row_factors(p, s1, c1x, c1y, s2, c2x, c2y, ra, rb, inv);
    }
    if (!PLAIN) inv = (variant & DFEPE_W8PT_NO_ROWNORM) ? 1.0 : inv;  // X_i = w_i p_i
    const float r = (float)(row_bilinear(ra, rb, f) * inv * (double)wf);
    // l1 = F^T x2 (row form x2 F), l2 = F x1, dd = x2^T F x1 = x1 . l1     (utils_F.py:402-411), fp32 like the reference
    float d = d_in;
    if constexpr (!DGIVEN) {
    const float l1x = fmaf(p.x2, of[0], fmaf(p.y2, of[3], p.z2 * of[6]));
    const float l1y = fmaf(p.x2, of[1], fmaf(p.y2, of[4], p.z2 * of[7]));
    const float l1z = fmaf(p.x2, of[2], fmaf(p.y2, of[5], p.z2 * of[8]));
    const float l2x = fmaf(p.x1, of[0], fmaf(p.y1, of[1], p.z1 * of[2]));
    const float l2y = fmaf(p.x1, of[3], fmaf(p.y1, of[4], p.z1 * of[5]));
    const float dd = fmaf(p.x1, l1x, fmaf(p.y1, l1y, p.z1 * l1z));
    const float n1 = hw_sqrt(fmaf(l1x, l1x, l1y * l1y)) + 1e-6f;  // v_sqrt_f32 / v_rcp_f32: 1 ulp, far inside the tolerance
    const float m2 = hw_sqrt(fmaf(l2x, l2x, l2y * l2y)) + 1e-6f;
    d = fminf(fabsf(dd) * (hw_rcp(n1) + hw_rcp(m2)), A.clamp_at);
    }
    if constexpr (!GUARD) {
      rdst[i] = r;
      edst[i] = d;
      if constexpr (IT > 0) wdst[i] = wsm[it];
    } else {
    if (valid) {
      rdst[i] = r;
      if (edst != nullptr) edst[i] = d;
      if constexpr (IT > 0) {
        if (A.logits_mode && A.weights_out != nullptr) A.weights_out[(size_t)pair * N + i] = wsm[it];
      }
    }
    }
  };
  auto out_guarded = [&](int it, const PRec& rec) { out_body(it, rec, std::true_type{}, std::false_type{}, 0.0f); };
  if constexpr (LEAN) for_points<IT>(nit, reload, point_out, out_guarded);
  else if constexpr (DFEPE_P6_FAST != 0 && IT > 1) {
    // all of the lane's correspondences but its last one exist whenever N > S (IT - 1) -- the shape the instantiation was chosen for
    // (N = 100: IT = 7, 96 < N) -- and the training call wants every per-correspondence output: 3 IT stores in a row
    if (N > S * (IT - 1) && edst != nullptr && A.logits_mode && A.weights_out != nullptr) {
      static_for<0, (IT - 1) / 2>([&](auto c) {
        constexpr int it = 2 * decltype(c)::value;
        float da, db;
        epi2(pk_make(pt[it].x1, pt[it + 1].x1), pk_make(pt[it].y1, pt[it + 1].y1), pk_make(pt[it].z1, pt[it + 1].z1),
             pk_make(pt[it].x2, pt[it + 1].x2), pk_make(pt[it].y2, pt[it + 1].y2), pk_make(pt[it].z2, pt[it + 1].z2), da, db);
        out_body(it, point(it, RawRec{}), std::false_type{}, std::true_type{}, da);
        out_body(it + 1, point(it + 1, RawRec{}), std::false_type{}, std::true_type{}, db);
      });
      if constexpr (((IT - 1) & 1) != 0) out_body(IT - 2, point(IT - 2, RawRec{}), std::false_type{}, std::false_type{}, 0.0f);
      out_body(IT - 1, point(IT - 1, RawRec{}), std::true_type{}, std::false_type{}, 0.0f);
    } else {
      for_points<IT>(nit, point_load, point, out_guarded);
    }
  } else for_points<IT>(nit, point_load, point, out_guarded);
  DFEPE_MARK("Pend");
}
