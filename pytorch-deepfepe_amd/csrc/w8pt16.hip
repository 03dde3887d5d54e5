// w8pt16 -- the row-per-pair kernels of the weighted 8-point fit: launchers.
//
// Grid: one 256-thread workgroup = 16 pairs (4 wavefronts x 4 rows).  B = 4096 pairs -> 256 workgroups = one per CU,
// one wavefront per SIMD; larger batches stack more wavefronts per SIMD.  No block-level barrier; each pair owns 36
// doubles of LDS for one exchange.  Bodies: w8pt16_body.h (forward), w8pt16_bwd_body.h (adjoint).
#include "dfepe_common.h"
#include "w8pt16_body.h"
#include "w8pt16_bwd_body.h"
#include "loss_head_body.h"
#include <cstdlib>

#include "cheirality_body.h"

namespace {

constexpr int kPairsPerBlock = 16;
constexpr int kCoopMaxN = 2048;  // cooperative workgroup per pair: N / 256 correspondences per lane in registers (IT <= 8)
// ... while the batch is small: the solver phase needs ~256 registers, so two workgroups (two pairs) fit a CU and a launch
// takes ceil(pairs / 512) rounds of ~13 us (N = 1000); from 4096 pairs on one row per pair (IT = 0, 78 us) is faster
constexpr int kCoopMaxPairs = 3072;
// the forward and the backward agree on this by construction (same N, same pair count, same flag), and the `save` record is the
// same either way
// the lean forward fit (<= 256 registers) from this many pairs on.  Measured against the resident build (scripts/ab_fit_sizes.py, us per
// 4096 pairs, N = 100): 4096 pairs 14.5 vs 12.5 (one wavefront per SIMD: the extra instructions only cost), 8192 11.4 vs 11.2, 16384
// 9.9 vs 10.9, 32768 9.1 vs 10.4 (issue floor of its 4 140 instructions at the sustained clock: ~8.1).  DFEPE_FIT_LEAN = 0 / 1 in the
// environment forces it off / on (A/B timing; the outputs are bit-identical either way)
constexpr int kLeanMinPairs = 12288;
static bool use_lean(int pairs) {
  static const int forced = [] { const char* e = getenv("DFEPE_FIT_LEAN"); return e ? atoi(e) : -1; }();
  return forced < 0 ? pairs >= kLeanMinPairs : forced != 0;
}
// N > 128, one row per pair: below this many pairs a SIMD holds a single wavefront, and two rows per pair (twice the wavefronts, each
// with half the per-correspondence work) are faster.  DFEPE_FIT_PAIR2 = 0 / 1 forces it off / on (A/B timing).
constexpr int kPair2MaxPairs = 8192;
static bool use_pair2(int pairs) {
  static const int forced = [] { const char* e = getenv("DFEPE_FIT_PAIR2"); return e ? atoi(e) : -1; }();
  return forced < 0 ? pairs < kPair2MaxPairs : forced != 0;
}
static bool use_coop(int N, int pairs, bool row_per_pair) { return N > 128 && N <= kCoopMaxN && pairs <= kCoopMaxPairs && !row_per_pair; }
// the FORWARD fit leaves the cooperative workgroup earlier since round 5: two rows of a wavefront per pair (w8pt16_pair2_fwd_kernel) beat it
// from ~1300 pairs on (N = 1000, scripts/fit_n1000_sizes.py: 1024 pairs 29.7 vs 34.1 us, 2048 54.7 vs 37.7, 3072 78.3 vs 57.0); the `save`
// record is the same whichever kernel wrote it, so the backward keeps its own threshold
constexpr int kCoopFwdMaxPairs = 1280;
static bool use_coop_fwd(int N, int pairs, bool row_per_pair, bool raw) {
  return use_coop(N, pairs, row_per_pair) && (pairs <= kCoopFwdMaxPairs || !raw);
}

// Kernel arguments (forward and backward alike): what a wavefront needs before it can issue its global loads comes first, as plain scalars / pointers --
// with -amdgpu-kernarg-preload-count=16 (build.py) the command processor hands those 16 dwords over in SGPRs at wave launch,
// so the loads do not wait for a scalar-cache miss on the kernarg segment first; the rest follows as a struct and is fetched
// in the shadow of the global loads.
struct W8FwdRest {
  float* epi_res;
  float* save;
  float* weights_out;
  int logits_mode;
  unsigned variant;
};

// Start offsets (round 6).  At 4096 pairs a CU holds ONE workgroup whose four wavefronts -- one per SIMD -- start in the same cycle and
// run the same straight-line instruction stream in lockstep, so they reach every shared unit of the CU (instruction fetch, the texture
// addresser with its loads and stores, the LDS exchange of the moments) in the same cycles.  Wavefront w waits w x DFEPE_*_STAGGER
// cycles first.  Measured on one box (scripts/ab_fit.sh, captured step at B = 4096, N = 100, ms): none 0.1058-0.1062; 8 / 16 cycles
// per wavefront 0.1029-0.1034; 32: 0.1041-0.1044; 64: 0.1034-0.1038; 128: 0.1034-0.1037; 192: 0.1041-0.1044; 512: 0.1051-0.1060 -- the
// gain is in not being aligned, more distance only delays the last wavefront.  Also measured and dropped: offsets between the
// workgroups of an XCD (+0.5 ... +1.5 us per step), repeating the offset at the phase boundaries of the body (+1.5 ... +6 us: the loop
// in the middle of the stream costs more than it spreads), the same on the loss tail's F-loss wavefronts (+-0).
#define DFEPE_STAGGER_WAIT(X)                                              \
  do {                                                                     \
    if constexpr ((X) >= 64) __builtin_amdgcn_s_sleep((X) / 64);           \
    else {                                                                 \
      if constexpr ((X) > 16) asm volatile("s_nop 15");                    \
      asm volatile("s_nop %0" ::"n"(((X) - 1) & 15));                      \
    }                                                                      \
  } while (0)
#ifndef DFEPE_PAIR2_STAGGER
#define DFEPE_PAIR2_STAGGER 16  // the N = 1000 fit with two rows per pair (two wavefronts per SIMD at 4096 pairs): wavefront w of a workgroup, and
                                // the workgroups a CU receives second (ids 256 apart), start 16 (w [+ 4]) cycles late: bench.py --config 5
                                // 0.1586-0.1588 -> 0.1561-0.1569 ms (same box, twice).  The same on the lean fit (config 4 as one 32768-pair
                                // batch: workgroups queue for the CUs, only the first wave of them starts together): +-0, not kept
#endif
#ifndef DFEPE_FWD_STAGGER
#define DFEPE_FWD_STAGGER 16  // cycles per wavefront index, forward fit (0: off)
#endif
#ifndef DFEPE_BWD_STAGGER
#define DFEPE_BWD_STAGGER 16  // ... backward fit
#endif
template <int IT, bool RAW, bool PLAIN>
__global__ void __launch_bounds__(256)
w8pt16_fwd_kernel(const float* pts1, const float* pts2, const float* wts, int B, int Bm, int N, float hw_sx, float hw_sy,
                  float clamp_at, float* F_out, float* residual, const W8FwdRest R) {
  __shared__ double xch[kPairsPerBlock * 36];
  const int row = (int)(threadIdx.x >> 4);
  const int pair = (int)blockIdx.x * kPairsPerBlock + row;
  if (pair >= B) return;  // a whole row leaves; rows never wait for each other
#if DFEPE_FWD_STAGGER
  for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) DFEPE_STAGGER_WAIT(DFEPE_FWD_STAGGER);
#endif
  W8Args A;
  A.pts1 = pts1; A.pts2 = pts2; A.wts = wts; A.B = B; A.Bm = Bm; A.N = N; A.hw_sx = hw_sx; A.hw_sy = hw_sy;
  A.clamp_at = clamp_at; A.F_out = F_out; A.residual = residual; A.epi_res = R.epi_res; A.save = R.save;
  A.weights_out = R.weights_out; A.logits_mode = R.logits_mode; A.variant = R.variant; A.row_per_pair = false;
  w8pt16_fwd_pair<IT, RAW, PLAIN>(A, pair, xch + row * 36);
}

// The same kernel in <= 256 registers (w8pt16_body.h: LEAN): two wavefronts per SIMD.  Taken from kLeanMinPairs pairs on, where a
// SIMD has wavefronts queueing for it (at 4096 pairs = one wavefront per SIMD the few extra instructions would only cost).
template <int IT, bool RAW, bool PLAIN>
__global__ void __launch_bounds__(256, 2)
w8pt16_fwd_lean_kernel(const float* pts1, const float* pts2, const float* wts, int B, int Bm, int N, float hw_sx, float hw_sy,
                       float clamp_at, float* F_out, float* residual, const W8FwdRest R) {
  __shared__ double xch[kPairsPerBlock * 36];
  const int row = (int)(threadIdx.x >> 4);
  const int pair = (int)blockIdx.x * kPairsPerBlock + row;
  if (pair >= B) return;
  W8Args A;
  A.pts1 = pts1; A.pts2 = pts2; A.wts = wts; A.B = B; A.Bm = Bm; A.N = N; A.hw_sx = hw_sx; A.hw_sy = hw_sy;
  A.clamp_at = clamp_at; A.F_out = F_out; A.residual = residual; A.epi_res = R.epi_res; A.save = R.save;
  A.weights_out = R.weights_out; A.logits_mode = R.logits_mode; A.variant = R.variant; A.row_per_pair = false;
  w8pt16_fwd_pair<IT, RAW, PLAIN, 1, true>(A, pair, xch + row * 36);
}

// N > 128 at a few thousand pairs: TWO rows of a wavefront per pair (w8pt16_body.h: ROWS = 2).  One row per pair is one wavefront per
// SIMD at 4096 pairs -- a lone in-order stream of 21 000 instructions at N = 1000 --; with two rows a pair's correspondences are
// walked by 32 lanes (half the per-correspondence instructions per wavefront), the eigen phases run once per wavefront as before,
// and the 2048 wavefronts are two per SIMD (236 registers) that fill each other's issue bubbles.
constexpr int kPairsPerBlock2 = 8;
template <bool RAW, bool PLAIN>
__global__ void __launch_bounds__(256, 2)  // instantiated for pixel matches only
w8pt16_pair2_fwd_kernel(const float* pts1, const float* pts2, const float* wts, int B, int Bm, int N, float hw_sx, float hw_sy,
                        float clamp_at, float* F_out, float* residual, const W8FwdRest R) {
  __shared__ double xch[kPairsPerBlock2 * 36];
  const int prow = (int)(threadIdx.x >> 5);  // pair within the workgroup
  const int pair = (int)blockIdx.x * kPairsPerBlock2 + prow;
  if (pair >= B) return;  // both rows of a pair leave together
#if DFEPE_PAIR2_STAGGER
  for (int k = 0; k < (int)(threadIdx.x >> 6) + 4 * (int)((blockIdx.x >> 8) & 1u); ++k) DFEPE_STAGGER_WAIT(DFEPE_PAIR2_STAGGER);
#endif
  W8Args A;
  A.pts1 = pts1; A.pts2 = pts2; A.wts = wts; A.B = B; A.Bm = Bm; A.N = N; A.hw_sx = hw_sx; A.hw_sy = hw_sy;
  A.clamp_at = clamp_at; A.F_out = F_out; A.residual = residual; A.epi_res = R.epi_res; A.save = R.save;
  A.weights_out = R.weights_out; A.logits_mode = R.logits_mode; A.variant = R.variant; A.row_per_pair = false;
  w8pt16_fwd_pair<0, RAW, PLAIN, 2>(A, pair, xch + prow * 36, nullptr, (int)(threadIdx.x >> 4) & 1);
}

// Cooperative variant: one 256-thread workgroup (16 rows) per pair, for N > 128 (w8pt16_body.h: W8Coop).
template <int IT, bool RAW, bool PLAIN>
__global__ void __launch_bounds__(256, (IT <= 4 ? 2 : 1))  // two workgroups per CU (<= 256 registers) while the correspondences allow it
w8pt16_coop_fwd_kernel(const float* pts1, const float* pts2, const float* wts, int B, int Bm, int N, float hw_sx, float hw_sy,
                       float clamp_at, float* F_out, float* residual, const W8FwdRest R) {
  __shared__ W8Coop co;
  const int pair = (int)blockIdx.x;
  W8Args A;
  A.pts1 = pts1; A.pts2 = pts2; A.wts = wts; A.B = B; A.Bm = Bm; A.N = N; A.hw_sx = hw_sx; A.hw_sy = hw_sy;
  A.clamp_at = clamp_at; A.F_out = F_out; A.residual = residual; A.epi_res = R.epi_res; A.save = R.save;
  A.weights_out = R.weights_out; A.logits_mode = R.logits_mode; A.variant = R.variant; A.row_per_pair = false;
  w8pt16_fwd_pair<IT, RAW, PLAIN, 16>(A, pair, nullptr, &co, (int)(threadIdx.x >> 4));
}

// Fit + E-from-F + cheirality-checked pose of one pair in ONE launch (BASELINE config 5 at small batch): the cooperative fit, then
// the same workgroup decomposes pre^T F pre and triangulates its pair's correspondences (cheirality_body.h) -- no second launch,
// no second ramp, F never leaves the CU.
struct W8PoseRest {
  const float* pre;
  const float* K;
  float depth_thres;
  float* Rt_cam;
  int* winner;
  int* counts;
};
template <int IT>
__global__ void __launch_bounds__(256, (IT <= 4 ? 2 : 1))
w8pt16_coop_pose_kernel(const float* pts1, const float* pts2, const float* wts, int B, int Bm, int N, float hw_sx, float hw_sy,
                        float clamp_at, float* F_out, float* residual, const W8FwdRest R, const W8PoseRest P) {
  __shared__ W8Coop co;
  __shared__ CheirLds cl;
  const int pair = (int)blockIdx.x;
  W8Args A;
  A.pts1 = pts1; A.pts2 = pts2; A.wts = wts; A.B = B; A.Bm = Bm; A.N = N; A.hw_sx = hw_sx; A.hw_sy = hw_sy;
  A.clamp_at = clamp_at; A.F_out = F_out; A.residual = residual; A.epi_res = R.epi_res; A.save = R.save;
  A.weights_out = R.weights_out; A.logits_mode = R.logits_mode; A.variant = R.variant; A.row_per_pair = false;
  w8pt16_fwd_pair<IT, true, true, 16>(A, pair, nullptr, &co, (int)(threadIdx.x >> 4));
  float Ef[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Ef[k] = co.of[k];  // published by row 0 before the output phase: every thread has passed that barrier
  cheirality_pair(Ef, P.pre, P.K, pts1, (size_t)pair, N, P.depth_thres, P.Rt_cam, P.winner, P.counts, cl);
}

struct W8BwdRest {
  const float* F_out;
  const float* g_F;
  const float* g_res;
  const float* g_epi;
  const float* g_w_extra;
  const float* g_scale;
  float* g_w;
  float* g_p1;
  float* g_p2;
  int logits_mode;
  unsigned variant;
};

template <int IT, bool RAW, bool PGRAD, bool PLAIN, bool UP = true>
__global__ void __launch_bounds__(256)
w8pt16_bwd_kernel(const float* pts1, const float* pts2, const float* wts, int B, int Bm, int N, float hw_sx, float hw_sy,
                  float clamp_at, const float* save, const W8BwdRest R) {
  const int row = (int)(threadIdx.x >> 4);
  const int pair = (int)blockIdx.x * kPairsPerBlock + row;
  if (pair >= B) return;
#if DFEPE_BWD_STAGGER
  for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) DFEPE_STAGGER_WAIT(DFEPE_BWD_STAGGER);
#endif
  W8BwdArgs A;
  A.pts1 = pts1; A.pts2 = pts2; A.wts = wts; A.B = B; A.Bm = Bm; A.N = N; A.hw_sx = hw_sx; A.hw_sy = hw_sy;
  A.clamp_at = clamp_at; A.save = save; A.F_out = R.F_out; A.g_F = R.g_F; A.g_res = R.g_res; A.g_epi = R.g_epi;
  A.g_w_extra = R.g_w_extra; A.g_scale = R.g_scale; A.g_w = R.g_w; A.g_p1 = R.g_p1; A.g_p2 = R.g_p2;
  A.logits_mode = R.logits_mode; A.variant = R.variant; A.pending_head = nullptr; A.row_per_pair = false;
  w8pt16_bwd_pair_impl<IT, RAW, PGRAD, PLAIN, 1, UP>(A, pair, nullptr);
}

template <int IT, bool RAW>
__global__ void __launch_bounds__(256, (IT <= 4 ? 2 : 1))
w8pt16_coop_bwd_kernel(const float* pts1, const float* pts2, const float* wts, int B, int Bm, int N, float hw_sx, float hw_sy,
                       float clamp_at, const float* save, const W8BwdRest R) {
  __shared__ W8BwdCoop co;
  const int pair = (int)blockIdx.x;
  W8BwdArgs A;
  A.pts1 = pts1; A.pts2 = pts2; A.wts = wts; A.B = B; A.Bm = Bm; A.N = N; A.hw_sx = hw_sx; A.hw_sy = hw_sy;
  A.clamp_at = clamp_at; A.save = save; A.F_out = R.F_out; A.g_F = R.g_F; A.g_res = R.g_res; A.g_epi = R.g_epi;
  A.g_w_extra = R.g_w_extra; A.g_scale = R.g_scale; A.g_w = R.g_w; A.g_p1 = R.g_p1; A.g_p2 = R.g_p2;
  A.logits_mode = R.logits_mode; A.variant = 0u; A.pending_head = nullptr; A.row_per_pair = false;
  w8pt16_bwd_pair_impl<IT, RAW, false, true, 16>(A, pair, nullptr, &co, (int)(threadIdx.x >> 4));
}

// The same backward fit with three more wavefronts per workgroup; in workgroup 0 they run the loss head that dfepe_loss_tail
// deferred (loss_head_body.h: one wavefront per kind of partial), elsewhere they leave at once.  The head is off the step's
// critical path this way: nothing in the backward needs its scalars; the 448-thread workgroup (two wavefronts on three of the
// SIMDs) limits this ONE launch of the step to 256 registers.
template <int IT, bool RAW, bool UP = true>
__global__ void __launch_bounds__(448)
w8pt16_bwd_head_kernel(const float* pts1, const float* pts2, const float* wts, int B, int Bm, int N, float hw_sx, float hw_sy,
                       float clamp_at, const float* save, const W8BwdRest R, const TailHead* __restrict__ head) {
  __shared__ TailHeadLds lds;
  if (blockIdx.x == 0) {  // uniform over the workgroup: all seven wavefronts are still alive here
    if (threadIdx.x == 0) lds.arrived = 0u;
    __syncthreads();
  }
  if (threadIdx.x >= 256u) {
    if (blockIdx.x == 0) {
      const TailHead H = *head;
      loss_head_run(H, (int)(threadIdx.x & 63u), (int)(threadIdx.x >> 6) - 4, &lds);
    }
    return;
  }
  const int row = (int)(threadIdx.x >> 4);
  const int pair = (int)blockIdx.x * kPairsPerBlock + row;
  if (pair >= B) return;
#if DFEPE_BWD_STAGGER
  for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) DFEPE_STAGGER_WAIT(DFEPE_BWD_STAGGER);
#endif
  W8BwdArgs A;
  A.pts1 = pts1; A.pts2 = pts2; A.wts = wts; A.B = B; A.Bm = Bm; A.N = N; A.hw_sx = hw_sx; A.hw_sy = hw_sy;
  A.clamp_at = clamp_at; A.save = save; A.F_out = R.F_out; A.g_F = R.g_F; A.g_res = R.g_res; A.g_epi = R.g_epi;
  A.g_w_extra = R.g_w_extra; A.g_scale = R.g_scale; A.g_w = R.g_w; A.g_p1 = R.g_p1; A.g_p2 = R.g_p2;
  A.logits_mode = R.logits_mode; A.variant = 0u; A.pending_head = nullptr; A.row_per_pair = false;
  w8pt16_bwd_pair_impl<IT, RAW, false, true, 1, UP>(A, pair, nullptr);
}

template <bool RAW, bool PLAIN>
void launch_fwd(const W8Args& A, hipStream_t st) {
  const dim3 grid((A.B + kPairsPerBlock - 1) / kPairsPerBlock), block(256);
  const int N = A.N;
  W8FwdRest R;
  R.epi_res = A.epi_res; R.save = A.save; R.weights_out = A.weights_out; R.logits_mode = A.logits_mode; R.variant = A.variant;
#define DFEPE_FWD(IT_)                                                                                                     \
  hipLaunchKernelGGL((w8pt16_fwd_kernel<IT_, RAW, PLAIN>), grid, block, 0, st, A.pts1, A.pts2, A.wts, A.B, A.Bm, A.N, A.hw_sx, \
                     A.hw_sy, A.clamp_at, A.F_out, A.residual, R)
  if (use_coop_fwd(N, A.B, A.row_per_pair, RAW)) {  // one workgroup per pair: 16 rows x IT correspondences per lane
    const dim3 cgrid(A.B);
#define DFEPE_CFWD(IT_)                                                                                                    \
  hipLaunchKernelGGL((w8pt16_coop_fwd_kernel<IT_, RAW, PLAIN>), cgrid, block, 0, st, A.pts1, A.pts2, A.wts, A.B, A.Bm, A.N, A.hw_sx, \
                     A.hw_sy, A.clamp_at, A.F_out, A.residual, R)
    if (N <= 512) DFEPE_CFWD(2);
    else if (N <= 1024) DFEPE_CFWD(4);
    else DFEPE_CFWD(8);
#undef DFEPE_CFWD
  } else if (RAW && N > 128 && use_pair2(A.B)) {  // two rows of a wavefront per pair (pixel matches: the homogeneous-point
    if constexpr (RAW) {                         // instantiations need 300 registers and keep the row kernel)
      const dim3 grid2((A.B + kPairsPerBlock2 - 1) / kPairsPerBlock2);
      hipLaunchKernelGGL((w8pt16_pair2_fwd_kernel<true, PLAIN>), grid2, block, 0, st, A.pts1, A.pts2, A.wts, A.B, A.Bm, A.N, A.hw_sx, A.hw_sy,
                         A.clamp_at, A.F_out, A.residual, R);
    }
  } else if (N > 128) DFEPE_FWD(0);  // any N: correspondences re-read per phase
  else if (N <= 16) DFEPE_FWD(1);
  else if (N <= 32) DFEPE_FWD(2);
  else if (N <= 64) DFEPE_FWD(4);
  else if (use_lean(A.B)) {  // 65 .. 128 correspondences at >= kLeanMinPairs pairs: the <= 256-register build, two wavefronts per SIMD
#define DFEPE_LFWD(IT_)                                                                                                    \
  hipLaunchKernelGGL((w8pt16_fwd_lean_kernel<IT_, RAW, PLAIN>), grid, block, 0, st, A.pts1, A.pts2, A.wts, A.B, A.Bm, A.N, A.hw_sx, \
                     A.hw_sy, A.clamp_at, A.F_out, A.residual, R)
    if (N <= 112) DFEPE_LFWD(7);
    else DFEPE_LFWD(8);
#undef DFEPE_LFWD
  } else if (N <= 112) DFEPE_FWD(7);
  else DFEPE_FWD(8);
#undef DFEPE_FWD
}

template <bool RAW, bool PGRAD, bool PLAIN>
void launch_bwd(const W8BwdArgs& A, hipStream_t st) {
  const dim3 grid((A.B + kPairsPerBlock - 1) / kPairsPerBlock), block(256);
  const int N = A.N;
  W8BwdRest R;
  R.F_out = A.F_out; R.g_F = A.g_F; R.g_res = A.g_res; R.g_epi = A.g_epi; R.g_w_extra = A.g_w_extra; R.g_scale = A.g_scale;
  R.g_w = A.g_w; R.g_p1 = A.g_p1; R.g_p2 = A.g_p2; R.logits_mode = A.logits_mode; R.variant = A.variant;
#define DFEPE_BWD(IT_)                                                                                                     \
  hipLaunchKernelGGL((w8pt16_bwd_kernel<IT_, RAW, PGRAD, PLAIN>), grid, block, 0, st, A.pts1, A.pts2, A.wts, A.B, A.Bm, A.N, A.hw_sx, \
                     A.hw_sy, A.clamp_at, A.save, R)
  if constexpr (!PGRAD && PLAIN) {
    if (use_coop(N, A.B, A.row_per_pair)) {
      const dim3 cgrid(A.B);
#define DFEPE_CBWD(IT_)                                                                                                    \
  hipLaunchKernelGGL((w8pt16_coop_bwd_kernel<IT_, RAW>), cgrid, block, 0, st, A.pts1, A.pts2, A.wts, A.B, A.Bm, A.N, A.hw_sx, A.hw_sy, \
                     A.clamp_at, A.save, R)
      if (N <= 512) DFEPE_CBWD(2);
      else if (N <= 1024) DFEPE_CBWD(4);
      else DFEPE_CBWD(8);
#undef DFEPE_CBWD
      return;
    }
  }
  if constexpr (!PGRAD && PLAIN) {
    if (!A.g_res && !A.g_epi && !A.g_w_extra) {  // g_F only: the instantiation without pass A and its loads
#define DFEPE_BWD0(IT_)                                                                                                    \
  hipLaunchKernelGGL((w8pt16_bwd_kernel<IT_, RAW, false, true, false>), grid, block, 0, st, A.pts1, A.pts2, A.wts, A.B, A.Bm, A.N, A.hw_sx, \
                     A.hw_sy, A.clamp_at, A.save, R)
      if (N > 128) DFEPE_BWD0(0);
      else if (N <= 16) DFEPE_BWD0(1);
      else if (N <= 32) DFEPE_BWD0(2);
      else if (N <= 64) DFEPE_BWD0(4);
      else if (N <= 112) DFEPE_BWD0(7);
      else DFEPE_BWD0(8);
#undef DFEPE_BWD0
      return;
    }
  }
  if (N > 128) DFEPE_BWD(0);
  else if (N <= 16) DFEPE_BWD(1);
  else if (N <= 32) DFEPE_BWD(2);
  else if (N <= 64) DFEPE_BWD(4);
  else if (N <= 112) DFEPE_BWD(7);
  else DFEPE_BWD(8);
#undef DFEPE_BWD
}

// with the deferred loss head riding along.  Built for the ONE shape that uses it: pixel matches and g_F only (the captured
// solver-only step of pipeline.hot_path_fused; 250 registers at N = 100).  The 448-thread workgroup caps the launch at 256
// registers, and the instantiations with pass A (g_residual / g_epi, the recurrent model's backward) or homogeneous points need
// more (they spilled 24..208 bytes of scratch in round 3): those shapes take the plain launch plus a head launch of its own.
void launch_bwd_head(const W8BwdArgs& A, hipStream_t st) {
  const dim3 grid((A.B + kPairsPerBlock - 1) / kPairsPerBlock), block(448);
  const int N = A.N;
  W8BwdRest R;
  R.F_out = A.F_out; R.g_F = A.g_F; R.g_res = A.g_res; R.g_epi = A.g_epi; R.g_w_extra = A.g_w_extra; R.g_scale = A.g_scale;
  R.g_w = A.g_w; R.g_p1 = A.g_p1; R.g_p2 = A.g_p2; R.logits_mode = A.logits_mode; R.variant = 0u;
  const TailHead* head = static_cast<const TailHead*>(A.pending_head);
#define DFEPE_BWDH(IT_)                                                                                                    \
  hipLaunchKernelGGL((w8pt16_bwd_head_kernel<IT_, true, false>), grid, block, 0, st, A.pts1, A.pts2, A.wts, A.B, A.Bm, A.N, A.hw_sx, \
                     A.hw_sy, A.clamp_at, A.save, R, head)
  if (N > 128) DFEPE_BWDH(0);
  else if (N <= 16) DFEPE_BWDH(1);
  else if (N <= 32) DFEPE_BWDH(2);
  else if (N <= 64) DFEPE_BWDH(4);
  else if (N <= 112) DFEPE_BWDH(7);
  else DFEPE_BWDH(8);
#undef DFEPE_BWDH
}

}  // namespace

// Called by dfepe_w8pt_fwd / dfepe_w8pt_bwd (w8pt_fwd.hip / w8pt_bwd.hip) after argument validation.
int dfepe_w8pt16_fwd_launch(const W8Args& A, bool raw, hipStream_t st) {
  const bool plain = A.variant == 0;
  if (raw) { if (plain) launch_fwd<true, true>(A, st); else launch_fwd<true, false>(A, st); }
  else { if (plain) launch_fwd<false, true>(A, st); else launch_fwd<false, false>(A, st); }
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

int dfepe_loss_head_from_workspace(const void* workspace_desc, hipStream_t st);  // loss_tail.hip

int dfepe_w8pt16_bwd_launch(const W8BwdArgs& A, bool raw, hipStream_t st) {
  const bool pgrad = A.g_p1 != nullptr;
  if (A.variant != 0u) {  // un-normalised rows: weight gradients only, no riding loss head
    if (pgrad) return DFEPE_ERR_UNSUPPORTED;
    if (raw) launch_bwd<true, false, false>(A, st); else launch_bwd<false, false, false>(A, st);
    if (hipGetLastError() != hipSuccess) return DFEPE_ERR_HIP;
    return (A.pending_head != nullptr) ? dfepe_loss_head_from_workspace(A.pending_head, st) : DFEPE_OK;
  }
  if (A.pending_head != nullptr && raw && !pgrad && !A.g_res && !A.g_epi && !A.g_w_extra && !use_coop(A.N, A.B, A.row_per_pair)) {
    launch_bwd_head(A, st);
    return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
  }
  if (raw) { if (pgrad) launch_bwd<true, true, true>(A, st); else launch_bwd<true, false, true>(A, st); }
  else { if (pgrad) launch_bwd<false, true, true>(A, st); else launch_bwd<false, false, true>(A, st); }
  if (hipGetLastError() != hipSuccess) return DFEPE_ERR_HIP;
  return (A.pending_head != nullptr) ? dfepe_loss_head_from_workspace(A.pending_head, st) : DFEPE_OK;
}

// ---- fit + E-from-F + cheirality-checked pose --------------------------------------------------------------------------------
extern "C" int dfepe_w8pt_fwd(const float* pts1, const float* pts2, const float* weights, int B, int N, int n_weight_sets, unsigned flags,
                              float image_w, float image_h, float clamp_at, float* F_out, float* residual, float* epi_res, float* save,
                              float* weights_out, void* stream);

extern "C" int dfepe_w8pt_pose_fwd(const float* matches, const float* weights, int B, int N, unsigned flags, float image_w, float image_h,
                                   float clamp_at, const float* K, const float* pre, float depth_thres, float* F_out, float* residual,
                                   float* epi_res, float* weights_out, float* Rt_cam, int* winner, int* counts, void* workspace,
                                   void* stream) {
  if (!(flags & DFEPE_W8PT_RAW_MATCHES) || (flags & ~(DFEPE_W8PT_RAW_MATCHES | DFEPE_W8PT_LOGITS | DFEPE_W8PT_ROW_PER_PAIR))) return DFEPE_ERR_INVALID_ARG;
  if (B < 0 || N <= 0) return DFEPE_ERR_INVALID_ARG;
  if (B == 0) return DFEPE_OK;
  if (!matches || !weights || !K || !F_out || !residual || !Rt_cam || !(image_w > 0.f && image_h > 0.f)) return DFEPE_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(matches) & 15u) return DFEPE_ERR_INVALID_ARG;
  // A/B switch (measurement only, read once): DFEPE_POSE_LAUNCHES=2 forces the two launches, =1 the fused one where it exists
  static const int forced_launches = [] { const char* e = getenv("DFEPE_POSE_LAUNCHES"); return e ? atoi(e) : 0; }();
  if (forced_launches == 2 || !use_coop_fwd(N, B, (flags & DFEPE_W8PT_ROW_PER_PAIR) != 0, true)) {  // any other shape: the two launches this one replaces
    const int rc = dfepe_w8pt_fwd(matches, nullptr, weights, B, N, 1, flags, image_w, image_h, clamp_at, F_out, residual, epi_res, nullptr,
                                  weights_out, stream);
    if (rc != DFEPE_OK) return rc;
    return dfepe_cheirality_ex(F_out, pre, K, matches, B, N, depth_thres, 0u, workspace, Rt_cam, winner, counts, stream);
  }
  W8FwdRest R;
  R.epi_res = epi_res; R.save = nullptr; R.weights_out = weights_out; R.logits_mode = (flags & DFEPE_W8PT_LOGITS) ? 1 : 0; R.variant = 0u;
  W8PoseRest P;
  P.pre = pre; P.K = K; P.depth_thres = depth_thres; P.Rt_cam = Rt_cam; P.winner = winner; P.counts = counts;
  const dim3 grid(B), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float sx = 2.0f / image_w, sy = 2.0f / image_h;
  if (N <= 512) hipLaunchKernelGGL((w8pt16_coop_pose_kernel<2>), grid, block, 0, st, matches, nullptr, weights, B, B, N, sx, sy, clamp_at, F_out, residual, R, P);
  else if (N <= 1024) hipLaunchKernelGGL((w8pt16_coop_pose_kernel<4>), grid, block, 0, st, matches, nullptr, weights, B, B, N, sx, sy, clamp_at, F_out, residual, R, P);
  else hipLaunchKernelGGL((w8pt16_coop_pose_kernel<8>), grid, block, 0, st, matches, nullptr, weights, B, B, N, sx, sy, clamp_at, F_out, residual, R, P);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
