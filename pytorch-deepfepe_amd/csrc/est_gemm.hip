// est_gemm -- the per-correspondence weight estimator on the matrix cores.
//
// Replaces: the Conv1d(k=1) -> InstanceNorm1d(affine) -> LeakyReLU stack of ErrorEstimator
// (deepFEPE/models/ErrorEstimators.py:47-64), called depth times per step (deepFEPE/models/DeepFNet.py:441,510), forward and
// backward.  Every 1x1 convolution is a GEMM  Y[C_out, cols] = W[C_out, C_in] X[C_in, cols]  over cols = pairs x N points.
//
// Precision: the reference trains this stack in fp32.  16-bit MFMA is 16x the fp32 MFMA rate, so every fp32 operand is carried as a
// sum of 16-bit PLANES, exact remainders of each other.
//   forward (round 5): TWO fp16 planes, a = a0 + a1 (a0 = fp16(a), a1 = fp16(a - a0): 22 mantissa bits), a product = a0 b0 + a0 b1 +
//     a1 b0 -- three MFMAs, fp32 accumulate, error ~2^-22 per product: the fp32 class (scripts/proto_split_fp16.py: logits 1e-6 from
//     the fp64 truth against 2.4-3.5e-6 for stock fp32; measured on the GPU 2.0-2.4e-6 with the fp32 accumulation).  Rounds 3-4 used
//     three bf16 planes and six products for the same accuracy: twice the matrix work, 6 instead of 4 bytes per element read.  fp16's
//     narrow exponent is handled where it bites: weights (1 / sqrt(fan-in): their low plane would be subnormal) are split scaled by a
//     power of two found on the device (est_absmax_kernel, wscale) and the accumulators scaled back; activations are O(1) behind an
//     InstanceNorm, |a| <= |gamma| sqrt(N - 1) + |beta| must stay below 65504 (beyond it the result is inf / NaN, loudly).
//   backward: two bf16 planes (i + j <= 1, three MFMAs, ~2^-16): gradients need far less than the activations the next fit is
//     sensitive to, and bf16 keeps fp32's exponent range -- no loss scaling.  A forward that will be differentiated therefore also
//     leaves each activation as two bf16 planes (est_in_bwd / est_gemm_tn read those); one that will not skips them.
//
// Layout: activations POINT-major and K-BLOCKED:  P[plane][C/32][col][32]  (the 32 channels of a K step are contiguous per
// column, and the columns of a K step are contiguous: the 208 x 32 operand tile of a block is ONE contiguous 13 KB run, so
// every LDS-DMA instruction moves a full contiguous KiB -- row-major [col][C] made each instruction touch 64 different lines
// and the address unit, not the matrix cores, set the pace), weights W[plane][C_in/32][C_out][32] alike.  Kernels:
//   est_gemm_nt   C[m][n] = sum_terms A_i[m][:] . B_j[n][:]   (both operands K-contiguous), 128 x 208 block tile = 128 channels
//                 x two whole pairs (N = 100: 2 x 100 columns + 8 of the next block, recomputed there), K step 32, operands staged
//                 by LDS-DMA (global_load_lds_dwordx4) as [plane][row][4 chunks of 8 k] with the chunk order XOR-permuted per group
//                 of four rows (on the SOURCE address: the LDS image of a DMA is lane-linear), so that an MFMA fragment is one
//                 conflict-free ds_read_b128; single LDS stage, two workgroups per CU overlap each other's load and MFMA phases.
//                 Epilogues: EPI_F32 (plain fp32 store, the data-gradient GEMM) and EPI_IN (forward: InstanceNorm statistics of
//                 each (channel, pair) straight from the accumulators -- a pair's 100 columns sit in the 16 lanes of a DPP row
//                 across 7 column tiles --, affine, LeakyReLU, split into three planes, 8-byte stores; the convolution bias
//                 cancels in the normalisation).
//   est_gemm_tn   dW[co][ci] = sum_cols dY[col][co] X[col][ci]: both operands are K-major here, fragments come from
//                 ds_read_b64_tr_b16 (the LDS transpose read) of an XOR-swizzled [k][128 channels] image, split-K over columns.
//   est_in_bwd    InstanceNorm + LeakyReLU adjoint in the point-major layout (x^ and the activation sign recovered from the
//                 stored planes), writes dY as two planes; streams at 5 TB/s.  Since round 5 only the layer under the head (and any
//                 N != 100) runs it as a launch of its own: below that the adjoint is the data-gradient GEMM's epilogue (EPI_INBWD:
//                 dA stays in the accumulators, the layer's output planes are read twice -- sums, then dY -- 228 registers, no
//                 scratch; round 3's first attempt held x^ and spilled).  dA never reaches memory: 8 of the 20 bytes per element.
//   est_norm_fwd_n / est_in_bwd_n   the same normalisation + activation + split, and its adjoint, for ANY number of points per
//                 pair behind the plain product (EPI_F32): a workgroup per (pair, 64 channels), or the pair's rows over several
//                 workgroups in two launches when a dozen pairs would not fill the chip.
//   est_head_*    the last Conv1d(256 -> O) and its adjoint pieces (a GEMV per output channel: VALU, HBM-bound).
#include "dfepe_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef unsigned short bf16_t;  // storage type of a plane element (bf16 or fp16 bits)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) short frag8;  // an MFMA operand fragment: eight 16-bit values of either format
enum { FMT_BF16 = 0, FMT_F16 = 1 };

template <int FMT>
__device__ __forceinline__ f32x4 mfma16(const frag8& a, const frag8& b, const f32x4& c) {
  if constexpr (FMT == FMT_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#define DFEPE_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define DFEPE_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int kPts = 100;          // points per pair the fused epilogues are built for (N of BASELINE configs 2-4)
constexpr int BM = 128;            // channels per block tile
constexpr int NT = 13;             // 16-column tiles per block: 2 pairs = 200 columns + 8
constexpr int BN = NT * 16;        // 208
constexpr int BSTEP = 2 * kPts;    // columns a block owns
constexpr int BK = 32;             // K step = one MFMA 16x16x32

__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a DPP row, in every lane
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  v += dpp_f32<0x140>(v);
  return v;
}

// fp32 -> three bf16 planes (round-to-nearest-even each, remainders exact in fp32), two values at a time
__device__ __forceinline__ void split3(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
  f32x2 v = {x, y};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  v -= __builtin_convertvector(h, f32x2);
  const bf16x2 m = __builtin_convertvector(v, bf16x2);
  v -= __builtin_convertvector(m, f32x2);
  const bf16x2 l = __builtin_convertvector(v, bf16x2);
  p0 = __builtin_bit_cast(unsigned, h); p1 = __builtin_bit_cast(unsigned, m); p2 = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split2(float x, float y, unsigned& p0, unsigned& p1) {
  f32x2 v = {x, y};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  v -= __builtin_convertvector(h, f32x2);
  const bf16x2 m = __builtin_convertvector(v, bf16x2);
  p0 = __builtin_bit_cast(unsigned, h); p1 = __builtin_bit_cast(unsigned, m);
}
// fp32 -> two fp16 planes (round-to-nearest-even each; 22 mantissa bits while the low plane stays in fp16's normal range,
// an absolute 2^-25 below it; |x| > 65504 overflows to inf / NaN -- loudly)
__device__ __forceinline__ void split2h(float x, float y, unsigned& p0, unsigned& p1) {
  f32x2 v = {x, y};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  v -= __builtin_convertvector(h, f32x2);
  const f16x2 m = __builtin_convertvector(v, f16x2);
  p0 = __builtin_bit_cast(unsigned, h); p1 = __builtin_bit_cast(unsigned, m);
}
__device__ __forceinline__ float f16_lo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
__device__ __forceinline__ float f16_hi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
// The forward products run on fp16 planes.  Weights are small (1 / sqrt(fan-in)): unscaled, their low plane would sit in fp16's
// subnormal range and lose bits, so a layer's weights are split as W * s with s the power of two that brings max |W| into [8, 16)
// (exact), and the product's accumulators are scaled back before anything else reads them.  `bits` = the fp32 bit pattern of max |W|
// (est_absmax_kernel); exponent clamped so that s and 1 / s^2 stay normal fp32 numbers.
__device__ __forceinline__ float wscale(unsigned bits, bool inverse) {
  int eb = (int)((bits >> 23) & 255u);
  eb = eb < 100 ? 100 : (eb > 150 ? 150 : eb);
  return __uint_as_float((unsigned)(inverse ? eb - 3 : 257 - eb) << 23);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// element (row, ch) of a K-blocked plane with `rows` rows
__device__ __forceinline__ size_t kb_index(size_t row, int ch, size_t rows) { return ((size_t)(ch >> 5) * rows + row) * 32 + (ch & 31); }
// chunk permutation of the staged [row][4 chunks] image: rows 4r..4r+3 of a 16-row tile hold chunk kg at position kg ^ f(r),
// f = (0, 2, 3, 1): each of the four 16-lane service groups of a ds_read_b128 then covers all 16 slots of the bank row
__device__ __forceinline__ int chunk_swz(int rowgroup) { return (0x78 >> (2 * (rowgroup & 3))) & 3; }

enum { EPI_F32 = 0, EPI_IN = 1, EPI_INBWD = 2 };
#ifndef DFEPE_NT2_BLOCKS
#define DFEPE_NT2_BLOCKS 3  // workgroups per CU the two-plane (data-gradient) product is compiled for
#endif
#ifndef DFEPE_NT_SPLIT
#define DFEPE_NT_SPLIT 0  // 1: est_gemm_nt fills its LDS stage in two halves, each under the other half's MFMAs, instead of issue / wait /
                          // multiply -- in EVERY build.  Measured at full grids (round 5, B = 4096, same box, twice): 8.18-8.20 against
                          // 8.15-8.16 ms per estimator call, forward alone 2.42 against 2.45 ms -- with two or three workgroups per CU
                          // the cycles between DMA issue and barrier (half of the K loop by the phase stamps of
                          // scripts/ubench/est_phases.hip) are already covered by the other workgroups' MFMAs.  At SMALL grids
                          // (DFEPE_NT_SPLIT_SMALL: the AHEAD = 2 builds) it is worth 15-20 %: B = 8, rocprofv3 --stats: forward layers
                          // 26.8 -> 22.7 us on average (1024 -> 512: 59.6 -> 47.0), the fused data gradient 31.2 -> 28.8
#endif
#ifndef DFEPE_NT_PAIR_TILES
#define DFEPE_NT_PAIR_TILES 0  // 1: the small-grid builds walk the column tiles in pairs (four accumulators in rotation, four fragment sets).
                               // Measured at B = 8 (rocprofv3 --stats, same box): no better than tile by tile for the fused layers and 70 % WORSE for
                               // the plain data gradient (17.8 vs 10.6 us): a lone wavefront's MFMAs are not waiting for their accumulators
#endif
#ifndef DFEPE_NT_TWO_STAGE
#define DFEPE_NT_TWO_STAGE 1  // the small-grid builds (AHEAD = 2: at most one workgroup per CU) double-buffer the whole LDS stage (86 KB)
#endif
#ifndef DFEPE_NT_SPLIT_SMALL
#define DFEPE_NT_SPLIT_SMALL 1  // ... in the small-grid builds (AHEAD = 2), where a workgroup has its CU to itself -- superseded there by
                                // DFEPE_NT_TWO_STAGE (two whole stages: 22.0 / 27.3 us); this is what they fall back to with it off
#endif
#ifndef DFEPE_INBWD_DEPTH
#define DFEPE_INBWD_DEPTH 4  // column tiles of the layer's output in flight in the fused adjoint's two passes
#endif
#ifndef DFEPE_FWD_BLOCKS
#define DFEPE_FWD_BLOCKS 3  // workgroups per CU the two-plane forward layer (fused epilogue) is compiled for: 168 registers, 43 KB of LDS each;
                            // measured at B = 4096: forward 2.49 ms against 2.67 with two (one fragment set then, fetched per tile, like the data gradient)
#endif

// -DDFEPE_EST_PHASE_CLOCKS (scripts/ubench/est_phases.hip only): lane 0 of every wavefront stamps the shader clock at the phase
// boundaries of est_gemm_nt_kernel and sums its waits in the K loop
#ifdef DFEPE_EST_PHASE_CLOCKS
__device__ unsigned long long* g_est_phase_clk;  // [workgroup * 4 + wavefront][8]
#define EST_STAMP(slot)                                                                                                  \
  do {                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    const unsigned long long t_ = __builtin_readcyclecounter();                                                          \
    if ((threadIdx.x & 63u) == 0u)                                                                                       \
      g_est_phase_clk[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 8 + (slot)] = t_;      \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  } while (0)
#else
#define EST_STAMP(slot) do {} while (0)
#endif

struct EpiArgs {
  // EPI_F32: out[col][m] fp32, ld = ldc; with gridDim.z = S > 1 (split-K) slice z of the K steps goes to out + z * split_stride
  float* out;
  int ldc;
  size_t split_stride;
  // EPI_F32, gx != null: the product is the gradient w.r.t. the estimator's input -- stored as gx[pair][ch][n], ch < gx_C0, col = pair * gx_N + n
  float* gx;
  int gx_C0, gx_N;
  long gx_sb, gx_sc;  // element strides of gx between pairs / channels (points are contiguous): [pairs][C0][N] dense, or the channel-major
                      // [C0][pairs][N] buffer the model keeps its estimator inputs in
  // EPI_IN
  const float* gamma;
  const float* beta;
  float eps, slope;
  bf16_t* planes;        // [3][ncols][M] bf16 (FMT_BF16) or [2][ncols][M] fp16 (FMT_F16): the next layer's operand
  size_t plane_stride;   // elements between planes
  float* rstd;           // [npairs][M]
  // FMT_F16
  const unsigned* absmax;  // bits of max |W| (the weights were split scaled, see wscale); null: unscaled
  bf16_t* planes_bwd;      // [2][ncols][M] bf16: what the backward reads (est_in_bwd, est_gemm_tn); null: not kept (no gradient wanted)
  size_t bwd_stride;
  // EPI_INBWD (the data gradient dA = dY_next W_next stays in the accumulators and goes straight through the InstanceNorm + LeakyReLU
  // adjoint of the layer below): gamma, beta, slope as above; planes / plane_stride = dY out [2][ncols][M] bf16
  const bf16_t* aout;      // that layer's output as the backward reads it, [2][ncols][M] bf16
  size_t aout_stride;
  const float* rstd_in;    // [npairs][M]
  float* dgamma_part;      // [npairs][M] each
  float* dbeta_part;
};

// C[m][n] = sum over plane pairs (i, j), i + j <= ORDER, of A_i[m][:] . B_j[n][:]
//   A: [NPA][M][K] (a_plane elements between planes), B: [NPB][ncols][K]; K % 32 == 0, M % 4 == 0.
// Block (bx, by): columns [200 bx, 200 bx + 208), channels [128 by, 128 by + 128); 4 wavefronts, wavefront w owns channels
// 32 w .. 32 w + 31 (two 16-row MFMA tiles) x all 13 column tiles: 26 accumulator tiles = 104 registers.
// One LDS stage, two workgroups per CU: while one multiplies, the other waits for its LDS-DMA.  Measured alternatives at
// B = 4096 (1024 -> 512 layer, 2.16 ms as built): two LDS stages with one workgroup per CU (DMA of step k + 1 under the MFMAs of
// step k) 2.68 ms -- a lone wavefront per SIMD issues its 16 DMA instructions, 45 fragment reads and 156 MFMAs in order;
// weights straight from global memory into a second register set + two LDS stages of the activations only, still two workgroups
// per CU: 1.47 vs 1.43 ms with two planes (206 registers), spills with three.  The step is bound by instruction issue around the
// MFMAs (DMA setup, fragment reads, barriers), not by an exposed load latency.
// AHEAD: column tiles whose B fragments are fetched ahead of the MFMAs that consume them.  -1: what the (planes, epilogue) combination
// is tuned for at a full grid -- 0 for the two-plane products compiled for three workgroups per CU (the third wavefront on the SIMD
// covers the LDS latency), 1 otherwise.  2: the build for SMALL grids (the reference's batch sizes: 16-64 workgroups on 256 CUs):
// a wavefront alone on its SIMD sees every LDS round trip (13 tiles x ~120 cycles against 96 cycles of MFMAs per tile with AHEAD = 0).
constexpr int nt_blocks(int NPA, int NPB, int EPI, int AHEAD) {
  return (AHEAD >= 0) ? 2 : ((NPA == 2 && NPB == 2 && EPI == EPI_F32) ? DFEPE_NT2_BLOCKS : ((NPA == 2 && NPB == 2 && EPI == EPI_IN) ? DFEPE_FWD_BLOCKS : 2));
}
template <int NPA, int NPB, int ORDER, int EPI, int FMT = FMT_BF16, int AHEAD = -1>
__global__ void __launch_bounds__(256, nt_blocks(NPA, NPB, EPI, AHEAD))
est_gemm_nt_kernel(const bf16_t* __restrict__ A, size_t a_plane, const bf16_t* __restrict__ B, size_t b_plane, int M, int ncols, int K,
                   const EpiArgs E) {
  constexpr int kABytes = NPA * BM * 64, kBBytes = NPB * BN * 64;
  // the small-grid builds (AHEAD = 2) own their CU: TWO stages, the next K step's DMA under the whole MFMA phase of this one
  constexpr bool kTwoStage = (AHEAD == 2) && (DFEPE_NT_TWO_STAGE != 0);
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[(kTwoStage ? 2 : 1) * (kABytes + kBBytes)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  // XCD-aware order: consecutive workgroup ids go round-robin to the 8 XCDs; the channel blocks of one column block share its
  // activation tile, so they are given consecutive slots of ONE XCD (its L2 then serves the re-reads)
  const int mblocks = (int)gridDim.y, cblocks = (int)gridDim.x;
  int bx = (int)blockIdx.x, by = (int)blockIdx.y;
  if ((cblocks & 7) == 0) {
    const int id = by * cblocks + bx;  // linear dispatch order (x fastest)
    const int xcd = id & 7, s = id >> 3;
    by = s % mblocks;
    bx = (s / mblocks) * 8 + xcd;
  }
  const int m0 = by * BM, n0 = bx * BSTEP;

  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging tasks of this wavefront: one contiguous KiB (16 rows x 64 B of one plane's K block) per LDS-DMA instruction ----
  constexpr int kATasks = NPA * (BM / 16), kBTasks = NPB * NT, kTasks = kATasks + kBTasks;
  constexpr int kPerWave = (kTasks + 3) / 4;
  const int drow = lane >> 2, dchunk = (lane & 3) ^ chunk_swz(lane >> 4);  // DMA role: row within the instruction, source chunk
  const int fsw = chunk_swz(c >> 2);                                       // fragment role: this lane's rows sit in row group c >> 2
  // Row i of the wavefront's A tile mt holds channel 8 (i >> 2) + 4 mt + (i & 3) of its 32 (not 16 mt + i): an accumulator lane
  // (row group g) then owns the EIGHT consecutive channels 8 g .. 8 g + 7 across its two tiles, so the epilogue stores 16 bytes per
  // lane and plane, 64 contiguous bytes per column, one contiguous KiB per store instruction
  const int fswA[2] = {chunk_swz(2 * (c >> 2)), chunk_swz(2 * (c >> 2) + 1)};
  // split-K (EPI_F32 only): workgroup z of gridDim.z walks the K steps [ks_begin, ks_end)
  const int nk_all = K / BK, kz = (int)blockIdx.z, nz = (int)gridDim.z;
  const int ks_begin = (nk_all * kz) / nz, nk = (nk_all * (kz + 1)) / nz;
  // The stage is filled in two groups -- group 0: the A rows and the column tiles [0, kHalf) of B; group 1: the column tiles
  // [kHalf, NT) -- so that the DMA of one half runs under the MFMAs of the other (DFEPE_NT_SPLIT, below).
  constexpr int kHalf = 7;
  auto issue_task_a = [&](int ks, unsigned char* lds, int t) {
    const int p = t / (BM / 16), rb = t % (BM / 16);
    int row = m0 + rb * 16 + drow;
    row = (row < M) ? row : M - 1;
    const bf16_t* src = A + (size_t)p * a_plane + ((size_t)ks * M + row) * 32 + dchunk * 8;
    unsigned char* dst = lds + (p * BM + rb * 16) * 64;
    __builtin_amdgcn_global_load_lds(DFEPE_GLOBAL_PTR(src), DFEPE_LDS_PTR(dst), 16, 0, 0);
  };
  auto issue_task_b = [&](int ks, unsigned char* lds, int p, int rb) {
    int row = n0 + rb * 16 + drow;
    row = (row < ncols) ? row : ncols - 1;
    const bf16_t* src = B + (size_t)p * b_plane + ((size_t)ks * ncols + row) * 32 + dchunk * 8;
    unsigned char* dst = lds + kABytes + (p * BN + rb * 16) * 64;
    __builtin_amdgcn_global_load_lds(DFEPE_GLOBAL_PTR(src), DFEPE_LDS_PTR(dst), 16, 0, 0);
  };
  auto stage_issue = [&](int ks, unsigned char* lds) {
#pragma unroll
    for (int i = 0; i < kPerWave; ++i) {
      const int t = wave + 4 * i;  // wave-uniform
      if (t < kATasks) issue_task_a(ks, lds, t);
      else if (t < kTasks) issue_task_b(ks, lds, (t - kATasks) / NT, (t - kATasks) % NT);
    }
  };
  auto group_issue = [&](int ks, unsigned char* lds, auto grp) {
    constexpr int G = decltype(grp)::value;
    constexpr int kTiles = G == 0 ? kHalf : NT - kHalf, kFirst = G == 0 ? 0 : kHalf;
    constexpr int kN = (G == 0 ? kATasks : 0) + NPB * kTiles;
#pragma unroll
    for (int i = 0; i < (kN + 3) / 4; ++i) {
      const int t = wave + 4 * i;
      if (G == 0 && t < kATasks) issue_task_a(ks, lds, t);
      else if (t < kN) {
        const int u = t - (G == 0 ? kATasks : 0);
        issue_task_b(ks, lds, u / kTiles, kFirst + u % kTiles);
      }
    }
  };
  auto a_from_lds = [&](const unsigned char* lds, frag8 (&a)[2][NPA]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int p = 0; p < NPA; ++p)
        a[mt][p] = *reinterpret_cast<const frag8*>(lds + ((p * BM + wave * 32 + 8 * (c >> 2) + 4 * mt + (c & 3)) * 4 + (g ^ fswA[mt])) * 16);
  };
  auto mfma_phase = [&](const unsigned char* lds, const frag8 (&a)[2][NPA], auto first_tile, auto end_tile) {
    constexpr int kT0 = decltype(first_tile)::value, kT1 = decltype(end_tile)::value;
    // the column tile's B fragments are fetched one tile ahead of the MFMAs that consume them (two register sets): the LDS
    // latency of tile nt + 1 hides behind the twelve MFMAs of tile nt instead of stalling every tile
    // (the two-plane product compiled for three workgroups per CU has 168 registers: one fragment set, fetched per tile -- the
    // third wavefront on the SIMD covers the LDS latency the second set would)
    constexpr bool kPairs = (AHEAD == 2) && (DFEPE_NT_PAIR_TILES != 0);
    constexpr int kD = kPairs ? 3 : ((AHEAD >= 0) ? AHEAD : ((nt_blocks(NPA, NPB, EPI, AHEAD) > 2) ? 0 : 1));  // tiles fetched ahead
    frag8 b[kD + 1][NPB];
    auto fetch_b = [&](int nt, frag8 (&dst)[NPB]) {
#pragma unroll
      for (int p = 0; p < NPB; ++p) dst[p] = *reinterpret_cast<const frag8*>(lds + kABytes + ((p * BN + nt * 16 + c) * 4 + (g ^ fsw)) * 16);
    };
#pragma unroll
    for (int d = 0; d < (kPairs ? 2 : kD); ++d)
      if (kT0 + d < kT1) fetch_b(kT0 + d, b[(kT0 + d) % (kD + 1)]);
    if constexpr (kPairs) {
      // a wavefront alone on its SIMD: two column tiles at a time, FOUR accumulators in rotation (an accumulator comes round every
      // fourth MFMA instead of every second)
#pragma unroll
      for (int nt = kT0; nt < kT1; nt += 2) {
        const bool two = nt + 1 < kT1;
        if (nt + 2 < kT1) fetch_b(nt + 2, b[(nt + 2) % 4]);  // the next pair's fragments, into the two sets this pair does not use
        if (nt + 3 < kT1) fetch_b(nt + 3, b[(nt + 3) % 4]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ord = ORDER; ord >= 0; --ord)
#pragma unroll
          for (int i = 0; i <= ord; ++i) {
            const int j = ord - i;
            if (i < NPA && j < NPB) {
#pragma unroll
              for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = mfma16<FMT>(a[mt][i], b[nt % (kD + 1)][j], acc[mt][nt]);
              if (two) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt][nt + 1] = mfma16<FMT>(a[mt][i], b[(nt + 1) % (kD + 1)][j], acc[mt][nt + 1]);
              }
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int nt = kT0; nt < kT1; ++nt) {
      if (nt + kD < kT1) fetch_b(nt + kD, b[(nt + kD) % (kD + 1)]);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this tile's MFMAs (the scheduler would sink it to its first use)
      // smallest terms first; the two row tiles alternate, so that no MFMA waits for the one issued just before it
#pragma unroll
      for (int ord = ORDER; ord >= 0; --ord)
#pragma unroll
        for (int i = 0; i <= ord; ++i) {
          const int j = ord - i;
          if (i < NPA && j < NPB) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              acc[mt][nt] = mfma16<FMT>(a[mt][i], b[nt % (kD + 1)][j], acc[mt][nt]);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  };

  EST_STAMP(0);
#ifdef DFEPE_EST_PHASE_CLOCKS
  unsigned long long wait_cycles = 0, mfma_cycles = 0;
#endif
  using T0 = std::integral_constant<int, 0>;
  using TH = std::integral_constant<int, kHalf>;
  using TN = std::integral_constant<int, NT>;
  if constexpr (kTwoStage) {
    constexpr int kStage = kABytes + kBBytes;
    stage_issue(ks_begin, lds_all);
    for (int ks = ks_begin; ks < nk; ++ks) {
      unsigned char* cur = lds_all + ((ks - ks_begin) & 1) * kStage;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this step's stage
      __syncthreads();                                  // ... everyone's, and everyone is done reading the other buffer (step ks - 1)
      if (ks + 1 < nk) stage_issue(ks + 1, lds_all + ((ks + 1 - ks_begin) & 1) * kStage);
      frag8 a[2][NPA];
      a_from_lds(cur, a);
      mfma_phase(cur, a, T0{}, TN{});
    }
  } else if constexpr (DFEPE_NT_SPLIT || (DFEPE_NT_SPLIT_SMALL && AHEAD == 2)) {
    // One LDS stage, filled in two halves: while the MFMAs of column tiles [0, 7) run, the DMA of tiles [7, 13) of the same K step is
    // in flight; while those of [7, 13) run, the DMA of the NEXT step's A rows and tiles [0, 7).  Two barriers per step, as before
    // (each one both releases a half for overwriting and publishes the other half's arrival).
    group_issue(ks_begin, lds_all, T0{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    group_issue(ks_begin, lds_all, std::integral_constant<int, 1>{});
    for (int ks = ks_begin; ks < nk; ++ks) {
#ifdef DFEPE_EST_PHASE_CLOCKS
      const unsigned long long t0_ = __builtin_readcyclecounter();
#endif
      frag8 a[2][NPA];
      a_from_lds(lds_all, a);
      mfma_phase(lds_all, a, T0{}, TH{});
#ifdef DFEPE_EST_PHASE_CLOCKS
      const unsigned long long t1_ = __builtin_readcyclecounter();
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tiles [7, 13) of this step
      __syncthreads();
#ifdef DFEPE_EST_PHASE_CLOCKS
      const unsigned long long t2_ = __builtin_readcyclecounter();
#endif
      if (ks + 1 < nk) group_issue(ks + 1, lds_all, T0{});
      mfma_phase(lds_all, a, TH{}, TN{});
#ifdef DFEPE_EST_PHASE_CLOCKS
      const unsigned long long t3_ = __builtin_readcyclecounter();
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next step's first half
      __syncthreads();
      if (ks + 1 < nk) group_issue(ks + 1, lds_all, std::integral_constant<int, 1>{});
#ifdef DFEPE_EST_PHASE_CLOCKS
      const unsigned long long t4_ = __builtin_readcyclecounter();
      wait_cycles += (t2_ - t1_) + (t4_ - t3_); mfma_cycles += (t1_ - t0_) + (t3_ - t2_);
#endif
    }
  } else {
  for (int ks = ks_begin; ks < nk; ++ks) {
#ifdef DFEPE_EST_PHASE_CLOCKS
    const unsigned long long t0_ = __builtin_readcyclecounter();
#endif
    stage_issue(ks, lds_all);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef DFEPE_EST_PHASE_CLOCKS
    const unsigned long long t1_ = __builtin_readcyclecounter();
#endif
    frag8 a[2][NPA];
    a_from_lds(lds_all, a);
    mfma_phase(lds_all, a, T0{}, TN{});
#ifdef DFEPE_EST_PHASE_CLOCKS
    const unsigned long long t2_ = __builtin_readcyclecounter();
    wait_cycles += t1_ - t0_; mfma_cycles += t2_ - t1_;
#endif
    __syncthreads();
  }
  }
  EST_STAMP(1);
#ifdef DFEPE_EST_PHASE_CLOCKS
  if ((threadIdx.x & 63u) == 0u) {
    unsigned long long* q = g_est_phase_clk + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 8;
    q[5] = wait_cycles; q[6] = mfma_cycles;
  }
#endif

  // ---- epilogue: lane holds, per column n0 + 16 nt + c, channels ch8 + 4 mt + r (mt = 0, 1; r = 0..3): eight in a row --------
  const int ch8 = m0 + wave * 32 + 8 * g;
  const bool chok = ch8 < M;  // M % 8 == 0
  float unscale = 1.0f;  // FMT_F16: the weights were split scaled by a power of two
  if constexpr (FMT == FMT_F16) unscale = E.absmax ? wscale(*E.absmax, true) : 1.0f;
  if constexpr (EPI == EPI_F32) {
    if (E.gx != nullptr) {  // (uniform) the input gradient, channel-major per pair like x itself: 16 consecutive points per channel and store
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int cl = nt * 16 + c, col = n0 + cl;
        if (cl < BSTEP && col < ncols) {
          const int pr = col / E.gx_N, pt = col - pr * E.gx_N;
          float* dst = E.gx + (size_t)pr * E.gx_sb + (size_t)ch8 * E.gx_sc + pt;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (ch8 + j < E.gx_C0) dst[(size_t)j * E.gx_sc] = acc[j >> 2][nt][j & 3] * unscale;
        }
      }
      return;
    }
    float* const out_z = E.out + (size_t)kz * E.split_stride;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int cl = nt * 16 + c, col = n0 + cl;
      if (cl < BSTEP && col < ncols && chok) {
        float* dst = out_z + (size_t)col * E.ldc + ch8;
        if constexpr (FMT == FMT_F16) {
          *reinterpret_cast<f32x4*>(dst) = acc[0][nt] * unscale;
          *reinterpret_cast<f32x4*>(dst + 4) = acc[1][nt] * unscale;
        } else {
          *reinterpret_cast<f32x4*>(dst) = acc[0][nt];
          *reinterpret_cast<f32x4*>(dst + 4) = acc[1][nt];
        }
      }
    }
  } else if constexpr (EPI == EPI_INBWD) {
    // The accumulators hold dA[col][ch] of TWO whole pairs x this lane's eight channels: est_in_bwd_kernel's arithmetic on them in
    // place.  Pass A: a = the layer's output (two bf16 planes, one 16-byte load per plane and column tile), dz = dA lrelu'(a) written
    // back into the accumulator, x^ = (z - beta) / gamma recovered from a, the two sums per (channel, pair) accumulated over the
    // pair's column tiles and then over the 16 lanes of the DPP row.  Pass B: x^ recomputed from the same planes (L2), dY = rstd gamma
    // (dz - mean(dz) - x^ mean(dz x^)), two bf16 planes, 16-byte stores.  dA never reaches memory.
    const bool t6p0 = c < 4, t12ok = c < 8;
    const int npairs = ncols / kPts;
    const int pair0 = 2 * bx, pair1 = (pair0 + 1 < npairs) ? pair0 + 1 : pair0;
    const int chc = chok ? ch8 : 0;
    const float islope = 1.0f / E.slope;
    // z from a = lrelu(z) without a compare (a mask per element kept for a second use is what spilled the first build of this
    // epilogue): a slope below one makes the pre-activation the smaller of a and a / slope, one above it the larger
    // (the median of a, a / slope and -inf resp. +inf: one instruction)
    const float med_lim = (E.slope <= 1.0f) ? -__builtin_inff() : __builtin_inff();
    auto unrelu = [&](float a) { return __builtin_amdgcn_fmed3f(a, a * islope, med_lim); };
    float ig[8], bt[8];
    {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(E.gamma + chc), g1 = *reinterpret_cast<const f32x4*>(E.gamma + chc + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(E.beta + chc), b1 = *reinterpret_cast<const f32x4*>(E.beta + chc + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gj = j < 4 ? g0[j & 3] : g1[j & 3];
        ig[j] = (fabsf(gj) > 1e-30f) ? 1.0f / gj : 0.0f;
        bt[j] = j < 4 ? b0[j & 3] : b1[j & 3];
      }
    }
    // the raw words of one column tile (two planes x 16 bytes), fetched one tile ahead of their use; the scheduling barriers keep the
    // compiler from hoisting all thirteen tiles' loads (104 registers) above the loop
    struct Raw { uint4 u0, u1; };
    size_t plane_at = kb_index(0, chc, (size_t)ncols);  // of this lane's channel group, column 0
    auto fetch = [&](int nt) {
      int cc = c;
      asm volatile("" : "+v"(cc));  // the address arithmetic stays with its tile (hoisted, the 26 addresses are 52 registers)
      int col = n0 + nt * 16 + cc;
      col = (col < ncols) ? col : ncols - 1;
      const size_t at = plane_at + (size_t)col * 32;
      Raw r;
      r.u0 = *reinterpret_cast<const uint4*>(E.aout + at);
      r.u1 = *reinterpret_cast<const uint4*>(E.aout + E.aout_stride + at);
      return r;
    };
    auto unpack = [&](const Raw& r, float (&a)[8]) {
      const unsigned w0[4] = {r.u0.x, r.u0.y, r.u0.z, r.u0.w}, w1[4] = {r.u1.x, r.u1.y, r.u1.z, r.u1.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[2 * q] = bf16_lo(w0[q]) + bf16_lo(w1[q]);
        a[2 * q + 1] = bf16_hi(w0[q]) + bf16_hi(w1[q]);
      }
    };
    float s1[2][8], s2[2][8];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[p][j] = 0.f; s2[p][j] = 0.f; }
    // kDepth tiles in flight: one iteration is ~80 vector instructions, a miss to HBM an order of magnitude more
    constexpr int kDepth = DFEPE_INBWD_DEPTH;
    Raw ring[kDepth];
#pragma unroll
    for (int i = 0; i < kDepth; ++i) ring[i] = fetch(i);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const Raw cur = ring[nt % kDepth];
      if (nt + kDepth < NT) ring[nt % kDepth] = fetch(nt + kDepth);
      __builtin_amdgcn_sched_barrier(0);
      float a[8];
      unpack(cur, a);
      const bool live = (nt < 12) || t12ok;  // lanes 8..15 of tile 12 belong to the next block
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = acc[j >> 2][nt][j & 3];
        const float e = live ? ((a[j] > 0.f) ? d : d * E.slope) : 0.f;
        const float x = unrelu(a[j]);  // z; x^ = (z - beta) / gamma is affine in it: sum dz x^ = (sum dz z - beta sum dz) / gamma, below
        acc[j >> 2][nt][j & 3] = e;
        if (nt < 6) { s1[0][j] += e; s2[0][j] = fmaf(e, x, s2[0][j]); }
        else if (nt > 6) { s1[1][j] += e; s2[1][j] = fmaf(e, x, s2[1][j]); }
        else {
          const float e0 = t6p0 ? e : 0.f, e1 = t6p0 ? 0.f : e;
          s1[0][j] += e0; s2[0][j] = fmaf(e0, x, s2[0][j]);
          s1[1][j] += e1; s2[1][j] = fmaf(e1, x, s2[1][j]);
        }
      }
      // pin the sums to this iteration: left free, instruction selection orders the pure accumulation chains after ALL thirteen
      // tiles' x^ (104 more live values), and the scheduling barriers then keep them there
#pragma unroll
      for (int p = 0; p < 2; ++p)
        if ((p == 0 && nt <= 6) || (p == 1 && nt >= 6))
          asm volatile("" : "+v"(s1[p][0]), "+v"(s1[p][1]), "+v"(s1[p][2]), "+v"(s1[p][3]), "+v"(s1[p][4]), "+v"(s1[p][5]), "+v"(s1[p][6]), "+v"(s1[p][7]),
                            "+v"(s2[p][0]), "+v"(s2[p][1]), "+v"(s2[p][2]), "+v"(s2[p][3]), "+v"(s2[p][4]), "+v"(s2[p][5]), "+v"(s2[p][6]), "+v"(s2[p][7]));
      __builtin_amdgcn_sched_barrier(0);
    }
    EST_STAMP(2);
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s1[p][j] = row16_sum(s1[p][j]);
        s2[p][j] = (row16_sum(s2[p][j]) - bt[j] * s1[p][j]) * ig[j];  // sum dz x^  (0 where gamma == 0, like est_in_bwd)
      }
    if (chok && c == 0) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int pr = pair0 + p;
        if (pr < npairs) {
          float* db = E.dbeta_part + (size_t)pr * M + ch8;
          float* dg = E.dgamma_part + (size_t)pr * M + ch8;
          *reinterpret_cast<f32x4*>(db) = f32x4{s1[p][0], s1[p][1], s1[p][2], s1[p][3]};
          *reinterpret_cast<f32x4*>(db + 4) = f32x4{s1[p][4], s1[p][5], s1[p][6], s1[p][7]};
          *reinterpret_cast<f32x4*>(dg) = f32x4{s2[p][0], s2[p][1], s2[p][2], s2[p][3]};
          *reinterpret_cast<f32x4*>(dg + 4) = f32x4{s2[p][4], s2[p][5], s2[p][6], s2[p][7]};
        }
      }
    }
    float kk[2][8];
    {
      const float inv_n = 1.0f / (float)kPts;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(E.gamma + chc), g1 = *reinterpret_cast<const f32x4*>(E.gamma + chc + 4);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float* rp = E.rstd_in + (size_t)(p ? pair1 : pair0) * M + chc;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // dY = k (dz - m1 - x^ m2), x^ = (z - beta) / gamma:  dY = k dz + c1 z + c0 -- two FMAs per element in pass B
          const float k = (j < 4 ? r0[j & 3] : r1[j & 3]) * (j < 4 ? g0[j & 3] : g1[j & 3]);
          const float m1 = s1[p][j] * inv_n, m2g = s2[p][j] * inv_n * ig[j];
          kk[p][j] = k;
          s2[p][j] = -k * m2g;                   // c1
          s1[p][j] = k * (m2g * bt[j] - m1);     // c0
        }
      }
    }
    // the planes are read AGAIN for pass B (L2): without the compiler barrier the second loads are merged with the first and all
    // 104 values of a stay live across the row sums (the first build spilled 130 registers that way)
    EST_STAMP(3);
    asm volatile("" : "+v"(plane_at) :: "memory");  // (and the addresses formed again: kept, they were 52 registers, spilled)
#pragma unroll
    for (int i = 0; i < kDepth; ++i) ring[i] = fetch(i);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const Raw cur = ring[nt % kDepth];
      if (nt + kDepth < NT) ring[nt % kDepth] = fetch(nt + kDepth);
      __builtin_amdgcn_sched_barrier(0);
      float a[8];
      unpack(cur, a);
      const bool first = (nt < 6) || (nt == 6 && t6p0);
      float y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c0 = first ? s1[0][j] : s1[1][j], c1 = first ? s2[0][j] : s2[1][j], k = first ? kk[0][j] : kk[1][j];
        y[j] = fmaf(k, acc[j >> 2][nt][j & 3], fmaf(c1, unrelu(a[j]), c0));
      }
      unsigned pl[2][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) split2(y[2 * q], y[2 * q + 1], pl[0][q], pl[1][q]);
      const int cl = nt * 16 + c, col = n0 + cl;
      if (cl < BSTEP && col < ncols && chok) {
        bf16_t* dst = E.planes + kb_index((size_t)col, ch8, (size_t)ncols);
        *reinterpret_cast<uint4*>(dst) = make_uint4(pl[0][0], pl[0][1], pl[0][2], pl[0][3]);
        *reinterpret_cast<uint4*>(dst + E.plane_stride) = make_uint4(pl[1][0], pl[1][1], pl[1][2], pl[1][3]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    EST_STAMP(4);
  } else {
    // pair 0 = columns 0..99 (tiles 0..5 and lanes 0..3 of tile 6), pair 1 = 100..199 (lanes 4..15 of tile 6, tiles 7..11,
    // lanes 0..7 of tile 12); lanes 8..15 of tile 12 belong to the next block
    const float inv_n = 1.0f / (float)kPts;
    const float inv_n2 = inv_n * unscale * unscale;
    const bool t6p0 = c < 4, t12ok = c < 8;
    const int pair0 = 2 * bx;
    const int chc = chok ? ch8 : 0;
    f32x4 gam[2], bet[2], mean0[2], mean1[2], rs0[2], rs1[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      gam[mt] = *reinterpret_cast<const f32x4*>(E.gamma + chc + 4 * mt);
      bet[mt] = *reinterpret_cast<const f32x4*>(E.beta + chc + 4 * mt);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) s0 += acc[mt][nt][r];
        s0 += t6p0 ? acc[mt][6][r] : 0.f;
        s1 += t6p0 ? 0.f : acc[mt][6][r];
#pragma unroll
        for (int nt = 7; nt < 12; ++nt) s1 += acc[mt][nt][r];
        s1 += t12ok ? acc[mt][12][r] : 0.f;
        const float mu0 = row16_sum(s0) * inv_n, mu1 = row16_sum(s1) * inv_n;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) { const float d = acc[mt][nt][r] - mu0; q0 = fmaf(d, d, q0); }
        { const float d = acc[mt][6][r] - (t6p0 ? mu0 : mu1); const float dd = d * d; q0 += t6p0 ? dd : 0.f; q1 += t6p0 ? 0.f : dd; }
#pragma unroll
        for (int nt = 7; nt < 12; ++nt) { const float d = acc[mt][nt][r] - mu1; q1 = fmaf(d, d, q1); }
        { const float d = acc[mt][12][r] - mu1; q1 += t12ok ? d * d : 0.f; }
        mean0[mt][r] = mu0; mean1[mt][r] = mu1;
        // biased variance, like F.instance_norm.  FMT_F16: the accumulators are s x the product (s a power of two: mean and
        // deviations scale exactly), so the variance is q / s^2 and the normalised value (acc - mean) (rstd / s)
        rs0[mt][r] = 1.0f / sqrtf(row16_sum(q0) * inv_n2 + E.eps);
        rs1[mt][r] = 1.0f / sqrtf(row16_sum(q1) * inv_n2 + E.eps);
      }
      if (chok && c == 0) {
        if ((size_t)(pair0) * kPts < (size_t)ncols) *reinterpret_cast<f32x4*>(E.rstd + (size_t)pair0 * M + ch8 + 4 * mt) = rs0[mt];
        if ((size_t)(pair0 + 1) * kPts < (size_t)ncols) *reinterpret_cast<f32x4*>(E.rstd + (size_t)(pair0 + 1) * M + ch8 + 4 * mt) = rs1[mt];
      }
      // z = (v - mean) * (rstd gamma) + beta  (the difference first: no cancellation against a large mean)
#pragma unroll
      for (int r = 0; r < 4; ++r) { rs0[mt][r] *= gam[mt][r] * unscale; rs1[mt][r] *= gam[mt][r] * unscale; }
    }
    EST_STAMP(2);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bool first = (nt < 6) || (nt == 6 && t6p0);
      const int cl = nt * 16 + c, col = n0 + cl;
      unsigned pl[4][4];  // FMT_BF16: three bf16 planes; FMT_F16: two fp16 planes (forward) + two bf16 planes (backward)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float z = fmaf(acc[mt][nt][r] - (first ? mean0[mt][r] : mean1[mt][r]), first ? rs0[mt][r] : rs1[mt][r], bet[mt][r]);
          v[r] = (z > 0.f) ? z : z * E.slope;
        }
        if constexpr (FMT == FMT_F16) {
          split2h(v[0], v[1], pl[0][2 * mt], pl[1][2 * mt]);
          split2h(v[2], v[3], pl[0][2 * mt + 1], pl[1][2 * mt + 1]);
          if (E.planes_bwd) {  // uniform
            split2(v[0], v[1], pl[2][2 * mt], pl[3][2 * mt]);
            split2(v[2], v[3], pl[2][2 * mt + 1], pl[3][2 * mt + 1]);
          }
        } else {
          split3(v[0], v[1], pl[0][2 * mt], pl[1][2 * mt], pl[2][2 * mt]);
          split3(v[2], v[3], pl[0][2 * mt + 1], pl[1][2 * mt + 1], pl[2][2 * mt + 1]);
        }
      }
      if (cl < BSTEP && col < ncols && chok) {
        const size_t at = kb_index((size_t)col, ch8, (size_t)ncols);
        bf16_t* dst = E.planes + at;
#pragma unroll
        for (int p = 0; p < (FMT == FMT_F16 ? 2 : 3); ++p) *reinterpret_cast<uint4*>(dst + p * E.plane_stride) = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
        if constexpr (FMT == FMT_F16) {
          if (E.planes_bwd) {
            bf16_t* bw = E.planes_bwd + at;
#pragma unroll
            for (int p = 0; p < 2; ++p) *reinterpret_cast<uint4*>(bw + p * E.bwd_stride) = make_uint4(pl[2 + p][0], pl[2 + p][1], pl[2 + p][2], pl[2 + p][3]);
          }
        }
      }
    }
    EST_STAMP(4);
  }
}

// ---- dW[co][ci] = sum_cols dY[col][co] X[col][ci], two planes each (three products) ------------------------------------------
// Block: 128 x 128 output tile (2 x 2 wavefronts of 64 x 64 = 4 x 4 MFMA tiles), one slice of the columns (split-K); the partial
// product goes to part[slice][co][ci] (summed by the caller: deterministic, no floating-point atomics).
// LDS image per (operand, plane): [4 channel blocks][32 k][4 chunks of 8 channels] = the K-blocked global layout as it is (each
// LDS-DMA instruction copies one contiguous KiB: 16 columns of one channel block); chunk position q of column k holds channel
// chunk q ^ 2 ((k >> 2) & 1)  (the swizzle lives on the SOURCE address, the image itself is lane-linear), so that the eight
// columns a ds_read_b64_tr_b16 half-wave touches fall on all 64 banks.
// (the body is shared by the one-problem launch and the table-driven one of round 6: `id` = the workgroup's linear index among the
// nx * ny * nslices of its problem)
__device__ __forceinline__ void est_gemm_tn_body(const bf16_t* __restrict__ dY, size_t dy_plane, int Cout, const bf16_t* __restrict__ X,
                                                 size_t x_plane, int Cin, int ncols, int cols_per_slice, float* __restrict__ part, int id,
                                                 int nx, int ny, int nslices) {
  constexpr int kOpBytes = 2 * BK * 256;  // two planes of [32][256 B]
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kOpBytes];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware order (-DDFEPE_TN_NO_XCD builds the launch-order variant for A/B timing): consecutive workgroup ids go round-robin to the 8 XCDs; all output tiles of one column slice
  // stream the same columns of dY and X, so a slice's tiles are given consecutive slots of ONE XCD and its L2 serves the re-reads
  // (in launch order the eight Cout tiles of a (Cin tile, slice) sat on eight different XCDs: X crossed the fabric eight times)
  int bx = id % nx, by = (id / nx) % ny, slice = id / (nx * ny);
#ifndef DFEPE_TN_NO_XCD
  if ((nslices & 7) == 0) {
    const int tiles = nx * ny;
    const int xcd = id & 7, j = id >> 3, t = j % tiles;
    slice = (j / tiles) * 8 + xcd;
    bx = t % nx; by = t / nx;
  }
#endif
  const int m0 = bx * 128, n0 = by * 128;
  const int kbeg = slice * cols_per_slice;
  int kend = kbeg + cols_per_slice;
  kend = (kend < ncols) ? kend : ncols;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int dcol = lane >> 2, dq = (lane & 3) ^ (2 * ((lane >> 4) & 1));  // DMA role: column within the instruction, source chunk
  const int c = lane & 15, g = lane >> 4;
  const int tr_row = (c >> 2), tr_piece = c & 3;  // ds_read_tr role inside the 16-lane group
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    // 2 operands x 2 planes x 4 channel blocks x 2 column halves = 32 DMA instructions per stage, 8 per wavefront
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = wave * 8 + i;  // wave-uniform: operand = t >> 4, plane = (t >> 3) & 1, channel block = (t >> 1) & 3, half = t & 1
      const int op = t >> 4, p = (t >> 3) & 1, cb = (t >> 1) & 3, half = t & 1;
      int col = k0 + half * 16 + dcol;
      col = (col < kend) ? col : kend - 1;  // past the slice: a copy of the last column, zeroed below
      const int C = op ? Cin : Cout, base = op ? n0 : m0;
      int ch = base + cb * 32;
      ch = (ch < C) ? ch : 0;  // channel blocks past the operand: any valid address, masked at the store
      const bf16_t* src = (op ? X + (size_t)p * x_plane : dY + (size_t)p * dy_plane) + kb_index((size_t)col, ch, (size_t)ncols) + dq * 8;
      unsigned char* dst = lds + op * kOpBytes + p * (BK * 256) + cb * (BK * 64) + half * 16 * 64;
      __builtin_amdgcn_global_load_lds(DFEPE_GLOBAL_PTR(src), DFEPE_LDS_PTR(dst), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (k0 + BK > kend) {  // ragged last step of the slice: rows past the end hold a copy of the last column -- zero them
      for (int e = tid; e < 2 * kOpBytes / 16; e += 256) {
        const int k = (e >> 2) & (BK - 1);  // image [operand][plane][channel block][32 columns][4 chunks]
        if (k0 + k >= kend) reinterpret_cast<uint4*>(lds)[e] = make_uint4(0, 0, 0, 0);
      }
      __syncthreads();
    }
    // one 16-channel fragment (8 k-values per lane: rows 4g..4g+3 and 16+4g..16+4g+3 of the stage) through the transposing read
    auto frag = [&](int op, int p, int t) {
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      // 16-channel tile t: channel block t >> 1, chunks 2 (t & 1), 2 (t & 1) + 1 of the column's 64 bytes
      const unsigned char* base = lds + op * kOpBytes + p * (BK * 256) + (t >> 1) * (BK * 64) + tr_piece * 8;
      const int ka = 4 * g + tr_row, kb = 16 + 4 * g + tr_row;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(base + ka * 64 + (((2 * (t & 1)) ^ (2 * ((ka >> 2) & 1))) * 16)));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(base + kb * 64 + (((2 * (t & 1)) ^ (2 * ((kb >> 2) & 1))) * 16)));
      return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    bf16x8 a[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int p = 0; p < 2; ++p) a[mt][p] = frag(0, p, wr * 4 + mt);
    bf16x8 b[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) b[0][p] = frag(1, p, wc * 4);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      if (nt + 1 < 4) {
#pragma unroll
        for (int p = 0; p < 2; ++p) b[(nt + 1) & 1][p] = frag(1, p, wc * 4 + nt + 1);
      }
      __builtin_amdgcn_sched_barrier(0);  // the next tile's fragments stay ahead of this tile's MFMAs
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][1], b[nt & 1][0], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][0], b[nt & 1][1], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][0], b[nt & 1][0], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  float* dst = part + (size_t)slice * Cout * Cin;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int ci = n0 + wc * 64 + nt * 16 + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = m0 + wr * 64 + mt * 16 + 4 * g + r;
        if (co < Cout && ci < Cin) dst[(size_t)co * Cin + ci] = acc[mt][nt][r];
      }
    }
}
__global__ void __launch_bounds__(256, 2)
est_gemm_tn_kernel(const bf16_t* __restrict__ dY, size_t dy_plane, int Cout, const bf16_t* __restrict__ X, size_t x_plane, int Cin,
                   int ncols, int cols_per_slice, float* __restrict__ part) {
  const int nx = (int)gridDim.x, ny = (int)gridDim.y;
  est_gemm_tn_body(dY, dy_plane, Cout, X, x_plane, Cin, ncols, cols_per_slice, part, (int)blockIdx.x + nx * ((int)blockIdx.y + ny * (int)blockIdx.z),
                   nx, ny, (int)gridDim.z);
}
// every layer's weight gradient of one backward in ONE launch (round 6): over few columns a backward keeps all its dY alive to the end
// anyway (the gamma == 0 fix), and five launches of 1-32 tiles each become one grid that fills the chip
struct TnTab {
  int n;
  const bf16_t* dY[8]; const bf16_t* X[8];
  size_t dy_plane[8], x_plane[8];
  int Cout[8], Cin[8], nx[8], ny[8], slices[8], cps[8];
  float* part[8];
  int first_block[9];  // multiples of 8: a problem's linear ids keep their XCD phase
};
__global__ void __launch_bounds__(256, 2) est_gemm_tn_multi_kernel(const TnTab T, int ncols) {
  int l = 0;
  while (l + 1 < T.n && (int)blockIdx.x >= T.first_block[l + 1]) ++l;
  const int id = (int)blockIdx.x - T.first_block[l];
  if (id >= T.nx[l] * T.ny[l] * T.slices[l]) return;  // padding up to the next multiple of 8
  est_gemm_tn_body(T.dY[l], T.dy_plane[l], T.Cout[l], T.X[l], T.x_plane[l], T.Cin[l], ncols, T.cps[l], T.part[l], id, T.nx[l], T.ny[l], T.slices[l]);
}

// ---- InstanceNorm + LeakyReLU adjoint, point-major ----------------------------------------------------------------------------
// One workgroup = one pair (100 columns) x 64 channels: 32 channel pairs x 8 row groups of 13 columns.  Inputs: the upstream
// gradient dA [cols][C] fp32 -- or, for the layer under the head, its rank-one form dlogit[col] * w_head[c] --, the layer's
// output planes (a = lrelu(z), z = gamma x^ + beta: z and x^ are recovered from them -- from the two leading planes, 2^-17 of a: the
// third would be 2 of the 14 bytes per element this HBM-bound kernel moves, for digits the two-plane products downstream drop),
// rstd, gamma, beta.  Outputs: dY as two planes, and the pair's contributions to d gamma / d beta ([pair][C] each, summed by the caller).
__global__ void __launch_bounds__(256)
est_in_bwd_kernel(const float* __restrict__ dA, const float* __restrict__ dlogit, const float* __restrict__ w_head,
                  const bf16_t* __restrict__ planes, size_t plane_stride, const float* __restrict__ rstd,
                  const float* __restrict__ gamma, const float* __restrict__ beta, float slope, int C, int ncols,
                  bf16_t* __restrict__ dYp, size_t dy_plane, float* __restrict__ dgamma_part, float* __restrict__ dbeta_part) {
  __shared__ float red[2][8][64];
  const int pair = (int)blockIdx.x, cb = (int)blockIdx.y * 64;
  const int cp = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int ch = cb + 2 * cp;
  const bool chok = ch < C;
  const int chc = chok ? ch : 0;
  const float g0 = gamma[chc], g1 = gamma[chc + 1], b0 = beta[chc], b1 = beta[chc + 1];
  const float ig0 = (fabsf(g0) > 1e-30f) ? 1.0f / g0 : 0.0f, ig1 = (fabsf(g1) > 1e-30f) ? 1.0f / g1 : 0.0f;
  const float islope = 1.0f / slope;
  const float r0 = rstd[(size_t)pair * C + chc], r1 = rstd[(size_t)pair * C + chc + 1];
  constexpr int RPG = 13;  // 8 x 13 >= 100
  float dz0[RPG], dz1[RPG], xh0[RPG], xh1[RPG];
  float s10 = 0.f, s11 = 0.f, s20 = 0.f, s21 = 0.f;
#pragma unroll
  for (int i = 0; i < RPG; ++i) {
    const int rl = rg * RPG + i;
    const bool live = rl < kPts;
    const size_t col = (size_t)pair * kPts + (live ? rl : 0);
    const size_t at = kb_index(col, chc, (size_t)ncols);
    const unsigned u0 = *reinterpret_cast<const unsigned*>(planes + at);
    const unsigned u1 = *reinterpret_cast<const unsigned*>(planes + plane_stride + at);
#ifdef DFEPE_INBWD_3PLANES
    const unsigned u2 = *reinterpret_cast<const unsigned*>(planes + 2 * plane_stride + at);
    const float a0 = (bf16_lo(u0) + bf16_lo(u1)) + bf16_lo(u2), a1 = (bf16_hi(u0) + bf16_hi(u1)) + bf16_hi(u2);
#else  // two planes: a to 2^-17, the class of the two-plane products this gradient feeds; the sign of a is its leading plane's
    const float a0 = bf16_lo(u0) + bf16_lo(u1), a1 = bf16_hi(u0) + bf16_hi(u1);
#endif
    float d0, d1;
    if (dA != nullptr) {
      const f32x2 d = *reinterpret_cast<const f32x2*>(dA + col * C + chc);
      d0 = d[0]; d1 = d[1];
    } else {
      const float dl = dlogit[col];
      d0 = dl * w_head[chc]; d1 = dl * w_head[chc + 1];
    }
    const float z0 = (a0 > 0.f) ? a0 : a0 * islope, z1 = (a1 > 0.f) ? a1 : a1 * islope;
    const float e0 = live ? ((a0 > 0.f) ? d0 : d0 * slope) : 0.f, e1 = live ? ((a1 > 0.f) ? d1 : d1 * slope) : 0.f;
    const float x0 = (z0 - b0) * ig0, x1 = (z1 - b1) * ig1;
    dz0[i] = e0; dz1[i] = e1; xh0[i] = x0; xh1[i] = x1;
    s10 += e0; s11 += e1; s20 = fmaf(e0, x0, s20); s21 = fmaf(e1, x1, s21);
  }
  red[0][rg][2 * cp] = s10; red[0][rg][2 * cp + 1] = s11;
  red[1][rg][2 * cp] = s20; red[1][rg][2 * cp + 1] = s21;
  __syncthreads();
  float S10 = 0.f, S11 = 0.f, S20 = 0.f, S21 = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    S10 += red[0][q][2 * cp]; S11 += red[0][q][2 * cp + 1];
    S20 += red[1][q][2 * cp]; S21 += red[1][q][2 * cp + 1];
  }
  if (rg == 0 && chok) {
    dbeta_part[(size_t)pair * C + ch] = S10; dbeta_part[(size_t)pair * C + ch + 1] = S11;
    dgamma_part[(size_t)pair * C + ch] = S20; dgamma_part[(size_t)pair * C + ch + 1] = S21;
  }
  const float inv_n = 1.0f / (float)kPts;
  const float k0 = r0 * g0, k1 = r1 * g1;
  const float m10 = S10 * inv_n, m11 = S11 * inv_n, m20 = S20 * inv_n, m21 = S21 * inv_n;
#pragma unroll
  for (int i = 0; i < RPG; ++i) {
    const int rl = rg * RPG + i;
    if (rl < kPts && chok) {
      const size_t col = (size_t)pair * kPts + rl;
      const float y0 = k0 * (dz0[i] - m10 - xh0[i] * m20), y1 = k1 * (dz1[i] - m11 - xh1[i] * m21);
      unsigned p0, p1;
      split2(y0, y1, p0, p1);
      const size_t at = kb_index(col, ch, (size_t)ncols);
      *reinterpret_cast<unsigned*>(dYp + at) = p0;
      *reinterpret_cast<unsigned*>(dYp + dy_plane + at) = p1;
    }
  }
}

// ---- any number of points per pair: InstanceNorm + LeakyReLU + split behind the plain product, and its adjoint ---------------
// The fused epilogue above is built around 100 points per pair.  For another N (the reference's SIFT configurations carry up to
// 2000 correspondences, deepFEPE/configs/*.yaml, at batch sizes of 4-12) a layer is the plain product est_gemm_nt<EPI_F32>
// followed by this kernel: one workgroup = one pair x 64 channels, 32 channel pairs x 32 row groups striding the pair's columns,
// four independent rows in flight per thread (with a dozen pairs the grid is a few hundred workgroups at most: the time is the
// chain of dependent load latencies, so the rows are spread over 1024 threads and the loads unrolled).  Three passes over the
// pair's fp32 block (mean; squared deviations from it -- the same two-pass variance as the epilogue --; normalise + activate +
// split): N x 64 x 4 bytes, L2-resident between the passes.
constexpr int kRG = 32;   // row groups of a workgroup
constexpr int kRU = 4;    // rows a thread has in flight

__device__ __forceinline__ void pair_reduce2(float (&red)[kRG][64], int rg, int cp, float a, float b, float& A, float& B) {
  __syncthreads();  // the previous reduction's readers are done
  red[rg][2 * cp] = a; red[rg][2 * cp + 1] = b;
  __syncthreads();
  A = 0.f; B = 0.f;
#pragma unroll
  for (int q = 0; q < kRG; ++q) { A += red[q][2 * cp]; B += red[q][2 * cp + 1]; }
}

// MODE 0: the whole pair in one workgroup (grid.z = 1).  With a dozen pairs that is too few workgroups, so the pair's rows can be
// split over grid.z = S workgroups and two launches: MODE 1 leaves each split's mean and sum of squared deviations from it in
// part[pair][split][2][C], MODE 2 merges the S partials (Chan's pairwise formula: M2 = sum M2_s + sum n_s (m_s - m)^2, as accurate
// as the two-pass form) and normalises its own rows.
template <int MODE>
__global__ void __launch_bounds__(1024)
est_norm_fwd_n_kernel(const float* __restrict__ Y, int ldy, int C, int N, size_t ncols, const float* __restrict__ gamma,
                      const float* __restrict__ beta, float eps, float slope, bf16_t* __restrict__ planes, size_t plane_stride,
                      bf16_t* __restrict__ planes_bwd, size_t bwd_stride, float* __restrict__ rstd, float* __restrict__ part) {
  __shared__ float red[kRG][64];
  const int pair = (int)blockIdx.x, cb = (int)blockIdx.y * 64;
  const int S = (int)gridDim.z, sp = (int)blockIdx.z;
  const int R = (N + S - 1) / S;
  const int r0 = sp * R, r1 = (r0 + R < N) ? r0 + R : N;  // this workgroup's rows (possibly none in the last split)
  const int cp = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int ch = cb + 2 * cp;
  const bool chok = ch < C;
  const int chc = chok ? ch : 0;
  const size_t col0 = (size_t)pair * N;
  const float* Yc = Y + col0 * ldy + chc;
  float mu0, mu1, Q0, Q1;
  if (MODE != 2) {
    const float inv_n = 1.0f / (float)((r1 > r0) ? r1 - r0 : 1);
    float s0 = 0.f, s1 = 0.f;
    for (int rb = r0 + rg; rb < r1; rb += kRG * kRU) {
      f32x2 y[kRU];
#pragma unroll
      for (int u = 0; u < kRU; ++u) { const int rl = rb + kRG * u; y[u] = *reinterpret_cast<const f32x2*>(Yc + (size_t)(rl < r1 ? rl : rb) * ldy); }
#pragma unroll
      for (int u = 0; u < kRU; ++u) { const bool live = rb + kRG * u < r1; s0 += live ? y[u][0] : 0.f; s1 += live ? y[u][1] : 0.f; }
    }
    float S0, S1;
    pair_reduce2(red, rg, cp, s0, s1, S0, S1);
    mu0 = S0 * inv_n; mu1 = S1 * inv_n;
    float q0 = 0.f, q1 = 0.f;
    for (int rb = r0 + rg; rb < r1; rb += kRG * kRU) {
      f32x2 y[kRU];
#pragma unroll
      for (int u = 0; u < kRU; ++u) { const int rl = rb + kRG * u; y[u] = *reinterpret_cast<const f32x2*>(Yc + (size_t)(rl < r1 ? rl : rb) * ldy); }
#pragma unroll
      for (int u = 0; u < kRU; ++u) {
        const bool live = rb + kRG * u < r1;
        const float d0 = y[u][0] - mu0, d1 = y[u][1] - mu1;
        q0 += live ? d0 * d0 : 0.f; q1 += live ? d1 * d1 : 0.f;
      }
    }
    pair_reduce2(red, rg, cp, q0, q1, Q0, Q1);
    if (MODE == 1) {
      if (rg == 0 && chok) {
        float* P = part + ((size_t)pair * S + sp) * 2 * C + ch;
        P[0] = mu0; P[1] = mu1; P[C] = Q0; P[C + 1] = Q1;
      }
      return;
    }
  } else {
    const float* P = part + (size_t)pair * S * 2 * C + chc;
    float a0 = 0.f, a1 = 0.f;
    for (int t = 0; t < S; ++t) {
      const int n_t = ((t + 1) * R < N ? (t + 1) * R : N) - t * R;
      if (n_t > 0) { a0 = fmaf((float)n_t, P[(size_t)t * 2 * C], a0); a1 = fmaf((float)n_t, P[(size_t)t * 2 * C + 1], a1); }
    }
    mu0 = a0 / (float)N; mu1 = a1 / (float)N;
    Q0 = 0.f; Q1 = 0.f;
    for (int t = 0; t < S; ++t) {
      const int n_t = ((t + 1) * R < N ? (t + 1) * R : N) - t * R;
      if (n_t > 0) {
        const float d0 = P[(size_t)t * 2 * C] - mu0, d1 = P[(size_t)t * 2 * C + 1] - mu1;
        Q0 += P[(size_t)t * 2 * C + C] + (float)n_t * d0 * d0; Q1 += P[(size_t)t * 2 * C + C + 1] + (float)n_t * d1 * d1;
      }
    }
  }
  const float inv_N = 1.0f / (float)N;
  const float rs0 = 1.0f / sqrtf(Q0 * inv_N + eps), rs1 = 1.0f / sqrtf(Q1 * inv_N + eps);  // biased variance, like F.instance_norm
  if (rg == 0 && sp == 0 && chok) { rstd[(size_t)pair * C + ch] = rs0; rstd[(size_t)pair * C + ch + 1] = rs1; }
  const float k0 = rs0 * gamma[chc], k1 = rs1 * gamma[chc + 1], b0 = beta[chc], b1 = beta[chc + 1];
  if (!chok) return;
  for (int rb = r0 + rg; rb < r1; rb += kRG * kRU) {
    f32x2 y[kRU];
#pragma unroll
    for (int u = 0; u < kRU; ++u) { const int rl = rb + kRG * u; y[u] = *reinterpret_cast<const f32x2*>(Yc + (size_t)(rl < r1 ? rl : rb) * ldy); }
#pragma unroll
    for (int u = 0; u < kRU; ++u) {
      const int rl = rb + kRG * u;
      if (rl >= r1) break;
      const float z0 = fmaf(y[u][0] - mu0, k0, b0), z1 = fmaf(y[u][1] - mu1, k1, b1);
      const float a0 = (z0 > 0.f) ? z0 : z0 * slope, a1 = (z1 > 0.f) ? z1 : z1 * slope;
      unsigned p0, p1;
      split2h(a0, a1, p0, p1);  // two fp16 planes: the next layer's operand
      const size_t at = kb_index(col0 + rl, ch, ncols);
      *reinterpret_cast<unsigned*>(planes + at) = p0;
      *reinterpret_cast<unsigned*>(planes + plane_stride + at) = p1;
      if (planes_bwd) {  // two bf16 planes: what the backward reads
        split2(a0, a1, p0, p1);
        *reinterpret_cast<unsigned*>(planes_bwd + at) = p0;
        *reinterpret_cast<unsigned*>(planes_bwd + bwd_stride + at) = p1;
      }
    }
  }
}

// The adjoint for any N: est_in_bwd_kernel's arithmetic with the pair's columns strided instead of held in registers -- one pass
// for the two sums, one that recomputes d z and x^ from the same inputs and writes dY.  MODE as above: 1 leaves the split's two
// sums in part[pair][split][2][C], 2 adds the S partials up and writes its own rows.
template <int MODE>
__global__ void __launch_bounds__(1024)
est_in_bwd_n_kernel(const float* __restrict__ dA, const float* __restrict__ dlogit, const float* __restrict__ w_head,
                    const bf16_t* __restrict__ planes, size_t plane_stride, const float* __restrict__ rstd,
                    const float* __restrict__ gamma, const float* __restrict__ beta, float slope, int C, int N, size_t ncols,
                    bf16_t* __restrict__ dYp, size_t dy_plane, float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                    float* __restrict__ part) {
  __shared__ float red[kRG][64];
  const int pair = (int)blockIdx.x, cb = (int)blockIdx.y * 64;
  const int S = (int)gridDim.z, sp = (int)blockIdx.z;
  const int R = (N + S - 1) / S;
  const int r0 = sp * R, r1 = (r0 + R < N) ? r0 + R : N;
  const int cp = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int ch = cb + 2 * cp;
  const bool chok = ch < C;
  const int chc = chok ? ch : 0;
  const float g0 = gamma[chc], g1 = gamma[chc + 1], b0 = beta[chc], b1 = beta[chc + 1];
  const float ig0 = (fabsf(g0) > 1e-30f) ? 1.0f / g0 : 0.0f, ig1 = (fabsf(g1) > 1e-30f) ? 1.0f / g1 : 0.0f;
  const float islope = 1.0f / slope;
  const float wh0 = dA ? 0.f : w_head[chc], wh1 = dA ? 0.f : w_head[chc + 1];
  const size_t col0 = (size_t)pair * N;
  struct Raw { unsigned u0, u1, u2; float d0, d1; };
  auto load = [&](int rl) {
    const size_t col = col0 + rl;
    const size_t at = kb_index(col, chc, ncols);
    Raw r;
    r.u0 = *reinterpret_cast<const unsigned*>(planes + at);
    r.u1 = *reinterpret_cast<const unsigned*>(planes + plane_stride + at);
#ifdef DFEPE_INBWD_3PLANES
    r.u2 = *reinterpret_cast<const unsigned*>(planes + 2 * plane_stride + at);
#else
    r.u2 = 0u;
#endif
    if (dA != nullptr) {
      const f32x2 d = *reinterpret_cast<const f32x2*>(dA + col * C + chc);
      r.d0 = d[0]; r.d1 = d[1];
    } else {
      const float dl = dlogit[col];
      r.d0 = dl * wh0; r.d1 = dl * wh1;
    }
    return r;
  };
  auto item = [&](const Raw& r, float& e0, float& e1, float& x0, float& x1) {
    const float a0 = (bf16_lo(r.u0) + bf16_lo(r.u1)) + bf16_lo(r.u2), a1 = (bf16_hi(r.u0) + bf16_hi(r.u1)) + bf16_hi(r.u2);
    const float z0 = (a0 > 0.f) ? a0 : a0 * islope, z1 = (a1 > 0.f) ? a1 : a1 * islope;
    e0 = (a0 > 0.f) ? r.d0 : r.d0 * slope; e1 = (a1 > 0.f) ? r.d1 : r.d1 * slope;
    x0 = (z0 - b0) * ig0; x1 = (z1 - b1) * ig1;
  };
  float S10, S11, S20, S21;
  if (MODE != 2) {
    float s10 = 0.f, s11 = 0.f, s20 = 0.f, s21 = 0.f;
    for (int rb = r0 + rg; rb < r1; rb += kRG * kRU) {
      Raw r[kRU];
#pragma unroll
      for (int u = 0; u < kRU; ++u) { const int rl = rb + kRG * u; r[u] = load(rl < r1 ? rl : rb); }
#pragma unroll
      for (int u = 0; u < kRU; ++u) {
        float e0, e1, x0, x1;
        item(r[u], e0, e1, x0, x1);
        if (rb + kRG * u >= r1) { e0 = 0.f; e1 = 0.f; }
        s10 += e0; s11 += e1; s20 = fmaf(e0, x0, s20); s21 = fmaf(e1, x1, s21);
      }
    }
    pair_reduce2(red, rg, cp, s10, s11, S10, S11);
    pair_reduce2(red, rg, cp, s20, s21, S20, S21);
    if (MODE == 1) {
      if (rg == 0 && chok) {
        float* P = part + ((size_t)pair * S + sp) * 2 * C + ch;
        P[0] = S10; P[1] = S11; P[C] = S20; P[C + 1] = S21;
      }
      return;
    }
  } else {
    const float* P = part + (size_t)pair * S * 2 * C + chc;
    S10 = 0.f; S11 = 0.f; S20 = 0.f; S21 = 0.f;
    for (int t = 0; t < S; ++t) {
      S10 += P[(size_t)t * 2 * C]; S11 += P[(size_t)t * 2 * C + 1];
      S20 += P[(size_t)t * 2 * C + C]; S21 += P[(size_t)t * 2 * C + C + 1];
    }
  }
  if (!chok) return;
  if (rg == 0 && sp == 0) {
    dbeta_part[(size_t)pair * C + ch] = S10; dbeta_part[(size_t)pair * C + ch + 1] = S11;
    dgamma_part[(size_t)pair * C + ch] = S20; dgamma_part[(size_t)pair * C + ch + 1] = S21;
  }
  const float inv_n = 1.0f / (float)N;
  const float k0 = rstd[(size_t)pair * C + ch] * g0, k1 = rstd[(size_t)pair * C + ch + 1] * g1;
  const float m10 = S10 * inv_n, m11 = S11 * inv_n, m20 = S20 * inv_n, m21 = S21 * inv_n;
  for (int rb = r0 + rg; rb < r1; rb += kRG * kRU) {
    Raw r[kRU];
#pragma unroll
    for (int u = 0; u < kRU; ++u) { const int rl = rb + kRG * u; r[u] = load(rl < r1 ? rl : rb); }
#pragma unroll
    for (int u = 0; u < kRU; ++u) {
      const int rl = rb + kRG * u;
      if (rl >= r1) break;
      float e0, e1, x0, x1;
      item(r[u], e0, e1, x0, x1);
      unsigned p0, p1;
      split2(k0 * (e0 - m10 - x0 * m20), k1 * (e1 - m11 - x1 * m21), p0, p1);
      const size_t at = kb_index(col0 + rl, ch, ncols);
      *reinterpret_cast<unsigned*>(dYp + at) = p0;
      *reinterpret_cast<unsigned*>(dYp + dy_plane + at) = p1;
    }
  }
}

// ---- round 6: the same two kernels with the pair's block RESIDENT IN REGISTERS, summing split-K partials on the way in ----------
// What the reference's own configurations need (N = 1000-2000 points, 4-12 pairs per batch, deepFEPE/configs/kitti_corr_baseline.yaml:
// 12-13) and what the K-heavy layers need at ANY N once a batch is so small that one 128 x 208 tile per workgroup leaves most CUs idle:
// the product is the plain GEMM (EPI_F32), optionally split over S slices of K (S partial products [S][cols][ld] fp32), and ONE more
// launch does everything else.  A workgroup = one pair x 32 channels (one K block of the plane layout) = 16 channel pairs x 64 row
// groups; thread (cp, rg) owns rows rg, rg + 64, ... (RPT = ceil(N / 64) of them, template: N <= 128 / 512 / 1024 / 2048) of its two
// channels: ONE pass over the fp32 block -- the partials added in slice order --, the values stay in registers for the two-pass
// statistics (mean, then squared deviations: the same arithmetic as the fused epilogue and est_norm_fwd_n) and the normalise +
// activate + split.  Against est_norm_fwd_n (three passes, two launches when the pair's rows are spread): one launch, one read.
// Geometry: CP channel pairs x RG = 1024 / CP row groups per workgroup; RPT rows per thread.  (CP, RPT) = (16, 2 / 8 / 16) serves
// N <= 128 / 512 / 1024 with 32 channels per workgroup; N <= 2048 takes (8, 16): 16 channels, because 1024 threads hold 128 registers
// each and the adjoint keeps four values per (row, channel pair).
// sums over the workgroup's row groups of NV values per thread (each channel pair's own): a wavefront holds 64 / CP row groups --
// cross-lane exchanges --, then sixteen partials per channel pair through LDS
template <int NV, int CP>
__device__ __forceinline__ void rg_all_sum(float (&red)[16][NV][CP], float (&v)[NV]) {
  const int cp = threadIdx.x & (CP - 1), w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int o = CP; o < 64; o <<= 1) v[i] += __shfl_xor(v[i], o, 64);
  }
  __syncthreads();  // the previous reduction's readers are done
  if ((threadIdx.x & 63) < CP) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[w][i][cp] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float t = red[0][i][cp];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][i][cp];
    v[i] = t;
  }
}

template <int RPT, int CP>
__global__ void __launch_bounds__(1024)
est_norm_fwd_r_kernel(const float* __restrict__ Y, int ldy, size_t split_stride, int S, int C, int N, size_t ncols,
                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float slope, bf16_t* __restrict__ planes,
                      size_t plane_stride, bf16_t* __restrict__ planes_bwd, size_t bwd_stride, float* __restrict__ rstd) {
  constexpr int RG = 1024 / CP;
  __shared__ float red[16][2][CP];
  const int pair = (int)blockIdx.x, cb = (int)blockIdx.y * (2 * CP);
  const int cp = threadIdx.x & (CP - 1), rg = threadIdx.x / CP;
  const int ch = cb + 2 * cp;  // C % 32 == 0: every channel of the workgroup exists
  const size_t col0 = (size_t)pair * N;
  const float* Yc = Y + col0 * ldy + ch;
  f32x2 v[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int rl = rg + RG * i;
    v[i] = *reinterpret_cast<const f32x2*>(Yc + (size_t)(rl < N ? rl : 0) * ldy);
  }
  for (int s = 1; s < S; ++s) {
    const float* Ys = Yc + (size_t)s * split_stride;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int rl = rg + RG * i;
      v[i] += *reinterpret_cast<const f32x2*>(Ys + (size_t)(rl < N ? rl : 0) * ldy);
    }
  }
  const float inv_N = 1.0f / (float)N;
  float a[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < RPT; ++i) { const bool live = rg + RG * i < N; a[0] += live ? v[i][0] : 0.f; a[1] += live ? v[i][1] : 0.f; }
  rg_all_sum<2, CP>(red, a);
  const float mu0 = a[0] * inv_N, mu1 = a[1] * inv_N;
  float q[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const bool live = rg + RG * i < N;
    const float d0 = v[i][0] - mu0, d1 = v[i][1] - mu1;
    q[0] += live ? d0 * d0 : 0.f; q[1] += live ? d1 * d1 : 0.f;
  }
  rg_all_sum<2, CP>(red, q);
  const float rs0 = 1.0f / sqrtf(q[0] * inv_N + eps), rs1 = 1.0f / sqrtf(q[1] * inv_N + eps);  // biased variance, like F.instance_norm
  if (rg == 0) { rstd[(size_t)pair * C + ch] = rs0; rstd[(size_t)pair * C + ch + 1] = rs1; }
  const float k0 = rs0 * gamma[ch], k1 = rs1 * gamma[ch + 1], b0 = beta[ch], b1 = beta[ch + 1];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int rl = rg + RG * i;
    if (rl < N) {
      const float z0 = fmaf(v[i][0] - mu0, k0, b0), z1 = fmaf(v[i][1] - mu1, k1, b1);
      const float a0 = (z0 > 0.f) ? z0 : z0 * slope, a1 = (z1 > 0.f) ? z1 : z1 * slope;
      unsigned p0, p1;
      split2h(a0, a1, p0, p1);  // two fp16 planes: the next layer's operand
      const size_t at = kb_index(col0 + rl, ch, ncols);
      *reinterpret_cast<unsigned*>(planes + at) = p0;
      *reinterpret_cast<unsigned*>(planes + plane_stride + at) = p1;
      if (planes_bwd) {  // two bf16 planes: what the backward reads
        split2(a0, a1, p0, p1);
        *reinterpret_cast<unsigned*>(planes_bwd + at) = p0;
        *reinterpret_cast<unsigned*>(planes_bwd + bwd_stride + at) = p1;
      }
    }
  }
}

// the adjoint, same geometry: dz and x^ of the thread's rows stay in registers between the two sums and the dY they feed (the rows come
// in four at a time: with all of a thread's raw words in flight at once the 128 registers of a 1024-thread workgroup spill)
template <int RPT, int CP>
__global__ void __launch_bounds__(1024)
est_in_bwd_r_kernel(const float* __restrict__ dA, int ldd, size_t split_stride, int S, const float* __restrict__ dlogit,
                    const float* __restrict__ w_head, const bf16_t* __restrict__ planes, size_t plane_stride, const float* __restrict__ rstd,
                    const float* __restrict__ gamma, const float* __restrict__ beta, float slope, int C, int N, size_t ncols,
                    bf16_t* __restrict__ dYp, size_t dy_plane, float* __restrict__ dgamma_part, float* __restrict__ dbeta_part) {
  constexpr int RG = 1024 / CP;
  constexpr int CHUNK = RPT < 4 ? RPT : 4;
  __shared__ float red[16][4][CP];
  const int pair = (int)blockIdx.x, cb = (int)blockIdx.y * (2 * CP);
  const int cp = threadIdx.x & (CP - 1), rg = threadIdx.x / CP;
  const int ch = cb + 2 * cp;
  const size_t col0 = (size_t)pair * N;
  const float g0 = gamma[ch], g1 = gamma[ch + 1], b0 = beta[ch], b1 = beta[ch + 1];
  const float ig0 = (fabsf(g0) > 1e-30f) ? 1.0f / g0 : 0.0f, ig1 = (fabsf(g1) > 1e-30f) ? 1.0f / g1 : 0.0f;
  const float islope = 1.0f / slope;
  const float wh0 = dA ? 0.f : w_head[ch], wh1 = dA ? 0.f : w_head[ch + 1];
  // RPT >= 16: x^ is not kept but formed again from a second read of the planes (L2) for the output pass -- dz alone is 32 registers
  constexpr bool kKeepX = RPT < 16;
  f32x2 dz[RPT], xh[kKeepX ? RPT : 1];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i0 = 0; i0 < RPT; i0 += CHUNK) {
    unsigned u0[CHUNK], u1[CHUNK];
    f32x2 d[CHUNK];
    int rgx = rg;
    asm volatile("" : "+v"(rgx));  // the address arithmetic stays with its chunk (hoisted, the thread's 3 x RPT addresses are what spills)
#pragma unroll
    for (int k = 0; k < CHUNK; ++k) {
      const int rl = rgx + RG * (i0 + k);
      const size_t col = col0 + (rl < N ? rl : 0);
      const size_t at = kb_index(col, ch, ncols);
      u0[k] = *reinterpret_cast<const unsigned*>(planes + at);
      u1[k] = *reinterpret_cast<const unsigned*>(planes + plane_stride + at);
      if (dA != nullptr) d[k] = *reinterpret_cast<const f32x2*>(dA + col * ldd + ch);
      else { const float dl = dlogit[col]; d[k] = f32x2{dl * wh0, dl * wh1}; }
    }
    if (dA != nullptr)
      for (int s = 1; s < S; ++s) {
        const float* Ds = dA + (size_t)s * split_stride;
#pragma unroll
        for (int k = 0; k < CHUNK; ++k) {
          const int rl = rgx + RG * (i0 + k);
          d[k] += *reinterpret_cast<const f32x2*>(Ds + (col0 + (rl < N ? rl : 0)) * ldd + ch);
        }
      }
#pragma unroll
    for (int k = 0; k < CHUNK; ++k) {
      const bool live = rg + RG * (i0 + k) < N;
      const float a0 = bf16_lo(u0[k]) + bf16_lo(u1[k]), a1 = bf16_hi(u0[k]) + bf16_hi(u1[k]);
      const float z0 = (a0 > 0.f) ? a0 : a0 * islope, z1 = (a1 > 0.f) ? a1 : a1 * islope;
      const float e0 = live ? ((a0 > 0.f) ? d[k][0] : d[k][0] * slope) : 0.f, e1 = live ? ((a1 > 0.f) ? d[k][1] : d[k][1] * slope) : 0.f;
      dz[i0 + k] = f32x2{e0, e1};
      const float x0 = (z0 - b0) * ig0, x1 = (z1 - b1) * ig1;
      if constexpr (kKeepX) xh[i0 + k] = f32x2{x0, x1};
      s[0] += e0; s[1] += e1;
      s[2] = fmaf(e0, x0, s[2]); s[3] = fmaf(e1, x1, s[3]);
    }
    __builtin_amdgcn_sched_barrier(0);  // one chunk's raw words at a time
  }
  rg_all_sum<4, CP>(red, s);
  if (rg == 0) {
    dbeta_part[(size_t)pair * C + ch] = s[0]; dbeta_part[(size_t)pair * C + ch + 1] = s[1];
    dgamma_part[(size_t)pair * C + ch] = s[2]; dgamma_part[(size_t)pair * C + ch + 1] = s[3];
  }
  const float inv_n = 1.0f / (float)N;
  const float k0 = rstd[(size_t)pair * C + ch] * g0, k1 = rstd[(size_t)pair * C + ch + 1] * g1;
  const float m10 = s[0] * inv_n, m11 = s[1] * inv_n, m20 = s[2] * inv_n, m21 = s[3] * inv_n;
#pragma unroll
  for (int i0 = 0; i0 < RPT; i0 += CHUNK) {
    f32x2 x[CHUNK];
    int rgx = rg;
    asm volatile("" : "+v"(rgx));
    if constexpr (kKeepX) {
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) x[k] = xh[i0 + k];
    } else {
      unsigned u0[CHUNK], u1[CHUNK];
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) {
        const int rl = rgx + RG * (i0 + k);
        const size_t at = kb_index(col0 + (rl < N ? rl : 0), ch, ncols);
        u0[k] = *reinterpret_cast<const unsigned*>(planes + at);
        u1[k] = *reinterpret_cast<const unsigned*>(planes + plane_stride + at);
      }
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) {
        const float a0 = bf16_lo(u0[k]) + bf16_lo(u1[k]), a1 = bf16_hi(u0[k]) + bf16_hi(u1[k]);
        const float z0 = (a0 > 0.f) ? a0 : a0 * islope, z1 = (a1 > 0.f) ? a1 : a1 * islope;
        x[k] = f32x2{(z0 - b0) * ig0, (z1 - b1) * ig1};
      }
    }
#pragma unroll
    for (int k = 0; k < CHUNK; ++k) {
      const int rl = rgx + RG * (i0 + k);
      if (rl < N) {
        unsigned p0, p1;
        split2(k0 * (dz[i0 + k][0] - m10 - x[k][0] * m20), k1 * (dz[i0 + k][1] - m11 - x[k][1] * m21), p0, p1);
        const size_t at = kb_index(col0 + rl, ch, ncols);
        *reinterpret_cast<unsigned*>(dYp + at) = p0;
        *reinterpret_cast<unsigned*>(dYp + dy_plane + at) = p1;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- head: logits[col] = sum_c w[c] a[col][c] + b (Conv1d(256 -> 1)); one 16-lane row per column ----------------------------
__global__ void __launch_bounds__(256)
est_head_fwd_kernel(const bf16_t* __restrict__ planes, size_t plane_stride, int C, int ncols, const float* __restrict__ w,
                    const float* __restrict__ bias, float* __restrict__ logits) {
  const int col = (int)blockIdx.x * 16 + (int)(threadIdx.x >> 4), l = threadIdx.x & 15;
  const int cc = (col < ncols) ? col : ncols - 1;
  float s = 0.f;
  for (int ch = 2 * l; ch < C; ch += 32) {
    const size_t at = kb_index((size_t)cc, ch, (size_t)ncols);
    const unsigned u0 = *reinterpret_cast<const unsigned*>(planes + at);
    const unsigned u1 = *reinterpret_cast<const unsigned*>(planes + plane_stride + at);
    s = fmaf(f16_lo(u0) + f16_lo(u1), w[ch], s);  // the forward's two fp16 planes
    s = fmaf(f16_hi(u0) + f16_hi(u1), w[ch + 1], s);
  }
  s = row16_sum(s);
  if (l == 0 && col < ncols) logits[col] = s + (bias ? bias[0] : 0.f);
}
// d w_head partials: part[block][c] = sum over the block's columns of dlogit[col] a[col][c].  Thread = (channel pair, one of
// eight column groups): the 16 channel pairs of a K block read one contiguous 64-byte row per column.
__global__ void __launch_bounds__(256)
est_head_dw_kernel(const bf16_t* __restrict__ planes, size_t plane_stride, int C, int ncols, int cols_per_block,
                   const float* __restrict__ dlogit, float* __restrict__ part, float* __restrict__ bias_part) {
  __shared__ float red[8][64];
  __shared__ float redb[8];
  float sd = 0.f;  // bias_part[block] = sum of dlogit over the block's columns (the head bias gradient's partials; round 6: was a launch of its own)
  const int c0 = (int)blockIdx.x * cols_per_block;
  int c1 = c0 + cols_per_block;
  c1 = (c1 < ncols) ? c1 : ncols;
  const int cp = threadIdx.x & 31, cg = threadIdx.x >> 5;
  for (int cb = 0; cb < C; cb += 64) {
    const int ch = cb + 2 * cp;
    float s0 = 0.f, s1 = 0.f;
    if (ch < C) {
      // two leading planes (a to 2^-17: a gradient), four columns in flight per thread
      int col = c0 + cg;
      for (; col + 24 < c1; col += 32) {
        unsigned u0[4], u1[4];
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t at = kb_index((size_t)(col + 8 * u), ch, (size_t)ncols);
          u0[u] = *reinterpret_cast<const unsigned*>(planes + at);
          u1[u] = *reinterpret_cast<const unsigned*>(planes + plane_stride + at);
          d[u] = dlogit[col + 8 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s0 = fmaf(bf16_lo(u0[u]) + bf16_lo(u1[u]), d[u], s0);
          s1 = fmaf(bf16_hi(u0[u]) + bf16_hi(u1[u]), d[u], s1);
        }
        if (cb == 0) sd += (d[0] + d[1]) + (d[2] + d[3]);
      }
      for (; col < c1; col += 8) {
        const size_t at = kb_index((size_t)col, ch, (size_t)ncols);
        const unsigned u0 = *reinterpret_cast<const unsigned*>(planes + at);
        const unsigned u1 = *reinterpret_cast<const unsigned*>(planes + plane_stride + at);
        const float d = dlogit[col];
        s0 = fmaf(bf16_lo(u0) + bf16_lo(u1), d, s0);
        s1 = fmaf(bf16_hi(u0) + bf16_hi(u1), d, s1);
        if (cb == 0) sd += d;
      }
    }
    red[cg][2 * cp] = s0; red[cg][2 * cp + 1] = s1;
    if (cb == 0 && cp == 0) redb[cg] = sd;  // (C >= 32: channel pair 0 exists and saw every column of its group)
    __syncthreads();
    if (cb == 0 && bias_part != nullptr && threadIdx.x == 0)
      bias_part[blockIdx.x] = ((redb[0] + redb[1]) + (redb[2] + redb[3])) + ((redb[4] + redb[5]) + (redb[6] + redb[7]));
    if (threadIdx.x < 64 && cb + (int)threadIdx.x < C) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += red[q][threadIdx.x];
      part[(size_t)blockIdx.x * C + cb + threadIdx.x] = t;
    }
    __syncthreads();
  }
}

// fp32 [rows][C] (ld = src_ld, first C_src columns used, the rest zero) -> NP K-blocked planes [C/32][rows][32]
template <int NP>
__global__ void __launch_bounds__(256)
est_split_kernel(const float* __restrict__ src, long rows, int C_src, int src_ld, int C, bf16_t* __restrict__ planes, size_t plane_stride) {
  const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (e >= rows * C) return;
  const long r = e / C;
  const int ch = (int)(e - r * C);
  const float x = (ch < C_src) ? src[r * src_ld + ch] : 0.f, y = (ch + 1 < C_src) ? src[r * src_ld + ch + 1] : 0.f;
  unsigned p0, p1, p2;
  split3(x, y, p0, p1, p2);
  const size_t at = kb_index((size_t)r, ch, (size_t)rows);
  *reinterpret_cast<unsigned*>(planes + at) = p0;
  if (NP > 1) *reinterpret_cast<unsigned*>(planes + plane_stride + at) = p1;
  if (NP > 2) *reinterpret_cast<unsigned*>(planes + 2 * plane_stride + at) = p2;
}

// the same into two fp16 planes, optionally scaled by the layer's power of two (weights: see wscale)
__global__ void __launch_bounds__(256)
est_split_f16_kernel(const float* __restrict__ src, long rows, int C_src, int src_ld, int C, const unsigned* __restrict__ absmax,
                     bf16_t* __restrict__ planes, size_t plane_stride) {
  const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (e >= rows * C) return;
  const float sc = absmax ? wscale(*absmax, false) : 1.0f;
  const long r = e / C;
  const int ch = (int)(e - r * C);
  const float x = (ch < C_src) ? src[r * src_ld + ch] * sc : 0.f, y = (ch + 1 < C_src) ? src[r * src_ld + ch + 1] * sc : 0.f;
  unsigned p0, p1;
  split2h(x, y, p0, p1);
  const size_t at = kb_index((size_t)r, ch, (size_t)rows);
  *reinterpret_cast<unsigned*>(planes + at) = p0;
  *reinterpret_cast<unsigned*>(planes + plane_stride + at) = p1;
}
// *word = max(*word, bits of max |src[i]|)  (non-negative floats order like their bit patterns; NaN / inf sort above: the clamp in
// wscale keeps the scale finite and the NaN travels in the data)
__global__ void __launch_bounds__(256)
est_absmax_kernel(const float* __restrict__ src, long n, unsigned* __restrict__ word) {
  unsigned m = 0u;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const unsigned b = __float_as_uint(src[i]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(word, m);
}

// ---- table-driven launches (round 5): at the reference's batch sizes (4-32 pairs) one estimator call is ~75 launches of which ~60
// are a few microseconds of bookkeeping each -- per layer a maximum, a split, a transposed split, three reductions, three fills.  These
// kernels do a call's worth of each in ONE launch, the per-layer pointers travelling in the kernel arguments.
constexpr int kTabMax = 8;    // hidden layers per table (the reference's estimator has five)
constexpr int kSegMax = 40;   // reduction segments per launch (>= 2 + 4 kTabMax: one backward never needs a second launch)
struct WprepTab {
  int n;
  const float* W[kTabMax];   // [Co][Ci] fp32
  int Co[kTabMax], Ci[kTabMax], K[kTabMax];  // K = Ci rounded up to 32
  bf16_t* pf[kTabMax];       // two fp16 planes of W * s, [K/32][Co][32] each
  bf16_t* pt[kTabMax];       // two bf16 planes of W^T (rows = K, channels = Co: [Co/32][K][32]) for the data gradient, or null
  unsigned* words;           // [n] bit patterns of max |W|
};
// kAbsBlocks workgroups per layer leave their partial maxima in part[layer][block] (no atomics, no zeroing beforehand); the split
// kernel merges them (and its first workgroup of a layer publishes words[layer] for the GEMM epilogues that follow)
constexpr int kAbsBlocks = 64;  // (a wavefront merges them with shuffles: <= 64)
__global__ void __launch_bounds__(256) est_wprep_absmax_kernel(const WprepTab T, unsigned* __restrict__ part) {
  __shared__ unsigned red[4];
  const int layer = (int)blockIdx.y;
  const float* W = T.W[layer];
  const long n = (long)T.Co[layer] * T.Ci[layer];
  unsigned m = 0u;
  constexpr long kStride = (long)kAbsBlocks * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * kStride < n; i += 4 * kStride) {  // four loads in flight
    const unsigned b0 = __float_as_uint(W[i]) & 0x7fffffffu, b1 = __float_as_uint(W[i + kStride]) & 0x7fffffffu,
                   b2 = __float_as_uint(W[i + 2 * kStride]) & 0x7fffffffu, b3 = __float_as_uint(W[i + 3 * kStride]) & 0x7fffffffu;
    const unsigned c0 = b0 > b1 ? b0 : b1, c1 = b2 > b3 ? b2 : b3, c = c0 > c1 ? c0 : c1;
    m = c > m ? c : m;
  }
  for (; i < n; i += kStride) {
    const unsigned b = __float_as_uint(W[i]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w) m = red[w] > m ? red[w] : m;
    part[layer * kAbsBlocks + blockIdx.x] = m;
  }
}
__global__ void __launch_bounds__(256) est_wprep_split_kernel(const WprepTab T, const unsigned* __restrict__ part) {
  const int layer = (int)blockIdx.y;
  const float* W = T.W[layer];
  const int Co = T.Co[layer], Ci = T.Ci[layer], K = T.K[layer];
  unsigned mx = part[layer * kAbsBlocks + (threadIdx.x & (kAbsBlocks - 1))];
#pragma unroll
  for (int o = kAbsBlocks / 2; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)mx, o, 64); mx = t > mx ? t : mx; }
  if (blockIdx.x == 0 && threadIdx.x == 0) T.words[layer] = mx;
  const float sc = wscale(mx, false);
  const long nf = (long)Co * K / 2, nt = T.pt[layer] ? (long)K * Co / 2 : 0;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < nf + nt; idx += (long)gridDim.x * 256) {
    unsigned p0, p1;
    if (idx < nf) {  // the forward's operand: row = output channel, channel = input channel
      const long e = idx * 2;
      const int r = (int)(e / K), ch = (int)(e - (long)r * K);
      const float x = (ch < Ci) ? W[(size_t)r * Ci + ch] * sc : 0.f, y = (ch + 1 < Ci) ? W[(size_t)r * Ci + ch + 1] * sc : 0.f;
      split2h(x, y, p0, p1);
      const size_t at = kb_index((size_t)r, ch, (size_t)Co);
      *reinterpret_cast<unsigned*>(T.pf[layer] + at) = p0;
      *reinterpret_cast<unsigned*>(T.pf[layer] + (size_t)Co * K + at) = p1;
    } else {         // the data gradient's: row = input channel (zero rows up to K), channel = output channel
      const long e = (idx - nf) * 2;
      const int r = (int)(e / Co), ch = (int)(e - (long)r * Co);
      const float x = (r < Ci) ? W[(size_t)ch * Ci + r] : 0.f, y = (r < Ci) ? W[(size_t)(ch + 1) * Ci + r] : 0.f;
      split2(x, y, p0, p1);
      const size_t at = kb_index((size_t)r, ch, (size_t)K);
      *reinterpret_cast<unsigned*>(T.pt[layer] + at) = p0;
      *reinterpret_cast<unsigned*>(T.pt[layer] + (size_t)K * Co + at) = p1;
    }
  }
}

// dst[c] = sum over r < rows of src[r * cols + c] for up to kSegMax independent segments (rows = 0: zeros): every reduction of one
// backward -- per-pair d gamma / d beta partials, split-K partials of the weight gradients, the head's -- and the exact-zero
// gradients of the cancelled convolution biases.  Fixed order of additions.  Two shapes of segment: TALL (rows > 32: the per-pair
// partials, thousands of rows x <= 1024 columns): a workgroup = 64 columns x 16 row groups, thread (column, group g) adds rows g,
// g + 16, ... and the sixteen partial sums are added in order; WIDE (rows <= 32: split-K partials, up to 2^19 columns): a workgroup
// = 1024 columns, a thread adds its column's rows in order.
struct ColSumTab {
  int n;
  const float* src[kSegMax];
  float* dst[kSegMax];
  int rows[kSegMax], cols[kSegMax];
  int ld_src[kSegMax], ld_dst[kSegMax];  // ld_dst > 0: the sums are a [cols / ld_src][ld_src] matrix of which the first ld_dst columns are kept, densely
                                         // (the first layer's weight gradient without its zero-padded input channels; round 6: was a launch of its own)
  int first_block[kSegMax + 1];  // prefix sums of the segments' workgroup counts
};
__device__ __forceinline__ void colsum_store(const ColSumTab& T, int seg, int c, float v) {
  const int ls = T.ld_src[seg], ld = T.ld_dst[seg];
  if (ld <= 0) { T.dst[seg][c] = v; return; }
  const int r = c / ls, k = c - r * ls;
  if (k < ld) T.dst[seg][(size_t)r * ld + k] = v;
}
__host__ __device__ inline int colsum_blocks(int rows, int cols) { return rows > 32 ? (cols + 63) / 64 : (cols + 1023) / 1024; }
__global__ void __launch_bounds__(1024) est_colsum_kernel(const ColSumTab T) {
  __shared__ float red[16][64];
  int seg = 0;
  while (seg + 1 < T.n && (int)blockIdx.x >= T.first_block[seg + 1]) ++seg;
  const int rows = T.rows[seg], cols = T.cols[seg], blk = (int)blockIdx.x - T.first_block[seg];
  const float* src = T.src[seg];
  if (rows <= 32) {  // WIDE (uniform per workgroup)
    const int c = blk * 1024 + (int)threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += src[(size_t)r * cols + c];
    colsum_store(T, seg, c, s);
    return;
  }
  const int c = blk * 64 + (int)(threadIdx.x & 63), g = (int)(threadIdx.x >> 6);
  float s = 0.f;
  if (c < cols) {
    int r = g;
    for (; r + 48 < rows; r += 64) {  // four loads in flight
      const float a0 = src[(size_t)r * cols + c], a1 = src[(size_t)(r + 16) * cols + c], a2 = src[(size_t)(r + 32) * cols + c],
                  a3 = src[(size_t)(r + 48) * cols + c];
      s = (((s + a0) + a1) + a2) + a3;
    }
    for (; r < rows; r += 16) s += src[(size_t)r * cols + c];
  }
  red[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && c < cols) {
    float t = red[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][threadIdx.x];
    colsum_store(T, seg, c, t);
  }
}

}  // namespace

extern "C" int dfepe_est_points(void) { return kPts; }

// at most one workgroup per CU: a wavefront is alone on its SIMD and the GEMM kernels take their AHEAD = 2 build (B fragments two
// column tiles ahead of the MFMAs, two LDS stages); DFEPE_EST_SMALL_GRID=0 / 1 forces the choice (A/B timing)
static bool small_grid(const dim3& grid) {
  static const char* force = getenv("DFEPE_EST_SMALL_GRID");
  if (force) return force[0] == '1';
  return (size_t)grid.x * grid.y * grid.z <= 256;  // one workgroup per CU at most: the two-stage build's 86 KB of LDS admit no second one
}

extern "C" int dfepe_est_absmax(const float* src, long n, unsigned* word, void* stream) {
  if (!src || !word || n < 0) return DFEPE_ERR_INVALID_ARG;
  if (n == 0) return DFEPE_OK;
  long blocks = (n + 2047) / 2048;
  blocks = blocks > 256 ? 256 : blocks;
  hipLaunchKernelGGL(est_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, n, word);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// every layer's weights in two launches: the maxima (one workgroup per layer), then the two fp16 planes of W * s for the forward
// and -- where planes_wt[l] is given -- the two bf16 planes of W^T the data gradient multiplies by
extern "C" size_t dfepe_est_wprep_workspace_bytes(int n_layers) { return (size_t)(n_layers > 0 ? n_layers : 0) * kAbsBlocks * sizeof(unsigned); }

extern "C" int dfepe_est_wprep(int n_layers, const float* const* W, const int* Co, const int* Ci, void* const* planes_f16,
                               void* const* planes_wt, unsigned* words, void* workspace, void* stream) {
  if (n_layers <= 0 || n_layers > kTabMax || !W || !Co || !Ci || !planes_f16 || !words || !workspace) return DFEPE_ERR_INVALID_ARG;
  WprepTab T{};
  T.n = n_layers; T.words = words;
  for (int l = 0; l < n_layers; ++l) {
    if (!W[l] || !planes_f16[l] || Co[l] <= 0 || Ci[l] <= 0) return DFEPE_ERR_INVALID_ARG;
    if (planes_wt && planes_wt[l] && (Co[l] & 31)) return DFEPE_ERR_INVALID_ARG;
    T.W[l] = W[l]; T.Co[l] = Co[l]; T.Ci[l] = Ci[l]; T.K[l] = (Ci[l] + 31) / 32 * 32;
    T.pf[l] = static_cast<bf16_t*>(planes_f16[l]); T.pt[l] = planes_wt ? static_cast<bf16_t*>(planes_wt[l]) : nullptr;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned* part = static_cast<unsigned*>(workspace);
  hipLaunchKernelGGL(est_wprep_absmax_kernel, dim3(kAbsBlocks, n_layers), dim3(256), 0, st, T, part);
  hipLaunchKernelGGL(est_wprep_split_kernel, dim3(256, n_layers), dim3(256), 0, st, T, static_cast<const unsigned*>(part));
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// dst[s][c] = sum_r src[s][r][c] for n_seg <= kSegMax segments in one launch (rows[s] = 0: zeros; src[s] may then be null); ld_src / ld_dst
// (null: plain): segment s is a [cols / ld_src][ld_src] matrix of which the first ld_dst columns are stored densely
static int colsum_launch(int n_seg, const float* const* src, const int* rows, const int* cols, float* const* dst, const int* ld_src,
                         const int* ld_dst, void* stream) {
  if (n_seg <= 0 || n_seg > kSegMax || !src || !rows || !cols || !dst) return DFEPE_ERR_INVALID_ARG;
  ColSumTab T{};
  T.n = n_seg;
  int blocks = 0;
  for (int s = 0; s < n_seg; ++s) {
    if (rows[s] < 0 || cols[s] <= 0 || !dst[s] || (rows[s] > 0 && !src[s])) return DFEPE_ERR_INVALID_ARG;
    T.src[s] = src[s]; T.dst[s] = dst[s]; T.rows[s] = rows[s]; T.cols[s] = cols[s];
    T.ld_src[s] = ld_src ? ld_src[s] : 0; T.ld_dst[s] = ld_dst ? ld_dst[s] : 0;
    if (T.ld_dst[s] > 0 && (T.ld_src[s] < T.ld_dst[s] || cols[s] % T.ld_src[s])) return DFEPE_ERR_INVALID_ARG;
    T.first_block[s] = blocks;
    blocks += colsum_blocks(rows[s], cols[s]);
  }
  T.first_block[n_seg] = blocks;
  hipLaunchKernelGGL(est_colsum_kernel, dim3(blocks), dim3(1024), 0, static_cast<hipStream_t>(stream), T);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
extern "C" int dfepe_est_colsum(int n_seg, const float* const* src, const int* rows, const int* cols, float* const* dst, void* stream) {
  return colsum_launch(n_seg, src, rows, cols, dst, nullptr, nullptr, stream);
}

extern "C" int dfepe_est_split_f16(const float* src, long rows, int C_src, int src_ld, int C, const unsigned* absmax, void* planes,
                                   size_t plane_stride, void* stream) {
  if (!src || !planes || rows < 0 || C <= 0 || (C & 31) || C_src <= 0 || C_src > C) return DFEPE_ERR_INVALID_ARG;
  if (rows == 0) return DFEPE_OK;
  const unsigned blocks = (unsigned)((rows * C / 2 + 255) / 256);
  hipLaunchKernelGGL(est_split_f16_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, rows, C_src, src_ld, C, absmax,
                     static_cast<bf16_t*>(planes), plane_stride);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_est_split(const float* src, long rows, int C_src, int src_ld, int C, int n_planes, void* planes, size_t plane_stride,
                               void* stream) {
  if (!src || !planes || rows < 0 || C <= 0 || (C & 31) || C_src <= 0 || C_src > C || n_planes < 1 || n_planes > 3) return DFEPE_ERR_INVALID_ARG;
  if (rows == 0) return DFEPE_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned blocks = (unsigned)((rows * C / 2 + 255) / 256);
  bf16_t* P = static_cast<bf16_t*>(planes);
  if (n_planes == 3) hipLaunchKernelGGL((est_split_kernel<3>), dim3(blocks), dim3(256), 0, st, src, rows, C_src, src_ld, C, P, plane_stride);
  else if (n_planes == 2) hipLaunchKernelGGL((est_split_kernel<2>), dim3(blocks), dim3(256), 0, st, src, rows, C_src, src_ld, C, P, plane_stride);
  else hipLaunchKernelGGL((est_split_kernel<1>), dim3(blocks), dim3(256), 0, st, src, rows, C_src, src_ld, C, P, plane_stride);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// forward layer: planes_out[2][ncols][M] (fp16) = split(lrelu(IN(W X))) for the next layer, planes_bwd[2][ncols][M] (bf16, or
// null) = the same activation as the backward reads it, rstd[npairs][M].  W, X: two fp16 planes each (three products); W split
// scaled with `absmax` (dfepe_est_absmax + dfepe_est_split_f16), or unscaled with absmax = null.
extern "C" int dfepe_est_layer_fwd(const void* W, size_t w_plane, const void* X, size_t x_plane, int M, int ncols, int K,
                                   const unsigned* absmax, const float* gamma, const float* beta, float eps, float slope,
                                   void* planes_out, size_t out_plane, void* planes_bwd, size_t bwd_plane, float* rstd, void* stream) {
  if (!W || !X || !gamma || !beta || !planes_out || !rstd || M <= 0 || (M & 31) || ncols <= 0 || (ncols % kPts) || K <= 0 || (K % BK))
    return DFEPE_ERR_INVALID_ARG;
  if (!(slope > 0.f)) return DFEPE_ERR_UNSUPPORTED;  // the backward inverts the activation
  EpiArgs E{};
  E.gamma = gamma; E.beta = beta; E.eps = eps; E.slope = slope; E.planes = static_cast<bf16_t*>(planes_out); E.plane_stride = out_plane;
  E.rstd = rstd; E.absmax = absmax; E.planes_bwd = static_cast<bf16_t*>(planes_bwd); E.bwd_stride = bwd_plane;
  const dim3 grid((ncols + BSTEP - 1) / BSTEP, (M + BM - 1) / BM), block(256);
  if (small_grid(grid))
    hipLaunchKernelGGL((est_gemm_nt_kernel<2, 2, 1, EPI_IN, FMT_F16, 2>), grid, block, 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(W), w_plane, static_cast<const bf16_t*>(X), x_plane, M, ncols, K, E);
  else
    hipLaunchKernelGGL((est_gemm_nt_kernel<2, 2, 1, EPI_IN, FMT_F16>), grid, block, 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(W), w_plane, static_cast<const bf16_t*>(X), x_plane, M, ncols, K, E);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// the forward's plain product (any number of points per pair): out[col][m] fp32 = (1 / s) sum_terms A_i[m][:] . B_j[col][:] on two
// fp16 planes each, A split scaled by s (absmax as above, or null).  splits = S > 1 (round 6): S workgroups share a tile's K steps and leave
// S partial products out[z][col][m], z < S, split_stride floats apart -- summed by dfepe_est_norm_fwd_r
static int nt_f16_launch(const void* A, size_t a_plane, const void* B, size_t b_plane, int M, int ncols, int K, const unsigned* absmax,
                         float* out, int ldc, int splits, size_t split_stride, void* stream) {
  if (!A || !B || !out || M <= 0 || (M & 7) || ncols <= 0 || K <= 0 || (K % BK) || ldc < M || (ldc & 3)) return DFEPE_ERR_INVALID_ARG;
  if (splits < 1 || splits > K / BK || (splits > 1 && split_stride < (size_t)ncols * ldc)) return DFEPE_ERR_INVALID_ARG;
  EpiArgs E{};
  E.out = out; E.ldc = ldc; E.absmax = absmax; E.split_stride = split_stride;
  const dim3 grid((ncols + BSTEP - 1) / BSTEP, (M + BM - 1) / BM, splits), block(256);
  if (small_grid(grid))
    hipLaunchKernelGGL((est_gemm_nt_kernel<2, 2, 1, EPI_F32, FMT_F16, 2>), grid, block, 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(A), a_plane, static_cast<const bf16_t*>(B), b_plane, M, ncols, K, E);
  else
    hipLaunchKernelGGL((est_gemm_nt_kernel<2, 2, 1, EPI_F32, FMT_F16>), grid, block, 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(A), a_plane, static_cast<const bf16_t*>(B), b_plane, M, ncols, K, E);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
extern "C" int dfepe_est_gemm_nt_f16(const void* A, size_t a_plane, const void* B, size_t b_plane, int M, int ncols, int K,
                                     const unsigned* absmax, float* out, int ldc, void* stream) {
  return nt_f16_launch(A, a_plane, B, b_plane, M, ncols, K, absmax, out, ldc, 1, 0, stream);
}
extern "C" int dfepe_est_gemm_nt_f16_splitk(const void* A, size_t a_plane, const void* B, size_t b_plane, int M, int ncols, int K,
                                            const unsigned* absmax, float* out, int ldc, int splits, size_t split_stride, void* stream) {
  return nt_f16_launch(A, a_plane, B, b_plane, M, ncols, K, absmax, out, ldc, splits, split_stride, stream);
}

// plain product: out[col][m] (fp32, ld = ldc) = sum_terms A_i[m][:] . B_j[col][:]; n_planes = 2 (three products) or 3 (six); splits as above
// (two planes only); gx != null (two planes, splits = 1): the product is the gradient w.r.t. the estimator's input and is stored as
// gx[pair * gx_sb + ch * gx_sc + n], ch < gx_C0, col = pair * gx_N + n, instead of out (round 6: was a transposing launch of its own)
static int nt_bf16_launch(const void* A, size_t a_plane, const void* B, size_t b_plane, int M, int ncols, int K, int n_planes, float* out,
                          int ldc, int splits, size_t split_stride, float* gx, int gx_C0, int gx_N, long gx_sb, long gx_sc, void* stream) {
  if (!A || !B || (!out && !gx) || M <= 0 || (M & 7) || ncols <= 0 || K <= 0 || (K % BK)) return DFEPE_ERR_INVALID_ARG;
  if (!gx && (ldc < M || (ldc & 3))) return DFEPE_ERR_INVALID_ARG;
  if (gx && (gx_C0 <= 0 || gx_C0 > M || gx_N <= 0 || (ncols % gx_N) || splits != 1 || n_planes != 2 || gx_sb <= 0 || gx_sc <= 0)) return DFEPE_ERR_INVALID_ARG;
  if (n_planes != 2 && n_planes != 3) return DFEPE_ERR_INVALID_ARG;
  if (splits < 1 || splits > K / BK || (splits > 1 && (n_planes != 2 || split_stride < (size_t)ncols * ldc))) return DFEPE_ERR_INVALID_ARG;
  EpiArgs E{};
  E.out = out; E.ldc = ldc; E.split_stride = split_stride; E.gx = gx; E.gx_C0 = gx_C0; E.gx_N = gx_N; E.gx_sb = gx_sb; E.gx_sc = gx_sc;
  const dim3 grid((ncols + BSTEP - 1) / BSTEP, (M + BM - 1) / BM, splits), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_planes == 3)
    hipLaunchKernelGGL((est_gemm_nt_kernel<3, 3, 2, EPI_F32>), grid, block, 0, st, static_cast<const bf16_t*>(A), a_plane,
                       static_cast<const bf16_t*>(B), b_plane, M, ncols, K, E);
  else if (small_grid(grid))
    hipLaunchKernelGGL((est_gemm_nt_kernel<2, 2, 1, EPI_F32, FMT_BF16, 2>), grid, block, 0, st, static_cast<const bf16_t*>(A), a_plane,
                       static_cast<const bf16_t*>(B), b_plane, M, ncols, K, E);
  else
    hipLaunchKernelGGL((est_gemm_nt_kernel<2, 2, 1, EPI_F32>), grid, block, 0, st, static_cast<const bf16_t*>(A), a_plane,
                       static_cast<const bf16_t*>(B), b_plane, M, ncols, K, E);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
extern "C" int dfepe_est_gemm_nt(const void* A, size_t a_plane, const void* B, size_t b_plane, int M, int ncols, int K, int n_planes,
                                 float* out, int ldc, void* stream) {
  return nt_bf16_launch(A, a_plane, B, b_plane, M, ncols, K, n_planes, out, ldc, 1, 0, nullptr, 0, 0, 0, 0, stream);
}
extern "C" int dfepe_est_gemm_nt_splitk(const void* A, size_t a_plane, const void* B, size_t b_plane, int M, int ncols, int K, float* out,
                                        int ldc, int splits, size_t split_stride, void* stream) {
  return nt_bf16_launch(A, a_plane, B, b_plane, M, ncols, K, 2, out, ldc, splits, split_stride, nullptr, 0, 0, 0, 0, stream);
}
extern "C" int dfepe_est_gemm_nt_gx(const void* A, size_t a_plane, const void* B, size_t b_plane, int M, int ncols, int K, float* gx, int C0,
                                    int N, long gx_stride_pair, long gx_stride_ch, void* stream) {
  return nt_bf16_launch(A, a_plane, B, b_plane, M, ncols, K, 2, nullptr, 0, 1, 0, gx, C0, N, gx_stride_pair, gx_stride_ch, stream);
}

// data gradient + the adjoint of the layer below in one launch (N = dfepe_est_points()):
//   dA[col][m] = sum_k WT[m][k] dY_next[col][k]  (two bf16 planes each, three products; never written),
//   dY[2][ncols][M] = InstanceNorm + LeakyReLU adjoint of the layer whose output planes are `aout`, and its per-pair d gamma / d beta
extern "C" int dfepe_est_dgrad_in_bwd(const void* WT, size_t wt_plane, const void* dY_next, size_t dyn_plane, int M, int ncols, int K,
                                      const void* aout, size_t aout_plane, const float* rstd, const float* gamma, const float* beta,
                                      float slope, void* dY, size_t dy_plane, float* dgamma_part, float* dbeta_part, void* stream) {
  if (!WT || !dY_next || !aout || !rstd || !gamma || !beta || !dY || !dgamma_part || !dbeta_part) return DFEPE_ERR_INVALID_ARG;
  if (M <= 0 || (M & 31) || ncols <= 0 || (ncols % kPts) || K <= 0 || (K % BK) || !(slope > 0.f)) return DFEPE_ERR_INVALID_ARG;
  EpiArgs E{};
  E.gamma = gamma; E.beta = beta; E.slope = slope; E.planes = static_cast<bf16_t*>(dY); E.plane_stride = dy_plane;
  E.aout = static_cast<const bf16_t*>(aout); E.aout_stride = aout_plane; E.rstd_in = rstd; E.dgamma_part = dgamma_part; E.dbeta_part = dbeta_part;
  const dim3 grid((ncols + BSTEP - 1) / BSTEP, (M + BM - 1) / BM), block(256);
  if (small_grid(grid))
    hipLaunchKernelGGL((est_gemm_nt_kernel<2, 2, 1, EPI_INBWD, FMT_BF16, 2>), grid, block, 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(WT), wt_plane, static_cast<const bf16_t*>(dY_next), dyn_plane, M, ncols, K, E);
  else
    hipLaunchKernelGGL((est_gemm_nt_kernel<2, 2, 1, EPI_INBWD>), grid, block, 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(WT), wt_plane, static_cast<const bf16_t*>(dY_next), dyn_plane, M, ncols, K, E);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// weight gradient partials: part[slices][Cout][Cin]
extern "C" int dfepe_est_gemm_tn(const void* dY, size_t dy_plane, int Cout, const void* X, size_t x_plane, int Cin, int ncols,
                                 int slices, float* part, void* stream) {
  if (!dY || !X || !part || Cout <= 0 || Cin <= 0 || (Cout & 31) || (Cin & 31) || ncols <= 0 || slices <= 0) return DFEPE_ERR_INVALID_ARG;
  int cps = (ncols + slices - 1) / slices;
  cps = ((cps + BK - 1) / BK) * BK;
  const dim3 grid((Cout + 127) / 128, (Cin + 127) / 128, slices), block(256);
  hipLaunchKernelGGL(est_gemm_tn_kernel, grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(dY), dy_plane, Cout,
                     static_cast<const bf16_t*>(X), x_plane, Cin, ncols, cps, part);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// the weight-gradient partials of n_layers <= 8 layers over the same ncols columns in ONE launch (host arrays indexed by layer)
extern "C" int dfepe_est_gemm_tn_multi(int n_layers, const void* const* dY, const size_t* dy_plane, const int* Cout, const void* const* X,
                                       const size_t* x_plane, const int* Cin, int ncols, const int* slices, float* const* part, void* stream) {
  if (n_layers <= 0 || n_layers > 8 || !dY || !dy_plane || !Cout || !X || !x_plane || !Cin || !slices || !part || ncols <= 0)
    return DFEPE_ERR_INVALID_ARG;
  TnTab T{};
  T.n = n_layers;
  int blocks = 0;
  for (int l = 0; l < n_layers; ++l) {
    if (!dY[l] || !X[l] || !part[l] || Cout[l] <= 0 || Cin[l] <= 0 || (Cout[l] & 31) || (Cin[l] & 31) || slices[l] <= 0) return DFEPE_ERR_INVALID_ARG;
    int cps = (ncols + slices[l] - 1) / slices[l];
    cps = ((cps + BK - 1) / BK) * BK;
    T.dY[l] = static_cast<const bf16_t*>(dY[l]); T.X[l] = static_cast<const bf16_t*>(X[l]); T.dy_plane[l] = dy_plane[l]; T.x_plane[l] = x_plane[l];
    T.Cout[l] = Cout[l]; T.Cin[l] = Cin[l]; T.nx[l] = (Cout[l] + 127) / 128; T.ny[l] = (Cin[l] + 127) / 128; T.slices[l] = slices[l]; T.cps[l] = cps;
    T.part[l] = part[l];
    T.first_block[l] = blocks;
    blocks += (T.nx[l] * T.ny[l] * slices[l] + 7) / 8 * 8;
  }
  T.first_block[n_layers] = blocks;
  hipLaunchKernelGGL(est_gemm_tn_multi_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), T, ncols);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// ---- d gamma of a channel whose gamma is EXACTLY zero (ADVICE r3 / VERDICT r4 7d) ---------------------------------------------------
// The adjoints above recover x^ from the stored activation as (z - beta) / gamma.  With gamma = 0 the activation is the constant
// lrelu(beta) and x^ is gone: they take it as 0, which leaves d beta and dY right (dY = 0 there anyway) and d gamma = sum dz x^ wrong.
// This kernel -- one workgroup per channel, leaving at once unless |gamma| < 1e-30, i.e. a handful of idle workgroups in any trained
// network -- recomputes that channel's pre-normalisation product y = W[ch] . x from the layer's INPUT planes and fp32 weights, forms
// x^ = (y - mean) rstd per pair and overwrites the channel's per-pair d gamma.  Slow (a serial GEMV per pair) and rare by construction.
constexpr int kFixMaxN = 4096;
struct ZeroFix {  // one layer's arguments
  const float* dA; const float* dlogit; const float* w_head;
  const bf16_t* out_planes; size_t out_stride;
  const bf16_t* in_planes; size_t in_stride;
  const float* W; int ldw, Ci;
  const float* rstd; const float* gamma;
  int C;
  float* dgamma_part;
  const bf16_t* dYn; size_t dyn_stride; const float* Wn; int ldwn, Cn;
};
struct ZeroFixTab { int n; float slope; int N; long n_pairs; ZeroFix L[kTabMax]; };
// grid (max C, layers): a launch covers every layer of a backward when their dY are still alive (few columns), one layer otherwise
__global__ void __launch_bounds__(256) est_dgamma_zero_kernel(const ZeroFixTab T) {
  const ZeroFix& A = T.L[blockIdx.y];
  const int ch = (int)blockIdx.x;
  if (ch >= A.C || fabsf(A.gamma[ch]) > 1e-30f) return;
  const float slope = T.slope;
  const int N = T.N, C = A.C, Ci = A.Ci;
  const long n_pairs = T.n_pairs;
  __shared__ float y[kFixMaxN];
  __shared__ float red[4];
  const int t = (int)threadIdx.x;
  const size_t ncols = (size_t)n_pairs * N;
  auto plane2 = [&](const bf16_t* P, size_t stride, size_t at) {  // the backward's two bf16 planes (2^-17: a gradient)
    return __uint_as_float((unsigned)P[at] << 16) + __uint_as_float((unsigned)P[stride + at] << 16);
  };
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  for (long pair = 0; pair < n_pairs; ++pair) {
    const size_t col0 = (size_t)pair * N;
    float s = 0.f;
    for (int r = t; r < N; r += 256) {
      float acc = 0.f;
      for (int k = 0; k < Ci; ++k) acc = fmaf(A.W[(size_t)ch * A.ldw + k], plane2(A.in_planes, A.in_stride, kb_index(col0 + r, k, ncols)), acc);
      y[r] = acc;
      s += acc;
    }
    const float mean = block_sum(s) / (float)N;
    const float rs = A.rstd[(size_t)pair * C + ch];
    float g = 0.f;
    for (int r = t; r < N; r += 256) {
      const size_t col = col0 + r;
      const size_t at = kb_index(col, ch, ncols);
      const float a = __uint_as_float((unsigned)A.out_planes[at] << 16) + __uint_as_float((unsigned)A.out_planes[A.out_stride + at] << 16);
      float d;
      if (A.dA != nullptr) d = A.dA[col * C + ch];
      else if (A.dlogit != nullptr) d = A.dlogit[col] * A.w_head[ch];
      else {  // the fused data gradient never wrote dA: this channel's column of dY_next W_next again (Cn terms per point)
        d = 0.f;
        for (int k = 0; k < A.Cn; ++k) d = fmaf(plane2(A.dYn, A.dyn_stride, kb_index(col, k, ncols)), A.Wn[(size_t)k * A.ldwn + ch], d);
      }
      const float dz = (a > 0.f) ? d : d * slope;
      g = fmaf(dz, (y[r] - mean) * rs, g);
    }
    const float G = block_sum(g);
    if (t == 0) A.dgamma_part[(size_t)pair * C + ch] = G;
  }
}

static int zero_fix_fill(ZeroFix& Z, const float* dA, const float* dlogit, const float* w_head, const void* out_planes, size_t out_plane,
                         const void* in_planes, size_t in_plane, const float* W, int ldw, int Ci, const float* rstd, const float* gamma, int C,
                         float* dgamma_part, const void* dY_next, size_t dyn_plane, const float* W_next, int ldw_next, int C_next) {
  const bool from_next = dY_next && W_next && C_next > 0 && ldw_next >= C;
  if ((!dA && !(dlogit && w_head) && !from_next) || !out_planes || !in_planes || !W || !rstd || !gamma || !dgamma_part) return DFEPE_ERR_INVALID_ARG;
  if (C <= 0 || Ci <= 0 || ldw < Ci) return DFEPE_ERR_INVALID_ARG;
  Z.dA = dA; Z.dlogit = dlogit; Z.w_head = w_head; Z.out_planes = static_cast<const bf16_t*>(out_planes); Z.out_stride = out_plane;
  Z.in_planes = static_cast<const bf16_t*>(in_planes); Z.in_stride = in_plane; Z.W = W; Z.ldw = ldw; Z.Ci = Ci; Z.rstd = rstd; Z.gamma = gamma;
  Z.C = C; Z.dgamma_part = dgamma_part; Z.dYn = static_cast<const bf16_t*>(dY_next); Z.dyn_stride = dyn_plane; Z.Wn = W_next; Z.ldwn = ldw_next;
  Z.Cn = C_next;
  return DFEPE_OK;
}

extern "C" int dfepe_est_dgamma_zero(const float* dA, const float* dlogit, const float* w_head, const void* out_planes, size_t out_plane,
                                     const void* in_planes, size_t in_plane, const float* W, int ldw, int Ci, const float* rstd,
                                     const float* gamma, float slope, int C, int N, long n_pairs, float* dgamma_part,
                                     const void* dY_next, size_t dyn_plane, const float* W_next, int ldw_next, int C_next, void* stream) {
  if (N <= 0 || N > kFixMaxN || n_pairs < 0 || !(slope > 0.f)) return DFEPE_ERR_INVALID_ARG;
  ZeroFixTab T{};
  T.n = 1; T.slope = slope; T.N = N; T.n_pairs = n_pairs;
  const int rc = zero_fix_fill(T.L[0], dA, dlogit, w_head, out_planes, out_plane, in_planes, in_plane, W, ldw, Ci, rstd, gamma, C, dgamma_part,
                               dY_next, dyn_plane, W_next, ldw_next, C_next);
  if (rc != DFEPE_OK) return rc;
  if (n_pairs == 0) return DFEPE_OK;
  hipLaunchKernelGGL(est_dgamma_zero_kernel, dim3(C, 1), dim3(256), 0, static_cast<hipStream_t>(stream), T);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// the same for n_layers <= 8 layers in one launch (arrays indexed by layer; ldw = Ci, ldw_next = C): what a backward over few columns
// does, where keeping every layer's dY alive to the end costs nothing
extern "C" int dfepe_est_dgamma_zero_multi(int n_layers, const float* const* dA, const float* const* dlogit, const float* const* w_head,
                                           const void* const* out_planes, const size_t* out_plane, const void* const* in_planes,
                                           const size_t* in_plane, const float* const* W, const int* Ci, const float* const* rstd,
                                           const float* const* gamma, const int* C, float* const* dgamma_part, const void* const* dY_next,
                                           const size_t* dyn_plane, const float* const* W_next, const int* C_next, float slope, int N,
                                           long n_pairs, void* stream) {
  if (n_layers <= 0 || n_layers > kTabMax || N <= 0 || N > kFixMaxN || n_pairs < 0 || !(slope > 0.f)) return DFEPE_ERR_INVALID_ARG;
  if (!dA || !dlogit || !w_head || !out_planes || !out_plane || !in_planes || !in_plane || !W || !Ci || !rstd || !gamma || !C || !dgamma_part ||
      !dY_next || !dyn_plane || !W_next || !C_next)
    return DFEPE_ERR_INVALID_ARG;
  ZeroFixTab T{};
  T.n = n_layers; T.slope = slope; T.N = N; T.n_pairs = n_pairs;
  int cmax = 0;
  for (int l = 0; l < n_layers; ++l) {
    const int rc = zero_fix_fill(T.L[l], dA[l], dlogit[l], w_head[l], out_planes[l], out_plane[l], in_planes[l], in_plane[l], W[l], Ci[l], Ci[l],
                                 rstd[l], gamma[l], C[l], dgamma_part[l], dY_next[l], dyn_plane[l], W_next[l], C[l], C_next[l]);
    if (rc != DFEPE_OK) return rc;
    cmax = C[l] > cmax ? C[l] : cmax;
  }
  if (n_pairs == 0) return DFEPE_OK;
  hipLaunchKernelGGL(est_dgamma_zero_kernel, dim3(cmax, n_layers), dim3(256), 0, static_cast<hipStream_t>(stream), T);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_est_in_bwd(const float* dA, const float* dlogit, const float* w_head, const void* planes, size_t plane_stride,
                                const float* rstd, const float* gamma, const float* beta, float slope, int C, int ncols, void* dY,
                                size_t dy_plane, float* dgamma_part, float* dbeta_part, void* stream) {
  if ((!dA && !(dlogit && w_head)) || !planes || !rstd || !gamma || !beta || !dY || !dgamma_part || !dbeta_part) return DFEPE_ERR_INVALID_ARG;
  if (C <= 0 || (C & 31) || ncols <= 0 || (ncols % kPts) || !(slope > 0.f)) return DFEPE_ERR_INVALID_ARG;
  const dim3 grid(ncols / kPts, (C + 63) / 64), block(256);
  hipLaunchKernelGGL(est_in_bwd_kernel, grid, block, 0, static_cast<hipStream_t>(stream), dA, dlogit, w_head,
                     static_cast<const bf16_t*>(planes), plane_stride, rstd, gamma, beta, slope, C, ncols, static_cast<bf16_t*>(dY), dy_plane,
                     dgamma_part, dbeta_part);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// InstanceNorm + LeakyReLU + split of a plain product, N points per pair (any N >= 1): Y fp32 [n_pairs * N][ldy].  splits = 1:
// one launch, a workgroup per (pair, 64 channels); splits = 2..64 (few pairs): the pair's rows over `splits` workgroups, two
// launches, part = workspace of n_pairs * splits * 2 * C floats.
extern "C" int dfepe_est_norm_fwd(const float* Y, int ldy, int C, long n_pairs, int N, const float* gamma, const float* beta, float eps,
                                  float slope, void* planes_out, size_t out_plane, void* planes_bwd, size_t bwd_plane, float* rstd,
                                  int splits, float* part, void* stream) {
  if (!Y || !gamma || !beta || !planes_out || !rstd || C <= 0 || (C & 31) || ldy < C || (ldy & 1) || n_pairs < 0 || N <= 0)
    return DFEPE_ERR_INVALID_ARG;
  if (splits < 1 || splits > 64 || (splits > 1 && !part)) return DFEPE_ERR_INVALID_ARG;
  if (!(slope > 0.f)) return DFEPE_ERR_UNSUPPORTED;  // the backward inverts the activation
  if (n_pairs == 0) return DFEPE_OK;
  if (n_pairs > 0x7fffffffL) return DFEPE_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)n_pairs, (C + 63) / 64, splits), block(kRG * 32);
  const size_t ncols = (size_t)n_pairs * N;
  bf16_t* P = static_cast<bf16_t*>(planes_out);
  bf16_t* Q = static_cast<bf16_t*>(planes_bwd);
  if (splits == 1) {
    hipLaunchKernelGGL(est_norm_fwd_n_kernel<0>, grid, block, 0, st, Y, ldy, C, N, ncols, gamma, beta, eps, slope, P, out_plane, Q, bwd_plane, rstd, part);
  } else {
    hipLaunchKernelGGL(est_norm_fwd_n_kernel<1>, grid, block, 0, st, Y, ldy, C, N, ncols, gamma, beta, eps, slope, P, out_plane, Q, bwd_plane, rstd, part);
    hipLaunchKernelGGL(est_norm_fwd_n_kernel<2>, grid, block, 0, st, Y, ldy, C, N, ncols, gamma, beta, eps, slope, P, out_plane, Q, bwd_plane, rstd, part);
  }
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// dfepe_est_in_bwd for N points per pair (any N >= 1; ncols = n_pairs * N); splits / part as in dfepe_est_norm_fwd
extern "C" int dfepe_est_in_bwd_n(const float* dA, const float* dlogit, const float* w_head, const void* planes, size_t plane_stride,
                                  const float* rstd, const float* gamma, const float* beta, float slope, int C, long n_pairs, int N, void* dY,
                                  size_t dy_plane, float* dgamma_part, float* dbeta_part, int splits, float* part, void* stream) {
  if ((!dA && !(dlogit && w_head)) || !planes || !rstd || !gamma || !beta || !dY || !dgamma_part || !dbeta_part) return DFEPE_ERR_INVALID_ARG;
  if (C <= 0 || (C & 31) || n_pairs < 0 || N <= 0 || !(slope > 0.f)) return DFEPE_ERR_INVALID_ARG;
  if (splits < 1 || splits > 64 || (splits > 1 && !part)) return DFEPE_ERR_INVALID_ARG;
  if (n_pairs == 0) return DFEPE_OK;
  if (n_pairs > 0x7fffffffL) return DFEPE_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)n_pairs, (C + 63) / 64, splits), block(kRG * 32);
  const size_t ncols = (size_t)n_pairs * N;
  const bf16_t* P = static_cast<const bf16_t*>(planes);
  bf16_t* D = static_cast<bf16_t*>(dY);
  if (splits == 1) {
    hipLaunchKernelGGL(est_in_bwd_n_kernel<0>, grid, block, 0, st, dA, dlogit, w_head, P, plane_stride, rstd, gamma, beta, slope, C, N, ncols, D,
                       dy_plane, dgamma_part, dbeta_part, part);
  } else {
    hipLaunchKernelGGL(est_in_bwd_n_kernel<1>, grid, block, 0, st, dA, dlogit, w_head, P, plane_stride, rstd, gamma, beta, slope, C, N, ncols, D,
                       dy_plane, dgamma_part, dbeta_part, part);
    hipLaunchKernelGGL(est_in_bwd_n_kernel<2>, grid, block, 0, st, dA, dlogit, w_head, P, plane_stride, rstd, gamma, beta, slope, C, N, ncols, D,
                       dy_plane, dgamma_part, dbeta_part, part);
  }
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// round 6: InstanceNorm + LeakyReLU + split of a plain product held in registers (N <= 2048), the product given as `splits` partials
// Y[z][n_pairs * N][ldy] (split_stride floats apart; splits = 1: the plain product).  One launch, one read of Y.
extern "C" int dfepe_est_norm_fwd_r(const float* Y, int ldy, int splits, size_t split_stride, int C, long n_pairs, int N, const float* gamma,
                                    const float* beta, float eps, float slope, void* planes_out, size_t out_plane, void* planes_bwd,
                                    size_t bwd_plane, float* rstd, void* stream) {
  if (!Y || !gamma || !beta || !planes_out || !rstd || C <= 0 || (C & 31) || ldy < C || (ldy & 1) || n_pairs < 0 || N <= 0 || N > 2048)
    return DFEPE_ERR_INVALID_ARG;
  if (splits < 1 || splits > 64) return DFEPE_ERR_INVALID_ARG;
  if (!(slope > 0.f)) return DFEPE_ERR_UNSUPPORTED;
  if (n_pairs == 0) return DFEPE_OK;
  if (n_pairs > 0x7fffffffL) return DFEPE_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)n_pairs, N <= 1024 ? C / 32 : C / 16), block(1024);
  const size_t ncols = (size_t)n_pairs * N;
  bf16_t* P = static_cast<bf16_t*>(planes_out);
  bf16_t* Q = static_cast<bf16_t*>(planes_bwd);
#define DFEPE_NORM_R(R, CP) hipLaunchKernelGGL((est_norm_fwd_r_kernel<R, CP>), grid, block, 0, st, Y, ldy, split_stride, splits, C, N, ncols, gamma, beta, eps, slope, P, out_plane, Q, bwd_plane, rstd)
  if (N <= 128) DFEPE_NORM_R(2, 16);
  else if (N <= 512) DFEPE_NORM_R(8, 16);
  else if (N <= 1024) DFEPE_NORM_R(16, 16);
  else DFEPE_NORM_R(16, 8);
#undef DFEPE_NORM_R
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
// ... and its adjoint: dA given as `splits` partials [z][n_pairs * N][ldd] (or the head's rank-one form: dA null)
extern "C" int dfepe_est_in_bwd_r(const float* dA, int ldd, int splits, size_t split_stride, const float* dlogit, const float* w_head,
                                  const void* planes, size_t plane_stride, const float* rstd, const float* gamma, const float* beta,
                                  float slope, int C, long n_pairs, int N, void* dY, size_t dy_plane, float* dgamma_part, float* dbeta_part,
                                  void* stream) {
  if ((!dA && !(dlogit && w_head)) || !planes || !rstd || !gamma || !beta || !dY || !dgamma_part || !dbeta_part) return DFEPE_ERR_INVALID_ARG;
  if (C <= 0 || (C & 31) || n_pairs < 0 || N <= 0 || N > 2048 || !(slope > 0.f)) return DFEPE_ERR_INVALID_ARG;
  if (dA && (ldd < C || (ldd & 1) || splits < 1 || splits > 64)) return DFEPE_ERR_INVALID_ARG;
  if (n_pairs == 0) return DFEPE_OK;
  if (n_pairs > 0x7fffffffL) return DFEPE_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)n_pairs, N <= 1024 ? C / 32 : C / 16), block(1024);
  const size_t ncols = (size_t)n_pairs * N;
  const bf16_t* P = static_cast<const bf16_t*>(planes);
  bf16_t* D = static_cast<bf16_t*>(dY);
#define DFEPE_INBWD_R(R, CP) hipLaunchKernelGGL((est_in_bwd_r_kernel<R, CP>), grid, block, 0, st, dA, ldd, split_stride, splits, dlogit, w_head, P, plane_stride, rstd, gamma, beta, slope, C, N, ncols, D, dy_plane, dgamma_part, dbeta_part)
  if (N <= 128) DFEPE_INBWD_R(2, 16);
  else if (N <= 512) DFEPE_INBWD_R(8, 16);
  else if (N <= 1024) DFEPE_INBWD_R(16, 16);
  else DFEPE_INBWD_R(16, 8);
#undef DFEPE_INBWD_R
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_est_head_fwd(const void* planes, size_t plane_stride, int C, int ncols, const float* w, const float* bias,
                                  float* logits, void* stream) {
  if (!planes || !w || !logits || C <= 0 || (C & 31) || ncols <= 0) return DFEPE_ERR_INVALID_ARG;
  hipLaunchKernelGGL(est_head_fwd_kernel, dim3((ncols + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const bf16_t*>(planes), plane_stride, C, ncols, w, bias, logits);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_est_head_dw(const void* planes, size_t plane_stride, int C, int ncols, int blocks, const float* dlogit, float* part,
                                 float* bias_part, void* stream) {
  if (!planes || !dlogit || !part || C <= 0 || (C & 31) || ncols <= 0 || blocks <= 0) return DFEPE_ERR_INVALID_ARG;
  const int cpb = (ncols + blocks - 1) / blocks;
  hipLaunchKernelGGL(est_head_dw_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(planes),
                     plane_stride, C, ncols, cpb, dlogit, part, bias_part);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

// =====================================================================================================================================
// One estimator pass per call (round 5): dfepe_est_forward / dfepe_est_backward run the whole Conv1d -> InstanceNorm -> LeakyReLU stack and
// its head (one output channel) through the entry points above.  Why: at the reference's batch sizes the HOST is the limiter of the eager
// training step (Python spent ~0.27 ms per forward and more per backward on ~30 launches, ~40 allocations and their bookkeeping: more than
// the GPU needs for the whole model's step).  The caller brings three buffers -- `saved` (what the backward reads: the layers' bf16
// planes, reciprocal deviations, transposed weight planes), a transient workspace per pass, the outputs -- whose sizes the *_bytes
// functions give; nothing is allocated here.
//
// Round 6 -- the step a train_good.py user runs (N = 1000-2000 points, 4-12 pairs: deepFEPE/configs/kitti_corr_baseline.yaml:12-13):
//   * every layer has a PLAN (fwd_plan / dgrad_plan): the fused epilogue (N = 100, grid large enough), or the plain product -- split over
//     S slices of K where a batch of a few pairs leaves a K-heavy layer on a dozen workgroups -- followed by ONE register-resident
//     normalisation launch that adds the slices (est_norm_fwd_r / est_in_bwd_r: N <= 2048; beyond, the strided kernels of round 4);
//   * `prep` (dfepe_est_prepare): the weights' planes made ONCE per model forward for an estimator that is called several times in it
//     (DeepFNet.update_weights: depth - 1 calls, deepFEPE/models/DeepFNet.py:510) instead of once per call;
//   * over few columns (all dY of a backward alive together: the gamma == 0 fix already wanted that) the five weight-gradient GEMMs are
//     ONE table-driven launch at the end (est_gemm_tn_multi);
//   * the head bias partials ride in est_head_dw, the first layer's cropped weight gradient in est_colsum, the input gradient's
//     transposition in its GEMM's epilogue: three launches per call gone.
namespace {

constexpr size_t kAlign = 256;
inline size_t up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }
inline int pad32(int c) { return (c + 31) / 32 * 32; }

struct PassDims {
  int n;              // hidden layers
  int Co[kTabMax], Ci[kTabMax], K[kTabMax];
  long B; int N; long cols;
  int C0, K0, Cmax, Kmax;
  bool fusable;       // N == kPts: the fused epilogues exist
  bool resident;      // N <= 2048: the register-resident normalisation kernels serve the plain products
};
int pass_dims(PassDims& D, int n_hidden, const int* Co, const int* Ci, long B, int C0, int N) {
  if (n_hidden <= 0 || n_hidden > kTabMax || !Co || !Ci || B <= 0 || N < 2 || C0 <= 0 || B * (long)N >= (1L << 31)) return DFEPE_ERR_INVALID_ARG;
  D.n = n_hidden; D.B = B; D.N = N; D.cols = B * (long)N; D.C0 = C0; D.K0 = pad32(C0); D.Cmax = 0; D.Kmax = 0;
  D.fusable = (N == kPts); D.resident = (N <= 2048);
  for (int l = 0; l < n_hidden; ++l) {
    if (Co[l] <= 0 || (Co[l] & 31) || Ci[l] != (l ? Co[l - 1] : C0)) return DFEPE_ERR_INVALID_ARG;
    D.Co[l] = Co[l]; D.Ci[l] = Ci[l]; D.K[l] = pad32(Ci[l]);
    D.Cmax = Co[l] > D.Cmax ? Co[l] : D.Cmax; D.Kmax = D.K[l] > D.Kmax ? D.K[l] : D.Kmax;
  }
  return DFEPE_OK;
}
int row_splits(long pairs, int c, int n) {  // estimator.py: _row_splits
  const long blocks = pairs * ((c + 63) / 64);
  if (blocks >= 512 || n < 256) return 1;
  long s = (1024 + blocks - 1) / blocks;
  s = s < n / 128 ? s : n / 128;
  s = s < 64 ? s : 64;
  return (int)(s > 1 ? s : 1);
}
int slices_for(int cout, int cin, long cols) {  // estimator.py: _slices_for (TN_BLOCKS = 768)
  const int tiles = ((cout + 127) / 128) * ((cin + 127) / 128);
  long s = 768 / tiles;
  s = s < 512 ? s : 512;
  s = s < cols / 256 ? s : cols / 256;
  s = s > 1 ? s : 1;
  return (int)(s >= 8 ? s - s % 8 : s);
}

// How one GEMM of a pass runs: through its fused epilogue, or as S partial products + a normalisation launch.
// DFEPE_EST_SPLITK = 0: fused wherever it exists and never split (round 5's behaviour); 2: the plain product wherever it can serve (A/B timing)
struct Plan { bool fused; int S; };
int splitk_mode() {
  static const char* e = getenv("DFEPE_EST_SPLITK");
  return e ? (e[0] - '0') : 1;
}
Plan gemm_plan(const PassDims& D, int M, int K, bool fusable) {
  const long tiles = ((D.cols + BSTEP - 1) / BSTEP) * ((M + BM - 1) / BM);
  const int nk = K / BK, mode = splitk_mode();
  Plan P{fusable, 1};
  if (!D.resident || mode == 0) return P;  // the strided kernels of round 4 take one plain product
  // a K-heavy product on a few workgroups is a chain of K steps with most of the chip idle: slices of >= 4 K steps each, up to ~256
  // workgroups, at most eight partials for the normalisation launch to add
  long S = 1;
  if (tiles < 128 && nk >= 8) {
    S = 256 / tiles;
    S = S < nk / 4 ? S : nk / 4;
    S = S < 8 ? S : 8;
    S = S > 1 ? S : 1;
  }
  if (fusable && mode != 2 && !(tiles <= 64 && nk >= 8)) return P;  // the fused epilogue: one launch, nothing written twice
  P.fused = false; P.S = (int)S;
  return P;
}
Plan fwd_plan(const PassDims& D, int l) { return gemm_plan(D, D.Co[l], D.K[l], D.fusable); }
// the data gradient of layer l (>= 1) = the upstream gradient of layer l - 1: M = K[l] rows (= Co[l - 1]), contraction over Co[l]
Plan dgrad_plan(const PassDims& D, int l) { return gemm_plan(D, D.K[l], D.Co[l], D.fusable && D.Ci[l] == D.K[l]); }

// the weights' planes of one estimator, made once and shared by its calls: words | fp16 planes per layer | transposed bf16 planes per layer
struct PrepLayout { size_t words, absws, wf[kTabMax], wt[kTabMax], total; };
PrepLayout prep_layout(int n, const int* Co, const int* Ci) {
  PrepLayout P{};
  size_t at = 0;
  P.words = at; at += up(sizeof(unsigned) * kTabMax);
  P.absws = at; at += up(dfepe_est_wprep_workspace_bytes(n));
  for (int l = 0; l < n; ++l) { P.wf[l] = at; at += up((size_t)2 * Co[l] * pad32(Ci[l]) * 2); }
  for (int l = 0; l < n; ++l) { P.wt[l] = at; at += up((size_t)2 * pad32(Ci[l]) * Co[l] * 2); }
  P.total = at;
  return P;
}

// what the backward reads, laid out in `saved`
struct SavedLayout {
  size_t words, act[kTabMax + 1], rstd[kTabMax], wt[kTabMax], total;
};
SavedLayout saved_layout(const PassDims& D, bool need_gx) {
  SavedLayout S{};
  size_t at = 0;
  S.words = at; at += up(sizeof(unsigned) * kTabMax);
  S.act[0] = at; at += up((size_t)2 * D.cols * D.K0 * 2);
  for (int l = 0; l < D.n; ++l) { S.act[l + 1] = at; at += up((size_t)2 * D.cols * D.Co[l] * 2); }
  for (int l = 0; l < D.n; ++l) { S.rstd[l] = at; at += up((size_t)D.B * D.Co[l] * 4); }
  (void)need_gx;  // the first layer's transposed planes (a few KB) are always there: the layout must not depend on what the caller
  for (int l = 0; l < D.n; ++l) { S.wt[l] = at; at += up((size_t)2 * D.K[l] * D.Co[l] * 2); }  // later asks the backward for
  S.total = at;
  return S;
}
struct FwdLayout {
  size_t absws, wf[kTabMax], xh, ping, pong, rstd_scratch, Y, npart, words, total;
};
FwdLayout fwd_layout(const PassDims& D, bool keep) {
  FwdLayout F{};
  size_t at = 0;
  F.absws = at; at += up(dfepe_est_wprep_workspace_bytes(D.n));
  F.words = at; at += up(sizeof(unsigned) * kTabMax);  // used when nothing is kept (no `saved`)
  for (int l = 0; l < D.n; ++l) { F.wf[l] = at; at += up((size_t)2 * D.Co[l] * D.K[l] * 2); }
  F.xh = at; at += up((size_t)2 * D.cols * D.K0 * 2);
  int c_even = 0, c_odd = 0;  // the layers' fp16 outputs take turns in two buffers: even layers in one, odd ones in the other
  for (int l = 0; l < D.n; ++l) { int& c = (l & 1) ? c_odd : c_even; c = D.Co[l] > c ? D.Co[l] : c; }
  F.ping = at; at += up((size_t)2 * D.cols * c_even * 2);
  F.pong = at; at += up((size_t)2 * D.cols * c_odd * 2);
  F.rstd_scratch = at; if (!keep) at += up((size_t)D.B * D.Cmax * 4);
  size_t ybytes = 0;  // the largest plain product of the pass, its split-K partials included
  for (int l = 0; l < D.n; ++l) {
    const Plan P = fwd_plan(D, l);
    if (!P.fused) { const size_t b = (size_t)P.S * D.cols * D.Co[l] * 4; ybytes = b > ybytes ? b : ybytes; }
  }
  F.Y = at; at += up(ybytes);
  F.npart = at; if (!D.resident) at += up((size_t)D.B * 64 * 2 * D.Cmax * 4);
  F.total = at;
  return F;
}
struct BwdLayout {
  size_t hpart, bpart, dY[kTabMax], dg[kTabMax], db[kTabMax], partw[kTabMax], dA, npart, total;
  int slices[kTabMax];
  bool keep_all;  // every layer's dY stays alive to the end of the backward: one gamma == 0 launch and one weight-gradient launch there
};
BwdLayout bwd_layout(const PassDims& D, bool need_gx) {
  BwdLayout L{};
  size_t at = 0;
  L.hpart = at; at += up((size_t)512 * D.Co[D.n - 1] * 4);
  L.bpart = at; at += up((size_t)512 * 4);
  size_t dy_total = 0;
  for (int l = 0; l < D.n; ++l) dy_total += (size_t)D.cols * D.Co[l] * 4;
  L.keep_all = dy_total <= ((size_t)64 << 20);
  if (const char* e = getenv("DFEPE_EST_KEEP_ALL")) L.keep_all = e[0] == '1';  // tests: both branches at any size
  if (L.keep_all) {
    for (int l = 0; l < D.n; ++l) { L.dY[l] = at; at += up((size_t)2 * D.cols * D.Co[l] * 2); }
  } else {  // two buffers taking turns: dY of layer l is read while dY of layer l - 1 is written
    int ca = 0, cb = 0;
    for (int l = D.n - 1, k = 0; l >= 0; --l, ++k) { int& c = (k & 1) ? cb : ca; c = D.Co[l] > c ? D.Co[l] : c; }
    const size_t a = at, b = at + up((size_t)2 * D.cols * ca * 2);
    at = b + up((size_t)2 * D.cols * cb * 2);
    for (int l = D.n - 1, k = 0; l >= 0; --l, ++k) L.dY[l] = (k & 1) ? b : a;
  }
  for (int l = 0; l < D.n; ++l) {
    L.dg[l] = at; at += up((size_t)D.B * D.Co[l] * 4);
    L.db[l] = at; at += up((size_t)D.B * D.Co[l] * 4);
    L.slices[l] = slices_for(D.Co[l], D.K[l], D.cols);
    L.partw[l] = at; at += up((size_t)L.slices[l] * D.Co[l] * D.K[l] * 4);
  }
  size_t dabytes = 0;  // the largest data gradient that is written (not fused, not the input's: that one goes straight to gx)
  for (int l = 1; l < D.n; ++l) {
    const Plan P = dgrad_plan(D, l);
    if (!P.fused) { const size_t b = (size_t)P.S * D.cols * D.K[l] * 4; dabytes = b > dabytes ? b : dabytes; }
  }
  (void)need_gx;
  L.dA = at; at += up(dabytes);
  L.npart = at; if (!D.resident) at += up((size_t)D.B * 64 * 2 * D.Cmax * 4);
  L.total = at;
  return L;
}

// x [B][C0][N] fp32 (element (b, c, n) at b * x_sb + c * x_sc + n: dense, or a view of the channel-major [C0][B][N] buffer) -> planes [cols = B N][K0]: two fp16 (what the first layer multiplies by) and, if wanted, two bf16 (what its
// weight gradient multiplies by); channels C0..K0 zero.  A thread = one column x one block of 32 channels: its reads of x run along n
// with its neighbours' (coalesced), its 64 bytes per plane are contiguous with theirs.
__global__ void __launch_bounds__(256)
est_input_split_kernel(const float* __restrict__ x, long x_sb, long x_sc, long B, int C0, int N, int K0, bf16_t* __restrict__ ph, size_t h_stride,
                       bf16_t* __restrict__ pb, size_t b_stride) {
  const long cols = B * (long)N;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;  // (channel block, column), column fastest
  const long col = t % cols;
  const int cb = (int)(t / cols) * 32;
  if (cb >= K0) return;
  const long b = col / N, n = col - b * N;
  unsigned h0[16], h1[16], q0[16], q1[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ch = cb + 2 * i;
    const float v0 = (ch < C0) ? x[(size_t)b * x_sb + (size_t)ch * x_sc + n] : 0.f, v1 = (ch + 1 < C0) ? x[(size_t)b * x_sb + (size_t)(ch + 1) * x_sc + n] : 0.f;
    split2h(v0, v1, h0[i], h1[i]);
    split2(v0, v1, q0[i], q1[i]);
  }
  const size_t at = kb_index((size_t)col, cb, (size_t)cols);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<uint4*>(ph + at + 8 * i) = make_uint4(h0[4 * i], h0[4 * i + 1], h0[4 * i + 2], h0[4 * i + 3]);
    *reinterpret_cast<uint4*>(ph + h_stride + at + 8 * i) = make_uint4(h1[4 * i], h1[4 * i + 1], h1[4 * i + 2], h1[4 * i + 3]);
  }
  if (pb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(pb + at + 8 * i) = make_uint4(q0[4 * i], q0[4 * i + 1], q0[4 * i + 2], q0[4 * i + 3]);
      *reinterpret_cast<uint4*>(pb + b_stride + at + 8 * i) = make_uint4(q1[4 * i], q1[4 * i + 1], q1[4 * i + 2], q1[4 * i + 3]);
    }
  }
}

}  // namespace

#define EST_TRY(call) do { const int rc_ = (call); if (rc_ != DFEPE_OK) return rc_; } while (0)

extern "C" size_t dfepe_est_saved_bytes(int n_hidden, const int* Co, const int* Ci, long B, int C0, int N, int need_gx) {
  PassDims D;
  if (pass_dims(D, n_hidden, Co, Ci, B, C0, N) != DFEPE_OK) return 0;
  return saved_layout(D, need_gx != 0).total;
}
extern "C" size_t dfepe_est_forward_workspace_bytes(int n_hidden, const int* Co, const int* Ci, long B, int C0, int N, int keep) {
  PassDims D;
  if (pass_dims(D, n_hidden, Co, Ci, B, C0, N) != DFEPE_OK) return 0;
  return fwd_layout(D, keep != 0).total;
}
extern "C" size_t dfepe_est_backward_workspace_bytes(int n_hidden, const int* Co, const int* Ci, long B, int C0, int N, int need_gx) {
  PassDims D;
  if (pass_dims(D, n_hidden, Co, Ci, B, C0, N) != DFEPE_OK) return 0;
  return bwd_layout(D, need_gx != 0).total;
}
extern "C" size_t dfepe_est_prep_bytes(int n_hidden, const int* Co, const int* Ci) {
  if (n_hidden <= 0 || n_hidden > kTabMax || !Co || !Ci) return 0;
  for (int l = 0; l < n_hidden; ++l)
    if (Co[l] <= 0 || (Co[l] & 31) || Ci[l] <= 0) return 0;
  return prep_layout(n_hidden, Co, Ci).total;
}
// the weights' planes (scales, scaled fp16 planes for the forward, transposed bf16 planes for the data gradients) of one estimator into
// `prep` (dfepe_est_prep_bytes, 16-byte aligned): two launches, once per model forward; dfepe_est_forward / _backward given `prep`
// read them instead of making their own.  The weights must not change between this call and the last backward that is handed `prep`.
extern "C" int dfepe_est_prepare(int n_hidden, const float* const* W, const int* Co, const int* Ci, void* prep, void* stream) {
  if (n_hidden <= 0 || n_hidden > kTabMax || !W || !Co || !Ci || !prep || ((uintptr_t)prep & 15)) return DFEPE_ERR_INVALID_ARG;
  for (int l = 0; l < n_hidden; ++l)
    if (Co[l] <= 0 || (Co[l] & 31) || Ci[l] <= 0) return DFEPE_ERR_INVALID_ARG;
  const PrepLayout P = prep_layout(n_hidden, Co, Ci);
  char* pp = static_cast<char*>(prep);
  void* pf[kTabMax]; void* pt[kTabMax];
  for (int l = 0; l < n_hidden; ++l) { pf[l] = pp + P.wf[l]; pt[l] = pp + P.wt[l]; }
  return dfepe_est_wprep(n_hidden, W, Co, Ci, pf, pt, reinterpret_cast<unsigned*>(pp + P.words), pp + P.absws, stream);
}

// logits [cols] = head(stack(x)); saved != null: everything dfepe_est_backward needs is left there (need_gx is accepted for symmetry with
// the size functions and ignored: the first layer's transposed weight planes are a few KB and always kept, so that a backward may ask
// for gx or not).  x [B][C0][N] with element (b, c, n) at b * x_stride_b + c * x_stride_c + n (dense: C0 N, N; the model's channel-major input
// buffers [C0][B][N]: N, B N -- no copy either way), W[l] [Co][Ci], gamma / beta [l] [Co], w_head [Co of the last layer], b_head [1] or null: fp32.
// prep: null, or the planes dfepe_est_prepare made of these very weights.
extern "C" int dfepe_est_forward(const float* x, long x_stride_b, long x_stride_c, long B, int C0, int N, int n_hidden, const float* const* W, const float* const* gamma,
                                 const float* const* beta, const int* Co, const int* Ci, const float* w_head, const float* b_head, float eps,
                                 float slope, void* saved, int need_gx, void* workspace, const void* prep, float* logits, void* stream) {
  PassDims D;
  EST_TRY(pass_dims(D, n_hidden, Co, Ci, B, C0, N));
  if (!x || !W || !gamma || !beta || !w_head || !workspace || !logits || x_stride_b <= 0 || x_stride_c <= 0) return DFEPE_ERR_INVALID_ARG;
  if (((uintptr_t)workspace & 15) || ((uintptr_t)saved & 15) || ((uintptr_t)prep & 15)) return DFEPE_ERR_INVALID_ARG;
  const bool keep = saved != nullptr;
  const SavedLayout S = saved_layout(D, need_gx != 0);
  const FwdLayout F = fwd_layout(D, keep);
  const PrepLayout P = prep_layout(D.n, Co, Ci);
  char* ws = static_cast<char*>(workspace);
  char* sv = static_cast<char*>(saved);
  const char* pp = static_cast<const char*>(prep);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long cols = D.cols;
  const unsigned* words = pp ? reinterpret_cast<const unsigned*>(pp + P.words) : reinterpret_cast<unsigned*>(keep ? sv + S.words : ws + F.words);
  // the input's planes
  {
    const long threads = cols * (D.K0 / 32);
    hipLaunchKernelGGL(est_input_split_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, x, x_stride_b, x_stride_c, B, C0, N, D.K0,
                       reinterpret_cast<bf16_t*>(ws + F.xh), (size_t)cols * D.K0, keep ? reinterpret_cast<bf16_t*>(sv + S.act[0]) : nullptr,
                       (size_t)cols * D.K0);
    if (hipGetLastError() != hipSuccess) return DFEPE_ERR_HIP;
  }
  // every layer's weights: scales, fp16 planes, transposed bf16 planes for the backward -- unless the caller prepared them
  if (!pp) {
    void* pf[kTabMax]; void* pt[kTabMax];
    for (int l = 0; l < D.n; ++l) { pf[l] = ws + F.wf[l]; pt[l] = keep ? sv + S.wt[l] : nullptr; }
    EST_TRY(dfepe_est_wprep(D.n, W, Co, Ci, pf, pt, const_cast<unsigned*>(words), ws + F.absws, stream));
  }
  const char* act = ws + F.xh;
  for (int l = 0; l < D.n; ++l) {
    const int C = D.Co[l], K = D.K[l];
    const void* wf = pp ? static_cast<const void*>(pp + P.wf[l]) : static_cast<const void*>(ws + F.wf[l]);
    char* out = ws + ((l & 1) ? F.pong : F.ping);
    void* out_b = keep ? sv + S.act[l + 1] : nullptr;
    float* rstd = reinterpret_cast<float*>(keep ? sv + S.rstd[l] : ws + F.rstd_scratch);
    const Plan pl = fwd_plan(D, l);
    if (pl.fused) {
      EST_TRY(dfepe_est_layer_fwd(wf, (size_t)C * K, act, (size_t)cols * K, C, (int)cols, K, words + l, gamma[l], beta[l], eps, slope, out,
                                  (size_t)cols * C, out_b, (size_t)cols * C, rstd, stream));
    } else {
      float* Y = reinterpret_cast<float*>(ws + F.Y);
      const size_t ystride = (size_t)cols * C;
      EST_TRY(nt_f16_launch(wf, (size_t)C * K, act, (size_t)cols * K, C, (int)cols, K, words + l, Y, C, pl.S, ystride, stream));
      if (D.resident) {
        EST_TRY(dfepe_est_norm_fwd_r(Y, C, pl.S, ystride, C, B, N, gamma[l], beta[l], eps, slope, out, (size_t)cols * C, out_b, (size_t)cols * C, rstd,
                                     stream));
      } else {
        const int sp = row_splits(B, C, N);
        EST_TRY(dfepe_est_norm_fwd(Y, C, C, B, N, gamma[l], beta[l], eps, slope, out, (size_t)cols * C, out_b, (size_t)cols * C, rstd, sp,
                                   sp > 1 ? reinterpret_cast<float*>(ws + F.npart) : nullptr, stream));
      }
    }
    act = out;
  }
  return dfepe_est_head_fwd(act, (size_t)cols * D.Co[D.n - 1], D.Co[D.n - 1], (int)cols, w_head, b_head, logits, stream);
}

// every gradient of one dfepe_est_forward(saved != null): g_W[l] [Co][Ci], g_bias[l] [Co] (zeros: the bias cancels in the
// normalisation), g_gamma[l], g_beta[l] [Co], g_w_head [C], g_b_head [1] or null, gx or null (needs need_gx at the forward; element
// (b, c, n) at b * gx_stride_b + c * gx_stride_c + n, like x);
// prep: what the forward was given (null: the transposed weight planes are in `saved`)
extern "C" int dfepe_est_backward(const float* g_logits, long B, int C0, int N, int n_hidden, const float* const* W, const float* const* gamma,
                                  const float* const* beta, const int* Co, const int* Ci, const float* w_head, float slope, const void* saved,
                                  void* workspace, const void* prep, float* const* g_W, float* const* g_bias, float* const* g_gamma,
                                  float* const* g_beta, float* g_w_head, float* g_b_head, float* gx, long gx_stride_b, long gx_stride_c, void* stream) {
  PassDims D;
  EST_TRY(pass_dims(D, n_hidden, Co, Ci, B, C0, N));
  if (!g_logits || !W || !gamma || !beta || !w_head || !saved || !workspace || !g_W || !g_bias || !g_gamma || !g_beta || !g_w_head)
    return DFEPE_ERR_INVALID_ARG;
  if (((uintptr_t)workspace & 15) || ((uintptr_t)saved & 15) || ((uintptr_t)prep & 15)) return DFEPE_ERR_INVALID_ARG;
  if (gx && (gx_stride_b <= 0 || gx_stride_c <= 0)) return DFEPE_ERR_INVALID_ARG;
  const bool need_gx = gx != nullptr;
  const SavedLayout S = saved_layout(D, need_gx);
  const BwdLayout L = bwd_layout(D, need_gx);
  const PrepLayout P = prep_layout(D.n, Co, Ci);
  char* ws = static_cast<char*>(workspace);
  const char* sv = static_cast<const char*>(saved);
  const char* pp = static_cast<const char*>(prep);
  const long cols = D.cols;
  const int n = D.n, Clast = D.Co[n - 1];
  const float* src[kSegMax]; float* dst[kSegMax]; int rows[kSegMax], ccols[kSegMax], lds_[kSegMax], ldd_[kSegMax];
  int nseg = 0;
  auto seg = [&](const float* s, int r, int c, float* d, int ld_src = 0, int ld_dst = 0) {
    src[nseg] = s; rows[nseg] = r; ccols[nseg] = c; dst[nseg] = d; lds_[nseg] = ld_src; ldd_[nseg] = ld_dst; ++nseg;
  };
  auto flush = [&]() -> int {
    if (nseg == 0) return DFEPE_OK;
    const int rc = colsum_launch(nseg, src, rows, ccols, dst, lds_, ldd_, stream);
    nseg = 0;
    return rc;
  };
  // head (the bias gradient's partial sums ride in the same launch)
  const int nblk = 512;
  float* hpart = reinterpret_cast<float*>(ws + L.hpart);
  float* bpart = reinterpret_cast<float*>(ws + L.bpart);
  EST_TRY(dfepe_est_head_dw(sv + S.act[n], (size_t)cols * Clast, Clast, (int)cols, nblk, g_logits, hpart, g_b_head ? bpart : nullptr, stream));
  seg(hpart, nblk, Clast, g_w_head);
  if (g_b_head) seg(bpart, nblk, 1, g_b_head);
  // gamma == 0 fixes (one launch at the end, or layer by layer) and the weight gradients (likewise): see bwd_layout
  const float* f_dA[kTabMax]; const float* f_dl[kTabMax]; const float* f_wh[kTabMax]; const void* f_out[kTabMax]; size_t f_outs[kTabMax];
  const void* f_in[kTabMax]; size_t f_ins[kTabMax]; const float* f_W[kTabMax]; int f_Ci[kTabMax]; const float* f_rstd[kTabMax];
  const float* f_gamma[kTabMax]; int f_C[kTabMax]; float* f_dg[kTabMax]; const void* f_dYn[kTabMax]; size_t f_dyns[kTabMax];
  const float* f_Wn[kTabMax]; int f_Cn[kTabMax];
  int nfix = 0;
  const void* t_dY[kTabMax]; size_t t_dys[kTabMax]; int t_Co[kTabMax]; const void* t_X[kTabMax]; size_t t_xs[kTabMax]; int t_K[kTabMax];
  int t_sl[kTabMax]; float* t_part[kTabMax];
  int ntn = 0;
  float* dA = reinterpret_cast<float*>(ws + L.dA);
  bool pending = false;  // dY / dg / db of the current layer already written by the fused data gradient of the layer above
  Plan up{true, 1};      // how the data gradient above this layer left dA (fused = false: `S` partials in dA)
  for (int l = n - 1; l >= 0; --l) {
    const int C = D.Co[l], K = D.K[l];
    const void* a_out = sv + S.act[l + 1];
    const void* a_in = sv + S.act[l];
    const float* rstd = reinterpret_cast<const float*>(sv + S.rstd[l]);
    char* dY = ws + L.dY[l];
    float* dg = reinterpret_cast<float*>(ws + L.dg[l]);
    float* db = reinterpret_cast<float*>(ws + L.db[l]);
    const bool under_head = (l == n - 1);
    if (!pending) {
      const float* up_dA = under_head ? nullptr : dA;
      const float* up_dl = under_head ? g_logits : nullptr;
      const float* up_wh = under_head ? w_head : nullptr;
      if (under_head && D.fusable) {
        EST_TRY(dfepe_est_in_bwd(nullptr, up_dl, up_wh, a_out, (size_t)cols * C, rstd, gamma[l], beta[l], slope, C, (int)cols, dY, (size_t)cols * C, dg,
                                 db, stream));
      } else if (D.resident) {
        EST_TRY(dfepe_est_in_bwd_r(up_dA, C, up.S, (size_t)cols * C, up_dl, up_wh, a_out, (size_t)cols * C, rstd, gamma[l], beta[l], slope, C, B, N, dY,
                                   (size_t)cols * C, dg, db, stream));
      } else {
        const int sp = row_splits(B, C, N);
        EST_TRY(dfepe_est_in_bwd_n(up_dA, up_dl, up_wh, a_out, (size_t)cols * C, rstd, gamma[l], beta[l], slope, C, B, N, dY, (size_t)cols * C, dg, db,
                                   sp, sp > 1 ? reinterpret_cast<float*>(ws + L.npart) : nullptr, stream));
      }
    }
    if (N <= kFixMaxN) {
      // the upstream gradient of the fix: the head's rank-one form, else recomputed for the channel from dY of the layer above (alive in
      // both layouts until the layer below has been written) -- dA is transient (split-K partials, one buffer for all layers) or never existed
      const int i = nfix;
      f_dA[i] = nullptr; f_dl[i] = under_head ? g_logits : nullptr; f_wh[i] = under_head ? w_head : nullptr;
      f_out[i] = a_out; f_outs[i] = (size_t)cols * C; f_in[i] = a_in; f_ins[i] = (size_t)cols * K; f_W[i] = W[l]; f_Ci[i] = D.Ci[l];
      f_rstd[i] = rstd; f_gamma[i] = gamma[l]; f_C[i] = C; f_dg[i] = dg;
      f_dYn[i] = under_head ? nullptr : ws + L.dY[l + 1]; f_dyns[i] = under_head ? 0 : (size_t)cols * D.Co[l + 1];
      f_Wn[i] = under_head ? nullptr : W[l + 1]; f_Cn[i] = under_head ? 0 : D.Co[l + 1];
      if (L.keep_all) ++nfix;
      else
        EST_TRY(dfepe_est_dgamma_zero(nullptr, f_dl[i], f_wh[i], f_out[i], f_outs[i], f_in[i], f_ins[i], f_W[i], f_Ci[i], f_Ci[i], f_rstd[i],
                                      f_gamma[i], slope, C, N, B, dg, f_dYn[i], f_dyns[i], f_Wn[i], C, f_Cn[i], stream));
    }
    pending = false;
    // dW = dY^T X
    float* partw = reinterpret_cast<float*>(ws + L.partw[l]);
    if (L.keep_all) {
      t_dY[ntn] = dY; t_dys[ntn] = (size_t)cols * C; t_Co[ntn] = C; t_X[ntn] = a_in; t_xs[ntn] = (size_t)cols * K; t_K[ntn] = K; t_sl[ntn] = L.slices[l];
      t_part[ntn] = partw; ++ntn;
    } else {
      EST_TRY(dfepe_est_gemm_tn(dY, (size_t)cols * C, C, a_in, (size_t)cols * K, K, (int)cols, L.slices[l], partw, stream));
    }
    if (nseg + 4 > kSegMax) EST_TRY(flush());  // (never with <= 8 layers: kSegMax >= 2 + 4 kTabMax; a flush before the fixes at the end
                                               // would sum uncorrected d gamma partials)
    seg(dg, (int)B, C, g_gamma[l]);
    seg(db, (int)B, C, g_beta[l]);
    if (K == D.Ci[l]) seg(partw, L.slices[l], C * K, g_W[l]);
    else seg(partw, L.slices[l], C * K, g_W[l], K, D.Ci[l]);  // the first layer: its K - C0 zero-padded input channels dropped
    seg(nullptr, 0, C, g_bias[l]);
    if (l > 0 || need_gx) {
      const void* WT = pp ? static_cast<const void*>(pp + P.wt[l]) : static_cast<const void*>(sv + S.wt[l]);
      if (l == 0) {
        EST_TRY(nt_bf16_launch(WT, (size_t)K * C, dY, (size_t)cols * C, K, (int)cols, C, 2, nullptr, 0, 1, 0, gx, C0, N, gx_stride_b, gx_stride_c, stream));
      } else {
        up = dgrad_plan(D, l);
        if (up.fused) {
          EST_TRY(dfepe_est_dgrad_in_bwd(WT, (size_t)K * C, dY, (size_t)cols * C, K, (int)cols, C, a_in, (size_t)cols * K,
                                         reinterpret_cast<const float*>(sv + S.rstd[l - 1]), gamma[l - 1], beta[l - 1], slope, ws + L.dY[l - 1],
                                         (size_t)cols * K, reinterpret_cast<float*>(ws + L.dg[l - 1]), reinterpret_cast<float*>(ws + L.db[l - 1]),
                                         stream));
          pending = true;
        } else {
          EST_TRY(nt_bf16_launch(WT, (size_t)K * C, dY, (size_t)cols * C, K, (int)cols, C, 2, dA, K, up.S, (size_t)cols * K, nullptr, 0, 0, 0, 0, stream));
        }
      }
    }
  }
  if (ntn > 0) EST_TRY(dfepe_est_gemm_tn_multi(ntn, t_dY, t_dys, t_Co, t_X, t_xs, t_K, (int)cols, t_sl, t_part, stream));
  if (nfix > 0)
    EST_TRY(dfepe_est_dgamma_zero_multi(nfix, f_dA, f_dl, f_wh, f_out, f_outs, f_in, f_ins, f_W, f_Ci, f_rstd, f_gamma, f_C, f_dg, f_dYn, f_dyns, f_Wn,
                                        f_Cn, slope, N, B, stream));
  return flush();
}
