// nn_match — two-way nearest-neighbour matching of unit-norm descriptors, the step that PRODUCES the (B,N,4)
// correspondences of the solver.
//
// Replaces, per image pair, the host/numpy loop of get_matches_from_SP (deepFEPE/train_good_utils.py:683-716):
//   matching_mask = SP_tracker.nn_match_two_way(desc1.T, desc2.T, nn_thresh)      (:687-691)
//   xs / offsets / quality gathered through crop_or_pad_choice                     (:693-716, utils_misc.py:139-161)
// nn_match_two_way lives in the un-vendored `superpoint` package (eric-yyjau/pytorch-superpoint,
// models/model_wrap.py, PointTracker; README.md:37-40 installs it unpinned); its published algorithm (Magic Leap's
// SuperPoint demo) is restated in oracle/deepf_oracle.py:nn_match_two_way:
//   dmat = sqrt(2 - 2 clip(desc1^T desc2, -1, 1)); idx = argmin_j dmat[i,j]; keep = dmat[i,idx] < nn_thresh and
//   argmin_i dmat[i,idx[i]] == i; matches = (i, idx[i], dmat[i,idx[i]]) for the kept i in increasing order.
//
// Kernel 1 (nn_match_tile): the only GEMM-shaped op of the whole path, so it runs on the matrix cores in exact fp32
// (v_mfma_f32_32x32x2_f32: the descriptors are fp32 and the arg-min decisions must not see bf16 rounding).  One
// 256-thread workgroup per 128x128 tile of dmat, 2x2 wavefronts of 64x64 (2x2 MFMA tiles of 32x32, 64 accumulator
// VGPRs), K = D in chunks of 16 staged K-major in LDS ([k][row]: a lane's MFMA operand A[i=l&31][k=l>>5] is then a
// conflict-free ds_read_b32), double-buffered with the next chunk prefetched into registers.  dmat never reaches HBM:
// the epilogue turns the accumulators into squared distances and folds them into per-row and per-column minima, packed
// as (value bits << 32 | index) so that one 64-bit atomicMin per row/column realises numpy's first-occurrence argmin.
// Kernel 2 (nn_match_finish): threshold + mutual check + order-preserving compaction (ballot / popcount).
// Kernel 3 (gather_matches): the crop/pad gather into xs [B,N,4], offsets [B,N,4], quality [B,N].
#include "dfepe_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

constexpr int TM = 128;  // tile edge (rows of image 1 x rows of image 2)
constexpr int TK = 16;   // K chunk: 2 x 2 x 16 x 128 floats = 32 KiB of LDS per workgroup, so four workgroups share a CU and one
                         // workgroup's epilogue / prologue overlaps the others' MFMA phases
constexpr unsigned kNoVal = 0xffffffffu;

// Squared distance of two unit descriptors from their dot product: t = 2 - 2 clip(dot, -1, 1), bit-identical to the
// reference's fp32 expression under the square root (one rounding).  The minima are taken over t; the reference takes
// them over sqrt(t), which is the same arg-min except when two DIFFERENT t round to the same square root and the larger
// one comes first -- a 1-ulp near-tie that already depends on the summation order of the BLAS behind np.dot.  Exact
// ties (duplicated descriptors, dots clipped at 1) keep numpy's first-occurrence rule.  The score is sqrtf(t_min).
__device__ __forceinline__ float dot_to_t(float dot) { return fmaf(-2.0f, __builtin_amdgcn_fmed3f(dot, -1.0f, 1.0f), 2.0f); }

template <int CTRL>
__device__ __forceinline__ u64 exchange_u64(u64 v) {
  union { u64 k; int i[2]; } a, r;
  a.k = v;
  if (CTRL == 0) {  // lane ^ 16: crosses the 16-lane DPP rows
    r.i[0] = __shfl_xor(a.i[0], 16, 64);
    r.i[1] = __shfl_xor(a.i[1], 16, 64);
  } else {
    r.i[0] = __builtin_amdgcn_update_dpp(a.i[0], a.i[0], (CTRL == 0) ? 0xB1 : CTRL, 0xf, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(a.i[1], a.i[1], (CTRL == 0) ? 0xB1 : CTRL, 0xf, 0xf, false);
  }
  return r.k;
}
// One step of a transposing min-reduction over the 32 lanes of a half: CNT keys per lane -> CNT/2; the lane with the
// decision bit set keeps the upper half of its array and hands the lower half to its partner (any partner on the other
// side of the bit works: lane^16, row_mirror, row_half_mirror, quad_perm xor 2 / xor 1).  After the five steps lane j
// holds the minimum over the half of key number j: 31 exchanges instead of 32 x 5.
template <int CNT, int CTRL>
__device__ __forceinline__ void halve_min(u64* a, bool upper) {
  constexpr int H = CNT / 2;
#pragma unroll
  for (int k = 0; k < H; ++k) {
    const u64 lo = a[k], hi = a[k + H];
    const u64 got = exchange_u64<CTRL>(upper ? lo : hi);
    const u64 keep = upper ? hi : lo;
    a[k] = (got < keep) ? got : keep;
  }
}

__global__ void __launch_bounds__(256, 4)
nn_match_tile_kernel(const float* __restrict__ desc1, const float* __restrict__ desc2, int N1, int N2, int D,
                     u64* __restrict__ rowkey, u64* __restrict__ colkey) {
  __shared__ float lds[2][2][TK][TM];  // [buffer][image][k][row]: 32 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware tile order: workgroup w runs on XCD w % 8 (round-robin dispatch), and each XCD has its own 4 MiB L2.  The
  // tiles of one image pair share its 2 x N x D descriptors, so every XCD is given whole pairs: the workgroups an XCD
  // has in flight (32 CUs x 4) then cover ~2 pairs = ~4 MiB of descriptors instead of a slice of 16 different pairs.
  const int tn = (N2 + TM - 1) / TM, tm = (N1 + TM - 1) / TM;
  const unsigned total = gridDim.x;
  unsigned w = blockIdx.x;
  if ((total & 7u) == 0u) w = (w & 7u) * (total >> 3) + (w >> 3);
  const int b = (int)(w / (unsigned)(tm * tn));
  const int trem = (int)(w % (unsigned)(tm * tn));
  const int m0 = (trem / tn) * TM, n0 = (trem % tn) * TM;

  // loader role: threads 0..127 stream one descriptor of image 1 each (64 contiguous bytes per chunk), 128..255 image 2
  const int img = tid >> 7, lr = tid & 127;
  const int grow = (img ? n0 : m0) + lr;
  const bool rvalid = grow < (img ? N2 : N1);
  const float* src = (img ? desc2 + (size_t)b * N2 * D : desc1 + (size_t)b * N1 * D) + (size_t)(rvalid ? grow : 0) * D;
  float4 pre[TK / 4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int q = 0; q < TK / 4; ++q)
      pre[q] = rvalid ? reinterpret_cast<const float4*>(src + k0)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto lstore = [&](int buf) {
    float* dst = &lds[buf][img][0][lr];
#pragma unroll
    for (int q = 0; q < TK / 4; ++q) {
      dst[(4 * q + 0) * TM] = pre[q].x;
      dst[(4 * q + 1) * TM] = pre[q].y;
      dst[(4 * q + 2) * TM] = pre[q].z;
      dst[(4 * q + 3) * TM] = pre[q].w;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int h = lane >> 5, jl = lane & 31;
  gload(0);
  lstore(0);
  __syncthreads();
  const int nk = D / TK;
  for (int c = 0; c < nk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nk) gload((c + 1) * TK);
    const float* Ab = &lds[buf][0][h][wr * 64 + jl];
    const float* Bb = &lds[buf][1][h][wc * 64 + jl];
#pragma unroll
    for (int kk = 0; kk < TK / 2; ++kk) {
      const float a0 = Ab[2 * kk * TM], a1 = Ab[2 * kk * TM + 32];
      const float b0 = Bb[2 * kk * TM], b1 = Bb[2 * kk * TM + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (c + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----------
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = dot_to_t(acc[i][j][r]);

  const int cg0 = n0 + wc * 64 + jl;  // global column of my element in column tile 0 (tile 1: +32)
  const int rbase = m0 + wr * 64 + 4 * h;
  const bool c0ok = cg0 < N2, c1ok = cg0 + 32 < N2;
  // column minima first (they read the accumulators in place): in-lane over my 32 rows, visited in increasing order so
  // that strict < keeps the first row, then the two halves through one 64-bit exchange
  u64 colk[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    unsigned bestv = kNoVal, bestr = kNoVal;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rg = rbase + rt * 32 + (r & 3) + 8 * (r >> 2);
        const unsigned k = (rg < N1) ? __float_as_uint(acc[rt][ct][r]) : kNoVal;
        if (k < bestv) { bestv = k; bestr = (unsigned)rg; }
      }
    }
    const u64 key = ((u64)bestv << 32) | bestr;
    const u64 other = (u64)__shfl_xor((long long)key, 32, 64);
    colk[ct] = (other < key) ? other : key;
  }
  {
    const u64 mycol = h ? colk[1] : colk[0];
    const int cg = cg0 + 32 * h;
    if (cg < N2 && (unsigned)(mycol >> 32) != kNoVal) atomicMin(&colkey[(size_t)b * N2 + cg], mycol);
  }
  // row minima: in-lane over the two column tiles (ties keep the smaller column), then the transposing reduction over
  // the 32 lanes of the half; key number q = 16 rt + r ends up in lane q of the half
  u64 rk[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const unsigned k0 = c0ok ? __float_as_uint(acc[q >> 4][0][q & 15]) : kNoVal;
    const unsigned k1 = c1ok ? __float_as_uint(acc[q >> 4][1][q & 15]) : kNoVal;
    const unsigned cb = (k1 < k0) ? (unsigned)(cg0 + 32) : (unsigned)cg0;
    rk[q] = ((u64)min(k0, k1) << 32) | cb;
  }
  halve_min<32, 0>(rk, (lane & 16) != 0);
  halve_min<16, 0x140>(rk, (lane & 8) != 0);  // row_mirror
  halve_min<8, 0x141>(rk, (lane & 4) != 0);   // row_half_mirror
  halve_min<4, 0x4E>(rk, (lane & 2) != 0);    // quad_perm [2,3,0,1]
  halve_min<2, 0xB1>(rk, (lane & 1) != 0);    // quad_perm [1,0,3,2]
  {
    const int r = jl & 15;
    const int rg = rbase + (jl >> 4) * 32 + (r & 3) + 8 * (r >> 2);
    if (rg < N1 && (unsigned)(rk[0] >> 32) != kNoVal) atomicMin(&rowkey[(size_t)b * N1 + rg], rk[0]);
  }
}

__global__ void __launch_bounds__(1024)
nn_match_finish_kernel(const u64* __restrict__ rowkey, const u64* __restrict__ colkey, int N1, int N2, float nn_thresh,
                       int* __restrict__ m_idx1, int* __restrict__ m_idx2, float* __restrict__ score,
                       int* __restrict__ count) {
  __shared__ int wsum[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int base = 0;
  for (int i0 = 0; i0 < N1; i0 += 1024) {  // one pass for up to 1024 keypoints: a single memory latency per pair
    const int i = i0 + tid;
    bool keep = false;
    unsigned j = 0;
    float d = 0.0f;
    if (i < N1) {
      const u64 key = rowkey[(size_t)b * N1 + i];
      if ((unsigned)(key >> 32) != kNoVal) {
        j = (unsigned)key;
        d = sqrtf(__uint_as_float((unsigned)(key >> 32)));  // the keys carry t = 2 - 2 clip(dot); the score is the distance
        // scores < nn_thresh, and the nearest neighbour of j in image 1 is i again
        keep = (d < nn_thresh) && ((unsigned)colkey[(size_t)b * N2 + j] == (unsigned)i);
      }
    }
    const u64 bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull)), total = __popcll(bal);
    if (lane == 0) wsum[wave] = total;
    __syncthreads();
    int off = base, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wave) off += wsum[w];
      all += wsum[w];
    }
    if (keep) {
      const size_t p = (size_t)b * N1 + off + before;
      m_idx1[p] = i;
      m_idx2[p] = (int)j;
      score[p] = d;
    }
    base += all;
    __syncthreads();
  }
  if (tid == 0) count[b] = base;
}

__global__ void __launch_bounds__(256)
gather_matches_kernel(const float* __restrict__ pts1, const float* __restrict__ pts2, const float* __restrict__ off1,
                      const float* __restrict__ off2, int B, int N1, int N2, const int* __restrict__ m_idx1,
                      const int* __restrict__ m_idx2, const float* __restrict__ score, const int* __restrict__ choice,
                      int n_out, float* __restrict__ xs, float* __restrict__ offsets, float* __restrict__ quality) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * n_out) return;
  const size_t b = t / n_out;
  const int c = choice[t];
  const int i = m_idx1[b * N1 + c], j = m_idx2[b * N1 + c];
  const float2 p1 = reinterpret_cast<const float2*>(pts1)[b * N1 + i];
  const float2 p2 = reinterpret_cast<const float2*>(pts2)[b * N2 + j];
  reinterpret_cast<float4*>(xs)[t] = make_float4(p1.x, p1.y, p2.x, p2.y);
  if (offsets != nullptr) {
    const float2 o1 = reinterpret_cast<const float2*>(off1)[b * N1 + i];
    const float2 o2 = reinterpret_cast<const float2*>(off2)[b * N2 + j];
    reinterpret_cast<float4*>(offsets)[t] = make_float4(o1.x, o1.y, o2.x, o2.y);
  }
  if (quality != nullptr) quality[t] = score[b * N1 + c];
}

}  // namespace

extern "C" size_t dfepe_nn_match_workspace_bytes(int B, int N1, int N2) {
  if (B <= 0 || N1 < 0 || N2 < 0) return 0;
  return (size_t)B * ((size_t)N1 + (size_t)N2) * sizeof(u64);
}

extern "C" int dfepe_nn_match_two_way(const float* desc1, const float* desc2, int B, int N1, int N2, int D, float nn_thresh,
                                      void* workspace, int* m_idx1, int* m_idx2, float* score, int* count, void* stream) {
  if (B < 0 || N1 < 0 || N2 < 0 || D <= 0) return DFEPE_ERR_INVALID_ARG;
  if (!(nn_thresh >= 0.0f)) return DFEPE_ERR_INVALID_ARG;  // the reference raises ValueError for a negative threshold
  if (B == 0) return DFEPE_OK;
  if (!count) return DFEPE_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (N1 == 0 || N2 == 0) {  // np.zeros((3, 0)): no matches
    return (hipMemsetAsync(count, 0, (size_t)B * sizeof(int), st) == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
  }
  if (!desc1 || !desc2 || !workspace || !m_idx1 || !m_idx2 || !score) return DFEPE_ERR_INVALID_ARG;
  if (D % 32 != 0) return DFEPE_ERR_UNSUPPORTED;  // K is consumed in 16-float chunks, two per 128-byte line (SuperPoint: D = 256)
  if (((uintptr_t)desc1 | (uintptr_t)desc2) & 15u) return DFEPE_ERR_INVALID_ARG;
  if (((uintptr_t)workspace) & 7u) return DFEPE_ERR_INVALID_ARG;
  u64* rowkey = static_cast<u64*>(workspace);
  u64* colkey = rowkey + (size_t)B * N1;
  if (hipMemsetAsync(workspace, 0xff, dfepe_nn_match_workspace_bytes(B, N1, N2), st) != hipSuccess) return DFEPE_ERR_HIP;
  const size_t tiles = (size_t)((N2 + TM - 1) / TM) * ((N1 + TM - 1) / TM) * B;
  if (tiles > 0x7fffffffu) return DFEPE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(nn_match_tile_kernel, dim3((unsigned)tiles), dim3(256), 0, st, desc1, desc2, N1, N2, D, rowkey, colkey);
  hipLaunchKernelGGL(nn_match_finish_kernel, dim3(B), dim3(1024), 0, st, rowkey, colkey, N1, N2, nn_thresh, m_idx1, m_idx2,
                     score, count);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

extern "C" int dfepe_gather_matches(const float* pts1, const float* pts2, const float* off1, const float* off2, int B, int N1,
                                    int N2, const int* m_idx1, const int* m_idx2, const float* score, const int* choice,
                                    int n_out, float* xs, float* offsets, float* quality, void* stream) {
  if (B < 0 || N1 <= 0 || N2 <= 0 || n_out < 0) return DFEPE_ERR_INVALID_ARG;
  if (B == 0 || n_out == 0) return DFEPE_OK;
  if (!pts1 || !pts2 || !m_idx1 || !m_idx2 || !choice || !xs) return DFEPE_ERR_INVALID_ARG;
  if (offsets && (!off1 || !off2)) return DFEPE_ERR_INVALID_ARG;
  if (quality && !score) return DFEPE_ERR_INVALID_ARG;
  const size_t n = (size_t)B * n_out;
  hipLaunchKernelGGL(gather_matches_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     pts1, pts2, off1, off2, B, N1, N2, m_idx1, m_idx2, score, choice, n_out, xs, offsets, quality);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
