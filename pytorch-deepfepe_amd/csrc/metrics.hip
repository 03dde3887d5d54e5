// metrics_summary -- the reductions of the validation summary on the device.
//
// Replaces the numpy post-processing of write_metrics_summary (deepFEPE/train_good_utils.py:758-856) over the per-pair
// results of val_rt (:553-646), which the reference collects on the host through a process pool:
//   epipolar-distance inlier ratios at 0.1 and 1.0 px (:776-777), F1 of "est < th" against "gt < th" (:788-797),
//   medians of the pose errors (:799-805), their maxima (:811-816) and the cumulative error ratios at the thresholds
//   0.01 ... 180 degrees (np.histogram + cumsum, :829-853).
// Two launches (the epipolar counts over B*N values, the pose-error statistics over B values) leave raw integer counts,
// maxima and medians in one small buffer: ONE device-to-host copy per validation epoch instead of one per pair.
// Integer atomics only: the result does not depend on the order of arrival.
#include "dfepe_common.h"

namespace {

constexpr int kNumThs = 13;
// np.histogram compares in float64 against float64 edges: a float32 error exactly at a float-rounded edge must land where numpy puts it
__constant__ double kThs[kNumThs] = {0.0, 0.01, 0.03, 0.05, 0.1, 0.3, 0.5, 1.0, 2.0, 5.0, 10.0, 90.0, 180.0};

__device__ __forceinline__ unsigned wave_count(bool p) { return (unsigned)__popcll(__ballot(p)); }

// counts[0..7]: est<0.1, est<1, TP/FP/FN at 0.1, TP/FP/FN at 1.0
__global__ void __launch_bounds__(256) epi_counts_kernel(const float* __restrict__ est, const float* __restrict__ gt, size_t n,
                                                         unsigned long long* __restrict__ counts) {
  unsigned c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float e = est[i];
    const bool e01 = e < 0.1f, e1 = e < 1.0f;
    c[0] += e01; c[1] += e1;
    if (gt != nullptr) {
      const float g = gt[i];
      const bool g01 = g < 0.1f, g1 = g < 1.0f;
      c[2] += (g01 && e01); c[3] += (!g01 && e01); c[4] += (g01 && !e01);
      c[5] += (g1 && e1);   c[6] += (!g1 && e1);   c[7] += (g1 && !e1);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    unsigned v = c[k];
    v += __shfl_xor(v, 32, WAVE); v += __shfl_xor(v, 16, WAVE); v += __shfl_xor(v, 8, WAVE);
    v += __shfl_xor(v, 4, WAVE);  v += __shfl_xor(v, 2, WAVE);  v += __shfl_xor(v, 1, WAVE);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&counts[k], (unsigned long long)v);
  }
}

// per error vector v (q then t): hist[12] bin counts of np.histogram(v, ths), max (as float bits), the two middle order
// statistics (rank counting: O(B^2) compares, B is a validation set of a few thousand pairs)
__global__ void __launch_bounds__(256) err_stats_kernel(const float* __restrict__ err_q, const float* __restrict__ err_t, int B,
                                                        unsigned long long* __restrict__ hist, unsigned* __restrict__ maxbits,
                                                        float* __restrict__ mids) {
  const int which = blockIdx.y;
  const float* v = which ? err_t : err_q;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < B;
  const float x = live ? v[i] : 0.0f;
  // np.histogram: bins [ths[k], ths[k+1]), the last one closed on the right; values outside [0, 180] are dropped
  int bin = -1;
  const double xd = (double)x;
  if (live && xd >= kThs[0] && xd <= kThs[kNumThs - 1]) {
    bin = kNumThs - 2;
#pragma unroll
    for (int k = kNumThs - 2; k >= 0; --k)
      if (xd < kThs[k + 1]) bin = k;
  }
  for (int k = 0; k < kNumThs - 1; ++k) {
    const unsigned cnt = wave_count(bin == k);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&hist[which * (kNumThs - 1) + k], (unsigned long long)cnt);
  }
  // non-negative floats order like their bit patterns, and the canonical NaN pattern sits above +inf: one NaN error makes the
  // maximum NaN, as np.amax does (a diverged run must stay visible, train_good_utils.py:811-816)
  if (live && (x >= 0.0f || x != x)) atomicMax(&maxbits[which], (x != x) ? 0x7FC00000u : __float_as_uint(x));
  if (live) {
    int less = 0, equal = 0, nans = 0;
    for (int j = 0; j < B; ++j) {
      const float y = v[j];
      less += (y < x) ? 1 : 0;
      equal += (y == x) ? 1 : 0;
      nans += (y != y) ? 1 : 0;
    }
    const int k0 = (B - 1) / 2, k1 = B / 2;  // the middle order statistics (the same one for odd B)
    if (nans) {  // np.median of a vector with a NaN is NaN; every lane sees the same count, so nobody writes anything else
      mids[2 * which] = mids[2 * which + 1] = __uint_as_float(0x7FC00000u);
    } else {
      if (less <= k0 && k0 < less + equal) mids[2 * which] = x;  // every lane that qualifies writes the same value
      if (less <= k1 && k1 < less + equal) mids[2 * which + 1] = x;
    }
  }
}

}  // namespace

extern "C" size_t dfepe_metrics_summary_bytes(void) { return 8 * 8 + 2 * (kNumThs - 1) * 8 + 2 * 4 + 4 * 4 + 8; }

extern "C" int dfepe_metrics_summary(const float* epi_est, const float* epi_gt, size_t n_epi, const float* err_q, const float* err_t,
                                     int B, void* out, void* stream) {
  if (!out || (n_epi && !epi_est) || B < 0 || (B && (!err_q || !err_t))) return DFEPE_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(out) & 7u) return DFEPE_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(out, 0, dfepe_metrics_summary_bytes(), st) != hipSuccess) return DFEPE_ERR_HIP;
  unsigned long long* counts = static_cast<unsigned long long*>(out);
  unsigned long long* hist = counts + 8;
  unsigned* maxbits = reinterpret_cast<unsigned*>(hist + 2 * (kNumThs - 1));
  float* mids = reinterpret_cast<float*>(maxbits + 2);
  if (n_epi) {
    const unsigned blocks = (unsigned)((n_epi + 256 * 8 - 1) / (256 * 8));
    hipLaunchKernelGGL(epi_counts_kernel, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, st, epi_est, epi_gt, n_epi, counts);
  }
  if (B) hipLaunchKernelGGL(err_stats_kernel, dim3((B + 255) / 256, 2), dim3(256), 0, st, err_q, err_t, B, hist, maxbits, mids);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
