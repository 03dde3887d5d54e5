// w8pt16 -- the row-per-pair kernels of the weighted 8-point fit: launchers.
//
// Grid: one 256-thread workgroup = 16 pairs (4 wavefronts x 4 rows).  B = 4096 pairs -> 256 workgroups = one per CU,
// one wavefront per SIMD; larger batches stack more wavefronts per SIMD.  No block-level barrier; each pair owns 36
// doubles of LDS for one exchange.  Bodies: w8pt16_body.h (forward), w8pt16_bwd_body.h (adjoint).
#include "dfepe_common.h"
#include "w8pt16_body.h"
#include "w8pt16_bwd_body.h"

namespace {

constexpr int kPairsPerBlock = 16;

template <int IT, bool RAW, bool PLAIN>
__global__ void __launch_bounds__(256) w8pt16_fwd_kernel(const W8Args A) {
  __shared__ double xch[kPairsPerBlock * 36];
  const int row = (int)(threadIdx.x >> 4);
  const int pair = (int)blockIdx.x * kPairsPerBlock + row;
  if (pair >= A.B) return;  // a whole row leaves; rows never wait for each other
  w8pt16_fwd_pair<IT, RAW, PLAIN>(A, pair, xch + row * 36);
}

template <int IT, bool RAW, bool PGRAD>
__global__ void __launch_bounds__(256) w8pt16_bwd_kernel(const W8BwdArgs A) {
  const int row = (int)(threadIdx.x >> 4);
  const int pair = (int)blockIdx.x * kPairsPerBlock + row;
  if (pair >= A.B) return;
  w8pt16_bwd_pair_impl<IT, RAW, PGRAD>(A, pair, nullptr);
}

template <bool RAW, bool PLAIN>
void launch_fwd(const W8Args& A, hipStream_t st) {
  const dim3 grid((A.B + kPairsPerBlock - 1) / kPairsPerBlock), block(256);
  const int N = A.N;
  if (N > 128) hipLaunchKernelGGL((w8pt16_fwd_kernel<0, RAW, PLAIN>), grid, block, 0, st, A);  // any N: correspondences re-read per phase
  else if (N <= 16) hipLaunchKernelGGL((w8pt16_fwd_kernel<1, RAW, PLAIN>), grid, block, 0, st, A);
  else if (N <= 32) hipLaunchKernelGGL((w8pt16_fwd_kernel<2, RAW, PLAIN>), grid, block, 0, st, A);
  else if (N <= 64) hipLaunchKernelGGL((w8pt16_fwd_kernel<4, RAW, PLAIN>), grid, block, 0, st, A);
  else if (N <= 112) hipLaunchKernelGGL((w8pt16_fwd_kernel<7, RAW, PLAIN>), grid, block, 0, st, A);
  else hipLaunchKernelGGL((w8pt16_fwd_kernel<8, RAW, PLAIN>), grid, block, 0, st, A);
}

template <bool RAW, bool PGRAD>
void launch_bwd(const W8BwdArgs& A, hipStream_t st) {
  const dim3 grid((A.B + kPairsPerBlock - 1) / kPairsPerBlock), block(256);
  const int N = A.N;
  if (N > 128) hipLaunchKernelGGL((w8pt16_bwd_kernel<0, RAW, PGRAD>), grid, block, 0, st, A);
  else if (N <= 16) hipLaunchKernelGGL((w8pt16_bwd_kernel<1, RAW, PGRAD>), grid, block, 0, st, A);
  else if (N <= 32) hipLaunchKernelGGL((w8pt16_bwd_kernel<2, RAW, PGRAD>), grid, block, 0, st, A);
  else if (N <= 64) hipLaunchKernelGGL((w8pt16_bwd_kernel<4, RAW, PGRAD>), grid, block, 0, st, A);
  else if (N <= 112) hipLaunchKernelGGL((w8pt16_bwd_kernel<7, RAW, PGRAD>), grid, block, 0, st, A);
  else hipLaunchKernelGGL((w8pt16_bwd_kernel<8, RAW, PGRAD>), grid, block, 0, st, A);
}

}  // namespace

// Called by dfepe_w8pt_fwd / dfepe_w8pt_bwd (w8pt_fwd.hip / w8pt_bwd.hip) after argument validation.
int dfepe_w8pt16_fwd_launch(const W8Args& A, bool raw, hipStream_t st) {
  const bool plain = A.variant == 0;
  if (raw) { if (plain) launch_fwd<true, true>(A, st); else launch_fwd<true, false>(A, st); }
  else { if (plain) launch_fwd<false, true>(A, st); else launch_fwd<false, false>(A, st); }
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}

int dfepe_w8pt16_bwd_launch(const W8BwdArgs& A, bool raw, hipStream_t st) {
  const bool pgrad = A.g_p1 != nullptr;
  if (raw) { if (pgrad) launch_bwd<true, true>(A, st); else launch_bwd<true, false>(A, st); }
  else { if (pgrad) launch_bwd<false, true>(A, st); else launch_bwd<false, false>(A, st); }
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
