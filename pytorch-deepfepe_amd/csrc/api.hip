// Library-level entry points of libdfepe_hip.so (see include/dfepe.h).
#include "dfepe_common.h"

extern "C" int dfepe_version(void) { return DFEPE_VERSION; }

extern "C" int dfepe_save_floats(void) { return DFEPE_SAVE_FLOATS; }

extern "C" const char* dfepe_strerror(int code) {
  switch (code) {
    case DFEPE_OK: return "ok";
    case DFEPE_ERR_INVALID_ARG: return "invalid argument (null pointer, bad size, misaligned buffer or bad flags)";
    case DFEPE_ERR_HIP: return "HIP runtime error while configuring or launching a kernel";
    case DFEPE_ERR_UNSUPPORTED: return "request not supported by this build";
    default: return "unknown dfepe error code";
  }
}

// ---- self-test of the row-group primitives (rowgroup.h) ----------------------------------------------------------------
// The row-per-pair kernels rest on DPP encodings (row_newbcast, row_mirror, row_half_mirror, quad_perm) whose semantics
// the host emulation of tests/emu/ can only assume; this kernel applies every primitive to caller-provided data so that
// tests/test_rowgroup_gpu.py can compare them with their definition on the hardware itself.
namespace {
constexpr int kSelftestOutputs = 14 + 2 + 9 + 9 + 2;
__global__ void __launch_bounds__(64) rowgroup_selftest_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                               double* __restrict__ out) {
  const int t = (int)threadIdx.x;
  const double a = x[t], b = y[t];
  const float af = (float)a;
  const int ai = (int)(a * 16.0);
  double r[kSelftestOutputs];
  r[0] = rg_bcast<0>(a);
  r[1] = rg_bcast<5>(a);
  r[2] = rg_bcast<15>(a);
  r[3] = rg_sum(a);
  r[4] = rg_sum_range<2, 8>(a);
  r[5] = rg_xchg<8>(a);
  r[6] = rg_xchg<4>(a);
  r[7] = rg_xchg<2>(a);
  r[8] = rg_xchg<1>(a);
  r[9] = (double)rg_max(af);
  r[10] = (double)rg_sum(ai);
  r[11] = rg_fma_bcast<3>(b, a, b);
  r[12] = (double)rg_sum(af);
  r[13] = (double)rg_bcast<9>(af) + (double)rg_bcast<12>(ai) + (double)rg_lane();
  // the fused broadcast-FMA chains; every DPP operand below is produced by the instruction right in front of the chain (the
  // hazard the chains pad for)
  double m[9], n[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) { m[j] = a + (double)j * b; n[j] = b - (double)j; }
  const double xa = a * b + 1.0;
  r[14] = rg_dot_bcast<1>(xa, m);
  const double xb = a - b;
  r[15] = rg_dot_bcast<6>(xb, m);
  const double xc = a * 3.0;
  rg_axpy_bcast<3>(m, xc, b);
#pragma unroll
  for (int j = 0; j < 9; ++j) r[16 + j] = m[j];
  const double xd = b * b, xe = a + 2.0;
  rg_axpy2_bcast<1>(n, xd, a, xe, b);
#pragma unroll
  for (int j = 0; j < 9; ++j) r[25 + j] = n[j];
  const double xf = a * a;
  r[34] = rg_sum_to8<0>(xf);
  r[35] = rg_sum_to8<5>(xf + 1.0);
#pragma unroll
  for (int k = 0; k < kSelftestOutputs; ++k) out[k * 64 + t] = r[k];
}
}  // namespace

extern "C" int dfepe_selftest_rowgroup(const double* x, const double* y, double* out, void* stream) {
  if (!x || !y || !out) return DFEPE_ERR_INVALID_ARG;
  hipLaunchKernelGGL(rowgroup_selftest_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), x, y, out);
  return (hipGetLastError() == hipSuccess) ? DFEPE_OK : DFEPE_ERR_HIP;
}
