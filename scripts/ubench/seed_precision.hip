// Accuracy of the fp64 hardware seeds (v_rcp_f64 / v_rsq_f64) on gfx950, raw and after one / two Newton-Raphson steps, against
// the correctly rounded host results.  Decides how many steps rcp_nr / rsqrt_nr need (csrc/dfepe_math.h).
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/seed_precision.hip -o gpurun_out/seedp && gpurun_out/seedp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__global__ void k(const double* x, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = fma(r0, fma(-v, r0, 1.0), r0);
  double r2 = fma(r1, fma(-v, r1, 1.0), r1);
  const double a = fabs(v), h = -0.5 * a;
  double q0 = __builtin_amdgcn_rsq(a);
  double q1 = fma(q0, fma(h, q0 * q0, 0.5), q0);
  double q2 = fma(q1, fma(h, q1 * q1, 0.5), q1);
  out[6 * i + 0] = r0; out[6 * i + 1] = r1; out[6 * i + 2] = r2;
  out[6 * i + 3] = q0; out[6 * i + 4] = q1; out[6 * i + 5] = q2;
}

int main() {
  const int n = 1 << 20;
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> m(1.0, 2.0);
  std::uniform_int_distribution<int> e(-300, 300), s(0, 1);
  std::vector<double> x(n), o(6 * (size_t)n);
  for (int i = 0; i < n; ++i) x[i] = std::ldexp(m(g), i < n / 2 ? (e(g) % 40) : e(g)) * (s(g) ? 1.0 : -1.0);
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * (size_t)n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
  double err[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double r = 1.0 / x[i], q = 1.0 / std::sqrt(std::fabs(x[i]));
    for (int c = 0; c < 3; ++c) err[c] = std::fmax(err[c], std::fabs(o[6 * (size_t)i + c] - r) / std::fabs(r));
    for (int c = 3; c < 6; ++c) err[c] = std::fmax(err[c], std::fabs(o[6 * (size_t)i + c] - q) / q);
  }
  printf("v_rcp_f64: raw %.3e, +1 NR %.3e, +2 NR %.3e (max relative error over %d values, exponents -300..300)\n", err[0], err[1], err[2], n);
  printf("v_rsq_f64: raw %.3e, +1 NR %.3e, +2 NR %.3e\n", err[3], err[4], err[5]);
  // specials
  double sp[4] = {0.0, 1e-320, 1e308, INFINITY};
  hipMemcpy(dx, sp, 32, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, dx, dout, 4);
  hipMemcpy(o.data(), dout, 24 * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; ++i) printf("x = %g: rcp raw %g, +1 NR %g; rsq raw %g, +1 NR %g\n", sp[i], o[6 * i], o[6 * i + 1], o[6 * i + 3], o[6 * i + 4]);
  return 0;
}
