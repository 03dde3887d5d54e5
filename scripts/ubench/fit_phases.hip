// Where the forward fit spends its cycles: the body of csrc/w8pt16_body.h compiled with -DDFEPE_PHASE_CLOCKS (lane 0 of every
// wavefront stamps the shader clock at each phase marker), run on B = 4096 synthetic pairs of N = 100 correspondences (logits in,
// every training output written), stamps averaged over the wavefronts.  Diagnostic only: the stamps (a scalar memory-clock read
// and a one-lane store each) add ~2 % to the kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -DDFEPE_PHASE_CLOCKS -Iinclude \
//         -Ipytorch-deepfepe_amd/csrc scripts/ubench/fit_phases.hip -o ab_libs/fit_phases && ab_libs/fit_phases
#include "dfepe_common.h"
#include "w8pt16_body.h"
#include <cstdio>
#include <random>
#include <vector>

__device__ unsigned long long* g_dfepe_phase_clk;

__global__ void __launch_bounds__(256) fit_kernel(const W8Args A) {
  __shared__ double xch[16 * 36];
  const int row = (int)(threadIdx.x >> 4);
  const int pair = (int)blockIdx.x * 16 + row;
  if (pair >= A.B) return;
  w8pt16_fwd_pair<7, true, true>(A, pair, xch + row * 36);
}

int main() {
  const int B = 4096, N = 100;
  std::mt19937 g(1);
  std::uniform_real_distribution<float> ux(0.f, 1241.f), uy(0.f, 376.f), ul(-2.f, 2.f), un(-1.f, 1.f);
  std::vector<float> m((size_t)B * N * 4), lg((size_t)B * N);
  for (size_t i = 0; i < (size_t)B * N; ++i) {  // a smooth map plus noise: the moment matrix has a spread spectrum like real pairs
    const float x = ux(g), y = uy(g);
    m[4 * i] = x; m[4 * i + 1] = y; m[4 * i + 2] = 0.97f * x + 0.02f * y + 11.f + un(g); m[4 * i + 3] = 1.01f * y - 0.01f * x - 3.f + un(g);
    lg[i] = ul(g);
  }
  float *dm, *dl, *dF, *dres, *depi, *dsave, *dw;
  hipMalloc(&dm, m.size() * 4); hipMalloc(&dl, lg.size() * 4); hipMalloc(&dF, B * 9 * 4); hipMalloc(&dres, lg.size() * 4);
  hipMalloc(&depi, lg.size() * 4); hipMalloc(&dsave, (size_t)B * DFEPE_SAVE_FLOATS * 4); hipMalloc(&dw, lg.size() * 4);
  hipMemcpy(dm, m.data(), m.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dl, lg.data(), lg.size() * 4, hipMemcpyHostToDevice);
  const int waves = B / 4;
  unsigned long long* dclk;
  hipMalloc(&dclk, (size_t)waves * 16 * 8);
  hipMemset(dclk, 0, (size_t)waves * 16 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_dfepe_phase_clk), &dclk, sizeof(dclk));
  W8Args A{};
  A.pts1 = dm; A.pts2 = nullptr; A.wts = dl; A.B = B; A.Bm = B; A.N = N; A.hw_sx = 2.f / 1241.f; A.hw_sy = 2.f / 376.f; A.clamp_at = 0.5f;
  A.F_out = dF; A.residual = dres; A.epi_res = depi; A.save = dsave; A.weights_out = dw; A.logits_mode = 1; A.variant = 0; A.row_per_pair = false;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(fit_kernel, dim3(B / 16), dim3(256), 0, 0, A);
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(fit_kernel, dim3(B / 16), dim3(256), 0, 0, A);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> clk((size_t)waves * 16);
  hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost);
  const char* names[13] = {"P0 loads, softmax, centroid sums", "P1 Hartley scale", "P2 moments", "P3 partial sums -> M", "P4 tridiagonalisation",
                           "P4b multisection", "P4c twisted factorisation", "P4d back-transform", "P5 orientation", "P5b rank-2 step + de-normalisation",
                           "P5s save record", "P6 per-correspondence outputs", "end"};
  double sum[13] = {0}, mx[13] = {0};
  unsigned long long first = ~0ull, last = 0;
  for (int w = 0; w < waves; ++w) {
    const unsigned long long* c = &clk[(size_t)w * 16];
    for (int k = 0; k < 12; ++k) { const double d = (double)(c[k + 1] - c[k]); sum[k] += d; if (d > mx[k]) mx[k] = d; }
    sum[12] += (double)(c[12] - c[0]); if ((double)(c[12] - c[0]) > mx[12]) mx[12] = (double)(c[12] - c[0]);
    if (c[0] < first) first = c[0];
    if (c[12] > last) last = c[12];
  }
  printf("kernel (with stamps): %.2f us per launch (HIP events, 20 launches); first P0 stamp -> last end stamp over all %d wavefronts (several launches, not comparable): %llu cycles\n",
         ms * 1e3 / 20, waves, last - first);
  printf("%-44s %10s %10s\n", "phase (shader-clock cycles)", "mean", "max");
  for (int k = 0; k < 12; ++k) printf("%-44s %10.1f %10.1f  (%4.1f %%)\n", names[k], sum[k] / waves, mx[k], 100.0 * sum[k] / sum[12]);
  printf("%-44s %10.1f %10.1f\n", "P0 .. end", sum[12] / waves, mx[12]);
  return 0;
}
