// Where the BACKWARD fit spends its cycles (9.3 cycles per instruction, 38 % wait): the body of
// csrc/w8pt16_bwd_body.h compiled with -DDFEPE_PHASE_CLOCKS (lane 0 of every wavefront stamps the shader clock at each phase marker),
// the benchmark's instantiation (pixel matches, logits in, g_F only: UP = false), B = 4096 pairs of N = 100, stamps averaged over the
// wavefronts; the forward fit of the same build writes the save records first.  With an argument: the cooperative forward fit instead
// (one four-wavefront workgroup per pair, N = 1000, that many pairs), stamps per wavefront INDEX of the workgroup.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -DDFEPE_PHASE_CLOCKS -Iinclude \
//         -Ipytorch-deepfepe_amd/csrc scripts/ubench/bwd_phases.hip -o scripts/ubench/_build/bwd_phases
#include "dfepe_common.h"
#include "w8pt16_bwd_body.h"
#include <cstdio>
#include <cstdlib>
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
__global__ void __launch_bounds__(256) bwd_kernel(const W8BwdArgs A) {
  const int row = (int)(threadIdx.x >> 4);
  const int pair = (int)blockIdx.x * 16 + row;
  if (pair >= A.B) return;
  w8pt16_bwd_pair_impl<7, true, false, true, 1, false>(A, pair, nullptr);
}
template <int IT>
__global__ void __launch_bounds__(256, (IT <= 4 ? 2 : 1)) coop_kernel(const W8Args A) {
  __shared__ W8Coop co;
  w8pt16_fwd_pair<IT, true, true, 16>(A, (int)blockIdx.x, nullptr, &co, (int)(threadIdx.x >> 4));
}

int main(int argc, char** argv) {
  const bool coop = argc > 1;
  const int B = coop ? atoi(argv[1]) : 4096, N = coop ? 1000 : 100;
  std::mt19937 g(1);
  std::uniform_real_distribution<float> ux(0.f, 1241.f), uy(0.f, 376.f), ul(-2.f, 2.f), un(-1.f, 1.f);
  std::vector<float> m((size_t)B * N * 4), lg((size_t)B * N), gF((size_t)B * 9);
  for (size_t i = 0; i < (size_t)B * N; ++i) {
    const float x = ux(g), y = uy(g);
    m[4 * i] = x; m[4 * i + 1] = y; m[4 * i + 2] = 0.97f * x + 0.02f * y + 11.f + un(g); m[4 * i + 3] = 1.01f * y - 0.01f * x - 3.f + un(g);
    lg[i] = ul(g);
  }
  for (auto& v : gF) v = un(g);
  float *dm, *dl, *dF, *dres, *depi, *dsave, *dw, *dgF, *dgw;
  hipMalloc(&dm, m.size() * 4); hipMalloc(&dl, lg.size() * 4); hipMalloc(&dF, B * 9 * 4); hipMalloc(&dres, lg.size() * 4);
  hipMalloc(&depi, lg.size() * 4); hipMalloc(&dsave, (size_t)B * DFEPE_SAVE_FLOATS * 4); hipMalloc(&dw, lg.size() * 4);
  hipMalloc(&dgF, gF.size() * 4); hipMalloc(&dgw, lg.size() * 4);
  hipMemcpy(dm, m.data(), m.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dl, lg.data(), lg.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dgF, gF.data(), gF.size() * 4, hipMemcpyHostToDevice);
  const int wpb = 4, blocks = coop ? B : B / 16, waves = blocks * wpb;
  unsigned long long* dclk;
  hipMalloc(&dclk, (size_t)waves * 16 * 8);
  hipMemset(dclk, 0, (size_t)waves * 16 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_dfepe_phase_clk), &dclk, sizeof(dclk));
  W8Args A{};
  A.pts1 = dm; A.pts2 = nullptr; A.wts = dl; A.B = B; A.Bm = B; A.N = N; A.hw_sx = 2.f / 1241.f; A.hw_sy = 2.f / 376.f; A.clamp_at = 0.5f;
  A.F_out = dF; A.residual = dres; A.epi_res = depi; A.save = dsave; A.weights_out = dw; A.logits_mode = 1; A.variant = 0; A.row_per_pair = false;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  if (coop) {
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(coop_kernel<4>, dim3(B), dim3(256), 0, 0, A);
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(coop_kernel<4>, dim3(B), dim3(256), 0, 0, A);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> clk((size_t)waves * 16);
    hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost);
    const char* names[13] = {"P0 loads, softmax, centroid sums", "P1 Hartley scale", "P2 moments", "P3 partial sums -> M", "P4 tridiagonalisation",
                             "P4b multisection", "P4c twisted factorisation", "P4d back-transform", "P5 orientation", "P5b rank-2 step + de-normalisation",
                             "P5s save record", "P6 per-correspondence outputs", "end"};
    printf("cooperative forward fit, %d pairs x %d: %.2f us per launch (with stamps)\n", B, N, ms * 1e3 / 20);
    printf("%-44s %12s %12s %12s %12s   (mean shader-clock cycles per wavefront index of the workgroup; wavefront 0 holds row 0)\n", "phase", "wave 0", "wave 1", "wave 2", "wave 3");
    double tot[4] = {0, 0, 0, 0};
    for (int k = 0; k < 12; ++k) {
      double s[4] = {0, 0, 0, 0};
      for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 4; ++w) { const unsigned long long* c = &clk[((size_t)b * 4 + w) * 16]; s[w] += (double)(c[k + 1] - c[k]); }
      printf("%-44s %12.1f %12.1f %12.1f %12.1f\n", names[k], s[0] / blocks, s[1] / blocks, s[2] / blocks, s[3] / blocks);
      for (int w = 0; w < 4; ++w) tot[w] += s[w] / blocks;
    }
    printf("%-44s %12.1f %12.1f %12.1f %12.1f\n", "P0 .. end", tot[0], tot[1], tot[2], tot[3]);
    return 0;
  }
  hipLaunchKernelGGL(fit_kernel, dim3(B / 16), dim3(256), 0, 0, A);
  hipDeviceSynchronize();
  hipMemset(dclk, 0, (size_t)waves * 16 * 8);
  W8BwdArgs Bk{};
  Bk.pts1 = dm; Bk.pts2 = nullptr; Bk.wts = dl; Bk.B = B; Bk.Bm = B; Bk.N = N; Bk.hw_sx = A.hw_sx; Bk.hw_sy = A.hw_sy; Bk.clamp_at = 0.5f;
  Bk.save = dsave; Bk.F_out = dF; Bk.g_F = dgF; Bk.g_w = dgw; Bk.logits_mode = 1; Bk.variant = 0;
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(bwd_kernel, dim3(B / 16), dim3(256), 0, 0, Bk);
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(bwd_kernel, dim3(B / 16), dim3(256), 0, 0, Bk);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> clk((size_t)waves * 16);
  hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost);
  const char* names[8] = {"B0 loads (save record, g_F, correspondences)", "B1 pass A (upstream per-correspondence terms: compiled out)", "B2 uniform: T2 g T1^T",
                          "B3 rank-2 adjoint", "B4 eigenvector adjoint: reflect g_f", "B5 (T - lam I)^+ and back", "B6 pass B: g_w, softmax adjoint, stores", "end"};
  double sum[8] = {0}, mx[8] = {0};
  for (int w = 0; w < waves; ++w) {
    const unsigned long long* c = &clk[(size_t)w * 16];
    for (int k = 0; k < 7; ++k) { const double d = (double)(c[k + 1] - c[k]); sum[k] += d; if (d > mx[k]) mx[k] = d; }
    sum[7] += (double)(c[7] - c[0]);
  }
  printf("backward fit (g_F only), %d pairs x %d: %.2f us per launch (with stamps)\n", B, N, ms * 1e3 / 20);
  printf("%-64s %10s %10s\n", "phase (shader-clock cycles)", "mean", "max");
  for (int k = 0; k < 7; ++k) printf("%-64s %10.1f %10.1f  (%4.1f %%)\n", names[k], sum[k] / waves, mx[k], 100.0 * sum[k] / sum[7]);
  printf("%-64s %10.1f\n", "B0 .. end", sum[7] / waves);
  return 0;
}
