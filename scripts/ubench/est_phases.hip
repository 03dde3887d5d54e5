// Where the estimator's GEMM kernels spend their cycles: csrc/est_gemm.hip compiled with -DDFEPE_EST_PHASE_CLOCKS (lane 0 of every
// wavefront stamps the shader clock at the phase boundaries of est_gemm_nt_kernel and sums, per K step, the cycles from issuing the
// LDS-DMA to the barrier behind it and the cycles of the fragment reads + MFMAs), run at B = 4096 pairs x 100 points on random planes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -DDFEPE_EST_PHASE_CLOCKS -Iinclude \
//         -Ipytorch-deepfepe_amd/csrc scripts/ubench/est_phases.hip -o ab_libs/est_phases && ab_libs/est_phases
#include "../../pytorch-deepfepe_amd/csrc/est_gemm.hip"
#include <cstdio>
#include <random>
#include <vector>

static void fill16(void* d, size_t n, int fmt, float scale) {  // random fp16 / bf16 bit patterns of magnitude ~scale
  std::vector<unsigned short> h(n);
  std::mt19937 g(7);
  std::normal_distribution<float> nd(0.f, scale);
  for (size_t i = 0; i < n; ++i) {
    const float v = nd(g);
    if (fmt == FMT_F16) { _Float16 x = (_Float16)v; h[i] = __builtin_bit_cast(unsigned short, x); }
    else { unsigned u = __builtin_bit_cast(unsigned, v); h[i] = (unsigned short)(u >> 16); }
  }
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

static void report(const char* what, unsigned long long* dclk, int wgs, float us) {
  std::vector<unsigned long long> c((size_t)wgs * 4 * 8);
  hipMemcpy(c.data(), dclk, c.size() * 8, hipMemcpyDeviceToHost);
  double loop = 0, epiA = 0, epiR = 0, epiB = 0, wait = 0, mfma = 0, total = 0;
  const int waves = wgs * 4;
  for (int w = 0; w < waves; ++w) {
    const unsigned long long* q = &c[(size_t)w * 8];
    loop += (double)(q[1] - q[0]); epiA += (double)(q[2] - q[1]);
    if (q[3]) { epiR += (double)(q[3] - q[2]); epiB += (double)(q[4] - q[3]); } else epiB += (double)(q[4] - q[2]);
    wait += (double)q[5]; mfma += (double)q[6]; total += (double)(q[4] - q[0]);
  }
  printf("%s: %.1f us per launch; per wavefront (100 MHz-independent shader-clock cycles, mean over %d wavefronts):\n", what, us, waves);
  printf("   K loop %.0f  (of it: DMA issue -> barrier %.0f, fragment reads + MFMAs %.0f)\n", loop / waves, wait / waves, mfma / waves);
  printf("   epilogue: first pass / statistics %.0f, row sums + constants %.0f, second pass / stores %.0f;  whole wavefront %.0f\n",
         epiA / waves, epiR / waves, epiB / waves, total / waves);
}

int main() {
  const int pairs = 4096, cols = pairs * 100;
  unsigned long long* dclk;
  const int max_wgs = (cols / 200) * 8;
  hipMalloc(&dclk, (size_t)max_wgs * 4 * 8 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_est_phase_clk), &dclk, sizeof(dclk));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Shape { const char* name; int M, K; bool fused; };
  const Shape shapes[] = {{"forward 1024 -> 512 (fp16, fused InstanceNorm epilogue, four planes out)", 512, 1024, false},
                          {"forward 128 -> 1024", 1024, 128, false},
                          {"data gradient 512 -> 1024 + adjoint of the 1024-channel layer", 1024, 512, true},
                          {"data gradient 256 -> 512 + adjoint of the 512-channel layer", 512, 256, true}};
  for (const Shape& S : shapes) {
    const int M = S.M, K = S.K;
    void *A, *Bp, *out, *outb, *aout;
    float *gamma, *beta, *rstd, *dg, *db;
    hipMalloc(&A, (size_t)2 * M * K * 2); hipMalloc(&Bp, (size_t)2 * cols * K * 2);
    hipMalloc(&out, (size_t)2 * cols * M * 2); hipMalloc(&outb, (size_t)2 * cols * M * 2); hipMalloc(&aout, (size_t)2 * cols * M * 2);
    hipMalloc(&gamma, M * 4); hipMalloc(&beta, M * 4); hipMalloc(&rstd, (size_t)pairs * M * 4); hipMalloc(&dg, (size_t)pairs * M * 4); hipMalloc(&db, (size_t)pairs * M * 4);
    const int fmt = S.fused ? FMT_BF16 : FMT_F16;
    fill16(A, (size_t)2 * M * K, fmt, 0.05f); fill16(Bp, (size_t)2 * cols * K, fmt, 1.0f); fill16(aout, (size_t)2 * cols * M, FMT_BF16, 1.0f);
    std::vector<float> ones(M, 1.0f), zeros(M, 0.1f), rs((size_t)pairs * M, 1.3f);
    hipMemcpy(gamma, ones.data(), M * 4, hipMemcpyHostToDevice); hipMemcpy(beta, zeros.data(), M * 4, hipMemcpyHostToDevice);
    hipMemcpy(rstd, rs.data(), rs.size() * 4, hipMemcpyHostToDevice);
    const int wgs = ((cols + 199) / 200) * ((M + 127) / 128);
    auto run = [&]() {
      if (S.fused) return dfepe_est_dgrad_in_bwd(A, (size_t)M * K, Bp, (size_t)cols * K, M, cols, K, aout, (size_t)cols * M, rstd, gamma, beta, 0.01f, out,
                                                 (size_t)cols * M, dg, db, nullptr);
      return dfepe_est_layer_fwd(A, (size_t)M * K, Bp, (size_t)cols * K, M, cols, K, nullptr, gamma, beta, 1e-5f, 0.01f, out, (size_t)cols * M, outb,
                                 (size_t)cols * M, rstd, nullptr);
    };
    for (int r = 0; r < 2; ++r) if (run() != 0) { printf("launch failed\n"); return 1; }
    hipMemset(dclk, 0, (size_t)max_wgs * 4 * 8 * 8);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) run();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    report(S.name, dclk, wgs, ms * 1e3f / 5);
    hipFree(A); hipFree(Bp); hipFree(out); hipFree(outb); hipFree(aout); hipFree(gamma); hipFree(beta); hipFree(rstd); hipFree(dg); hipFree(db);
  }
  return 0;
}
