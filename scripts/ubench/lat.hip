// Micro-benchmarks of single-wavefront issue behaviour on gfx950 (one wave per SIMD is the regime of the row-per-pair
// kernels at B = 4096): dependent vs independent chains of fp64/fp32 FMAs, DPP moves, fp64 transcendentals.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/lat.hip -o gpurun_out/lat && gpurun_out/lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, double seed, long long* cyc) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
  const double m = 0.999999, c = 1e-9;
  double b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  float g0 = 0, g1 = 0, g2 = 0, g3 = 0;
  long long t0 = __builtin_amdgcn_s_memrealtime();
  long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 4; ++r) {
      if (MODE == 0) { a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); }          // dependent fp64 fma
      if (MODE == 1) { a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c); }          // 4 independent chains
      if (MODE == 2) { f0 = fmaf(f0, 0.999f, 1e-9f); f0 = fmaf(f0, 0.999f, 1e-9f); f0 = fmaf(f0, 0.999f, 1e-9f); f0 = fmaf(f0, 0.999f, 1e-9f); }
      if (MODE == 3) { f0 = fmaf(f0, 0.999f, 1e-9f); f1 = fmaf(f1, 0.999f, 1e-9f); f2 = fmaf(f2, 0.999f, 1e-9f); f3 = fmaf(f3, 0.999f, 1e-9f); }
      if (MODE == 4) {  // dependent: bcast (v_mov_b64_dpp) + fma
        for (int q = 0; q < 4; ++q) {
          long long x = __builtin_bit_cast(long long, a0);
          long long y = __builtin_amdgcn_update_dpp(x, x, 0x153, 0xf, 0xf, true);
          a0 = fma(__builtin_bit_cast(double, y), m, c);
        }
      }
      if (MODE == 5) { a0 = __builtin_amdgcn_rcp(a0); a0 = __builtin_amdgcn_rcp(a0); a0 = __builtin_amdgcn_rcp(a0); a0 = __builtin_amdgcn_rcp(a0); }  // dependent v_rcp_f64
      if (MODE == 6) { a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3); }
      if (MODE == 7) { f0 = __builtin_amdgcn_rcpf(f0); f1 = __builtin_amdgcn_rcpf(f1); f2 = __builtin_amdgcn_rcpf(f2); f3 = __builtin_amdgcn_rcpf(f3); }
      if (MODE == 8) { f0 = __builtin_amdgcn_rcpf(f0); f0 = __builtin_amdgcn_rcpf(f0); f0 = __builtin_amdgcn_rcpf(f0); f0 = __builtin_amdgcn_rcpf(f0); }
      if (MODE == 9) { a0 = __builtin_amdgcn_rsq(a0); a1 = __builtin_amdgcn_rsq(a1); a2 = __builtin_amdgcn_rsq(a2); a3 = __builtin_amdgcn_rsq(a3); }
      if (MODE == 10) {  // fused v_fmac_f64_dpp row_newbcast, dependent through the accumulator only
        for (int q = 0; q < 4; ++q) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(a1), "v"(a2));
      }
      if (MODE == 11) {  // fused dpp, DPP source written by the previous instruction (hazard case) with explicit nop
        for (int q = 0; q < 4; ++q) asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(a0), "v"(a2));
      }
      if (MODE == 12) { a0 = a0 * m; a0 = a0 + c; a0 = a0 * m; a0 = a0 + c; }  // dependent mul / add f64
      if (MODE == 13) {  // cvt chain: f64->f32->f64
        for (int q = 0; q < 2; ++q) { float t = (float)a0; a0 = (double)t; }
      }
      // ---- issue cost of single instructions, four independent destinations each (no dependence between them)
      if (MODE == 14) asm volatile("v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                                   "v_mov_b64_dpp %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                                   : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      if (MODE == 15) asm volatile("v_mov_b32_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                                   "v_mov_b32_dpp %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                                   : "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
      if (MODE == 16) asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7"
                                   : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
      if (MODE == 17) asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7"
                                   : "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      if (MODE == 18) asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %6, vcc\n v_cndmask_b32 %2, %6, %7, vcc\n v_cndmask_b32 %3, %7, %4, vcc"
                                   : "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3) : "vcc");
      if (MODE == 19) asm volatile("v_mul_f64 %0, %4, %5\n v_mul_f64 %1, %5, %6\n v_add_f64 %2, %6, %7\n v_add_f64 %3, %7, %4"
                                   : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      if (MODE == 20) asm volatile("v_accvgpr_write_b32 a0, %4\n v_accvgpr_write_b32 a1, %5\n v_accvgpr_read_b32 %0, a2\n v_accvgpr_read_b32 %1, a3\n"
                                   : "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3) : "a0", "a1", "a2", "a3");
      if (MODE == 21) asm volatile("v_mov_b64 %0, %4\n v_mov_b64 %1, %5\n v_mov_b64 %2, %6\n v_mov_b64 %3, %7"
                                   : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      if (MODE == 22) asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %5, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                                   "v_fmac_f64_dpp %2, %6, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %7, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                                   : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      if (MODE == 23) asm volatile("v_exp_f32 %0, %4\n v_exp_f32 %1, %5\n v_rsq_f32 %2, %6\n v_sqrt_f32 %3, %7"
                                   : "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
    }
  }
  long long c1 = __builtin_readcyclecounter();
  long long t1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 + b0 + b1 + b2 + b3 + g0 + g1 + g2 + g3;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}

template <int MODE>
void run(const char* name, int ops_per_rep4, int blocks) {
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * 64 * blocks);
  hipMalloc(&cyc, 16);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 1.0, cyc);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 1.0, cyc);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  const double n = 64.0 * (REP / 4) * ops_per_rep4;
  printf("%-44s blocks %5d: %7.2f us, %8lld shader cycles, %6.2f cycles/op, realtime ticks %lld (100 MHz) -> %.2f GHz\n", name, blocks, ms * 1e3, h[0],
         (double)h[0] / n, h[1], (double)h[0] / ((double)h[1] * 10.0) );
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {1, 1024, 4096}) {
    run<0>("fp64 fma dependent", 4, blocks);
    run<1>("fp64 fma 4 independent chains", 4, blocks);
    run<2>("fp32 fma dependent", 4, blocks);
    run<3>("fp32 fma 4 independent chains", 4, blocks);
    run<4>("fp64 bcast(dpp mov) + fma dependent", 8, blocks);
    run<12>("fp64 mul, add dependent", 4, blocks);
    run<13>("cvt f64->f32->f64 dependent", 4, blocks);
    run<5>("v_rcp_f64 dependent", 4, blocks);
    run<6>("v_rcp_f64 independent", 4, blocks);
    run<8>("v_rcp_f32 dependent", 4, blocks);
    run<7>("v_rcp_f32 independent", 4, blocks);
    run<9>("v_rsq_f64 independent", 4, blocks);
    run<10>("v_fmac_f64_dpp (acc chain)", 4, blocks);
    run<11>("s_nop 1 + v_fmac_f64_dpp (src = acc)", 4, blocks);
    run<14>("v_mov_b64_dpp x4 independent", 4, blocks);
    run<15>("v_mov_b32_dpp x4 independent", 4, blocks);
    run<16>("v_cvt_f64_f32 x4 independent", 4, blocks);
    run<17>("v_cvt_f32_f64 x4 independent", 4, blocks);
    run<18>("v_cndmask_b32 x4 independent", 4, blocks);
    run<19>("v_mul_f64 x2 + v_add_f64 x2 independent", 4, blocks);
    run<20>("v_accvgpr_write x2 + read x2", 4, blocks);
    run<21>("v_mov_b64 x4 independent", 4, blocks);
    run<22>("v_fmac_f64_dpp x4 independent accumulators", 4, blocks);
    run<23>("v_exp_f32 x2, v_rsq_f32, v_sqrt_f32", 4, blocks);
  }
  return 0;
}
