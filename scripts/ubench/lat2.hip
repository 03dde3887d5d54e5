// Second set of single-wavefront issue-cost probes on gfx950 (one wavefront per SIMD): selects on an SGPR mask, the
// compare -> mask -> select round trip through the scalar file, exec-mask regions, global stores with one or all lanes active,
// LDS stores / wide loads.  Companion of lat.hip; same harness.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/lat2.hip -o ab_libs/lat2 && ab_libs/lat2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP 64
template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, float* sink, double seed, long long* cyc, unsigned long long mask) {
  __shared__ double lds[64 * 40];
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
  float g0 = 0, g1 = 0, g2 = 0, g3 = 0;
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f q4 = {f0, f1, f2, f3};
  float* mine = sink + (size_t)blockIdx.x * 4096 + threadIdx.x * 4;
  double* lp = lds + threadIdx.x;
  long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 4; ++r) {
      if (MODE == 0) asm volatile("v_cndmask_b32_e64 %0, %4, %5, %8\n v_cndmask_b32_e64 %1, %5, %6, %8\n v_cndmask_b32_e64 %2, %6, %7, %8\n v_cndmask_b32_e64 %3, %7, %4, %8"
                                  : "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "s"(mask));
      if (MODE == 1) asm volatile("v_cmp_gt_f64 vcc, %4, %5\n v_cndmask_b32 %0, %6, %7, vcc\n v_cmp_gt_f64 vcc, %5, %4\n v_cndmask_b32 %1, %7, %6, vcc"
                                  : "=v"(g0), "=v"(g1) : "v"(g2), "v"(g3), "v"(a0), "v"(a1), "v"(f0), "v"(f1) : "vcc");
      if (MODE == 2) asm volatile("v_cmp_gt_f64_e64 s[20:21], %2, %3\n s_and_b64 s[20:21], s[20:21], %6\n v_cndmask_b32_e64 %0, %4, %5, s[20:21]\n"
                                  "v_cmp_gt_f64_e64 s[22:23], %3, %2\n s_and_b64 s[22:23], s[22:23], %6\n v_cndmask_b32_e64 %1, %5, %4, s[22:23]"
                                  : "=v"(g0), "=v"(g1) : "v"(a0), "v"(a1), "v"(f0), "v"(f1), "s"(mask) : "s20", "s21", "s22", "s23");
      if (MODE == 3) {  // dependent: compare on the previous select's result
        asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %1, %2, vcc\n v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %2, %1, vcc"
                     : "+v"(g0) : "v"(f0), "v"(f1) : "vcc");
      }
      if (MODE == 4) {  // exec-mask region: saveexec, two VALU, restore (x2)
        asm volatile("s_and_saveexec_b64 s[20:21], %2\n v_add_f32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]\n"
                     "s_and_saveexec_b64 s[20:21], %2\n v_add_f32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]"
                     : "+v"(g0) : "v"(f1), "s"(mask) : "s20", "s21");
      }
      if (MODE == 5) {  // global_store_dword, all lanes (x2)
        asm volatile("global_store_dword %0, %1, off\n global_store_dword %0, %2, off offset:2048" : : "v"(mine), "v"(f0), "v"(f1) : "memory");
      }
      if (MODE == 6) {  // global_store_dwordx4, all lanes (x2)
        asm volatile("global_store_dwordx4 %0, %1, off\n global_store_dwordx4 %0, %1, off offset:2048" : : "v"(mine), "v"(q4) : "memory");
      }
      if (MODE == 7) {  // global_store_dwordx4 with ONE lane active (x2)
        asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 1\n global_store_dwordx4 %0, %1, off\n global_store_dwordx4 %0, %1, off offset:2048\n s_mov_b64 exec, s[20:21]"
                     : : "v"(mine), "v"(q4) : "memory", "s20", "s21");
      }
      if (MODE == 8) {  // ds_write_b64 x2
        asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %2 offset:512" : : "v"((unsigned)(size_t)lp), "v"(a0), "v"(a1) : "memory");
      }
      if (MODE == 9) {  // ds_read_b128 x2 + wait
        v4f q0, q1;
        asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1) : "v"((unsigned)(threadIdx.x * 16)) : "memory");
        g0 += q0.x + q1.y;
      }
      if (MODE == 10) {  // v_readfirstlane + s_bcnt + v_mov (scalar round trip) x2
        asm volatile("v_readfirstlane_b32 s20, %1\n s_bcnt1_i32_b32 s20, s20\n v_mov_b32 %0, s20\n v_readfirstlane_b32 s21, %0\n s_bcnt1_i32_b32 s21, s21\n v_mov_b32 %0, s21"
                     : "+v"(g0) : "v"(f0) : "s20", "s21", "scc");
      }
    }
  }
  long long c1 = __builtin_readcyclecounter();
  __syncthreads();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 + g0 + g1 + g2 + g3 + lds[threadIdx.x];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}

template <int MODE>
void run(const char* name, int ops_per_rep4, int blocks) {
  double* out; long long* cyc; float* sink;
  hipMalloc(&out, sizeof(double) * 64 * blocks);
  hipMalloc(&cyc, 16);
  hipMalloc(&sink, (size_t)blocks * 4096 * 4 + 65536);
  for (int w = 0; w < 2; ++w) {
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, sink, 1.0, cyc, 0x5555aaaa3333ccccull);
    hipDeviceSynchronize();
  }
  long long h[2]; hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = 64.0 * (REP / 4) * ops_per_rep4;
  printf("%-64s blocks %5d: %9lld cycles, %6.2f cycles/op\n", name, blocks, h[0], (double)h[0] / n);
  hipFree(out); hipFree(cyc); hipFree(sink);
}

int main(int argc, char** argv) {
  const int only = (argc > 1) ? atoi(argv[1]) : -1;  // one probe per process: a probe that wedges does not take the others with it
  setvbuf(stdout, nullptr, _IONBF, 0);
  for (int blocks : {1, 1024}) {
    if (only < 0 || only == 0) run<0>("v_cndmask_b32_e64 on an SGPR mask, x4 independent", 4, blocks);
    if (only < 0 || only == 1) run<1>("v_cmp_gt_f64 vcc + v_cndmask vcc (pair), x2", 4, blocks);
    if (only < 0 || only == 2) run<2>("v_cmp_gt_f64 -> s_and_b64 -> v_cndmask (triple), x2", 6, blocks);
    if (only < 0 || only == 3) run<3>("v_cmp_gt_f32 vcc + v_cndmask, dependent chain, x2", 4, blocks);
    if (only < 0 || only == 4) run<4>("s_and_saveexec + v_add + s_or exec (triple), x2", 6, blocks);
    if (only < 0 || only == 5) run<5>("global_store_dword all lanes, x2 (per store)", 2, blocks);
    if (only < 0 || only == 6) run<6>("global_store_dwordx4 all lanes, x2 (per store)", 2, blocks);
    if (only < 0 || only == 7) run<7>("global_store_dwordx4 one lane active, x2 (per store)", 2, blocks);
    if (only < 0 || only == 8) run<8>("ds_write_b64 x2 (per write)", 2, blocks);
    if (only < 0 || only == 9) run<9>("ds_read_b128 x2 + wait (per read)", 2, blocks);
    if (only < 0 || only == 10) run<10>("v_readfirstlane + s_bcnt1 + v_mov (triple), x2 dependent", 6, blocks);
  }
  return 0;
}
