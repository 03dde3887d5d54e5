// HOST EMULATION of pytorch-deepfepe_amd/csrc/rowgroup.h -- TEST INFRASTRUCTURE ONLY, never part of the product.
//
// The row-per-pair kernel bodies (csrc/w8pt16_body.h, csrc/w8pt16_bwd_body.h) are written against the row-group interface
// alone.  This header implements that interface on the CPU: the 16 lanes of a row are 16 cooperatively scheduled fibres
// (ucontext) of one thread; every rg_* exchange publishes the lane's value, yields once around the ring (so that all 16
// lanes have published) and reads its partner's.  Values alternate between two buffers, so a lane that runs ahead to its
// next exchange cannot overwrite what a slower lane has not read yet.  tests/emu/emu_w8pt16.cpp compiles the SAME body
// headers with g++ against this file (include path order picks this rowgroup.h), and tests/test_emu_cpu.py compares the
// result with the oracle: the arithmetic of the HIP kernels is checked without a GPU.  The DPP encodings themselves are
// the only thing this cannot see; tests/test_rowgroup_gpu.py covers those on the GPU box.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <ucontext.h>

#include <type_traits>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
// the cooperative (16 rows per pair) variant of the bodies is device-only: the emulation runs one row per pair, where the
// block-level barrier is never reached
#define DFEPE_BLOCK_SYNC() abort()
// likewise the two-rows-of-a-wavefront variant (ROWS = 2): never instantiated here
inline float rg_xrow(float) { abort(); }
inline double rg_xrow(double) { abort(); }

struct float4 {
  float x, y, z, w;
};

// two fp32 values handled alike (the device packs them into one v_pk_* instruction; each half rounds like the scalar operation)
struct pk2 { float x, y; };
inline pk2 pk_make(float a, float b) { return pk2{a, b}; }
inline pk2 pk_splat(float a) { return pk2{a, a}; }
inline pk2 pk_fma(pk2 a, pk2 b, pk2 c) { return pk2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
inline pk2 pk_mul(pk2 a, pk2 b) { return pk2{a.x * b.x, a.y * b.y}; }
inline pk2 pk_add(pk2 a, pk2 b) { return pk2{a.x + b.x, a.y + b.y}; }
inline float pk_lo(pk2 a) { return a.x; }
inline float pk_hi(pk2 a) { return a.y; }

inline float hw_rsq(float x) { return 1.0f / sqrtf(x); }
inline float hw_rcp(float x) { return 1.0f / x; }
inline float hw_sqrt(float x) { return sqrtf(x); }
// the fp64 hardware seeds are good to ~2^-23: the emulation keeps 24 bits of the exact result, so that the Newton step counts of
// dfepe_math.h are exercised with a seed no better than the hardware's
inline double emu_seed24(double v) {
  uint64_t u;
  memcpy(&u, &v, 8);
  u &= ~((1ull << 29) - 1ull);
  memcpy(&v, &u, 8);
  return v;
}
inline double hw_rsq64(double x) { return emu_seed24(1.0 / sqrt(x)); }
inline double hw_rcp64(double x) { return emu_seed24(1.0 / x); }

namespace emu {
constexpr int kLanes = 16;
struct Row {
  ucontext_t sched;
  ucontext_t fib[kLanes];
  bool done[kLanes];
  int cur;
  uint64_t buf[2][kLanes];
  unsigned op[kLanes];  // exchanges this lane has entered
};
extern thread_local Row* g_row;

// hand the thread to the scheduler; returns when every other live lane has run up to its next yield
inline void yield() {
  Row* r = g_row;
  swapcontext(&r->fib[r->cur], &r->sched);
}

template <class T>
inline uint64_t pack(T v) {
  static_assert(sizeof(T) <= 8, "exchange payload");
  uint64_t u = 0;
  memcpy(&u, &v, sizeof(T));
  return u;
}
template <class T>
inline T unpack(uint64_t u) {
  T v;
  memcpy(&v, &u, sizeof(T));
  return v;
}

// publish v, wait for the row, return the row's values (valid until this lane's next-but-one exchange)
template <class T>
inline const uint64_t* exchange(T v) {
  Row* r = g_row;
  const int me = r->cur;
  const unsigned slot = r->op[me]++ & 1u;
  r->buf[slot][me] = pack(v);
  yield();
  return r->buf[slot];
}
}  // namespace emu

inline int rg_lane() { return emu::g_row->cur; }

template <int K, class T>
inline T rg_bcast(T v) {
  static_assert(K >= 0 && K < 16, "lane of a 16-lane row");
  return emu::unpack<T>(emu::exchange(v)[K]);
}
template <int K>
inline double rg_fma_bcast(double acc, double x, double y) { return fma(rg_bcast<K>(x), y, acc); }

// same association order as the DPP butterflies (xor 1, xor 2, k^7, 15-k), so sums round identically
template <class T, class Op>
inline T rg_butterfly(T v, Op op) {
  const int l = rg_lane();
  v = op(v, emu::unpack<T>(emu::exchange(v)[l ^ 1]));
  v = op(v, emu::unpack<T>(emu::exchange(v)[l ^ 2]));
  v = op(v, emu::unpack<T>(emu::exchange(v)[l ^ 7]));
  v = op(v, emu::unpack<T>(emu::exchange(v)[15 - l]));
  return v;
}
inline double rg_sum(double v) { return rg_butterfly(v, [](double a, double b) { return a + b; }); }
inline float rg_sum(float v) { return rg_butterfly(v, [](float a, float b) { return a + b; }); }
inline int rg_sum(int v) { return rg_butterfly(v, [](int a, int b) { return a + b; }); }
inline int rg_count(bool p) { return rg_sum(p ? 1 : 0); }  // lanes of the row for which p holds
inline float rg_max(float v) { return rg_butterfly(v, [](float a, float b) { return fmaxf(a, b); }); }

template <int K0, int K1>
inline double rg_sum_range(double v) {
  if constexpr (K0 >= K1) return rg_bcast<K1>(v);
  else return rg_bcast<K0>(v) + rg_sum_range<K0 + 1, K1>(v);
}

// the fused chains of csrc/rowgroup.h, same order of operations (two partial sums over alternating lanes)
template <int J0>
inline double rg_dot_bcast(double x, const double* a) {
  double acc[2] = {0.0, 0.0};
  const double b[9] = {0.0, rg_bcast<1>(x), rg_bcast<2>(x), rg_bcast<3>(x), rg_bcast<4>(x), rg_bcast<5>(x), rg_bcast<6>(x), rg_bcast<7>(x), rg_bcast<8>(x)};
  for (int j = J0; j <= 8; ++j) acc[(j - J0) & 1] = fma(b[j], a[j], acc[(j - J0) & 1]);
  return (J0 == 8) ? acc[0] : acc[0] + acc[1];
}
template <int J0>
inline void rg_axpy_bcast(double* a, double x1, double y1) {
  const double b[9] = {0.0, rg_bcast<1>(x1), rg_bcast<2>(x1), rg_bcast<3>(x1), rg_bcast<4>(x1), rg_bcast<5>(x1), rg_bcast<6>(x1), rg_bcast<7>(x1), rg_bcast<8>(x1)};
  for (int j = J0; j <= 8; ++j) a[j] = fma(b[j], y1, a[j]);
}
template <int J0>
inline void rg_axpy2_bcast(double* a, double x1, double y1, double x2, double y2) {
  rg_axpy_bcast<J0>(a, x1, y1);
  rg_axpy_bcast<J0>(a, x2, y2);
}
template <int J0>
inline double rg_sum_to8(double x) {
  const double b[9] = {rg_bcast<0>(x), rg_bcast<1>(x), rg_bcast<2>(x), rg_bcast<3>(x), rg_bcast<4>(x), rg_bcast<5>(x), rg_bcast<6>(x), rg_bcast<7>(x), rg_bcast<8>(x)};
  if (J0 == 8) return b[8];
  double acc[2] = {b[J0], b[J0 + 1]};
  for (int j = J0 + 2; j <= 8; ++j) acc[(j - J0) & 1] += b[j];
  return acc[0] + acc[1];
}

template <int STEP>
inline double rg_xchg(double v) {
  static_assert(STEP == 8 || STEP == 4 || STEP == 2 || STEP == 1, "reduce-scatter step");
  const int l = rg_lane();
  const int partner = (STEP == 8) ? 15 - l : (STEP == 4) ? (l ^ 7) : (STEP == 2) ? (l ^ 2) : (l ^ 1);
  return emu::unpack<double>(emu::exchange(v)[partner]);
}

// all lanes' earlier shared-memory writes are visible after this
inline void rg_sync() { emu::yield(); }

// scheduling hint on the device, nothing to do here
inline void rg_pin(float&) {}
inline void rg_pin(double&) {}
