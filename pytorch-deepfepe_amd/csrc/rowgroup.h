// Row-group primitives for gfx950: a "row" is one of the four 16-lane DPP rows of a wavefront.
//
// The small-N solver kernels (w8pt16_*) give every image pair ONE ROW instead of one wavefront: four pairs share a
// wavefront, per-pair uniform arithmetic is replicated 16x instead of 64x, and everything a pair's lanes exchange goes
// through DPP inside the row (quad_perm / row_mirror / row_half_mirror for butterflies, row_newbcast for broadcasts) --
// full-rate VALU modifiers, no LDS round trip, no SGPR broadcast (which could not differ between the four pairs).
// Rows of one wavefront may diverge (different trip counts): DPP never crosses a row, so an exec-masked row is invisible
// to the others.  All lanes of a row must reach every rg_* call together.
//
// The kernel bodies (w8pt16_body.h) are written against this interface only; tests/emu/rowgroup.h implements the same
// interface on the host (16 fibres per pair) so that the bodies themselves are checked against the oracle without a GPU.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float hw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ __forceinline__ int rg_lane() { return (int)(threadIdx.x & 15u); }

template <int CTRL>
__device__ __forceinline__ int rg_dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float rg_dpp_f32(float v) {
  return __int_as_float(rg_dpp_i32<CTRL>(__float_as_int(v)));
}
template <int CTRL>
__device__ __forceinline__ double rg_dpp_f64(double v) {
  union { double d; int i[2]; } a, r;
  a.d = v;
  r.i[0] = rg_dpp_i32<CTRL>(a.i[0]);
  r.i[1] = rg_dpp_i32<CTRL>(a.i[1]);
  return r.d;
}

// value of lane K of the row, in every lane of the row (row_newbcast:K; one v_mov_b64_dpp for a double)
template <int K>
__device__ __forceinline__ double rg_bcast(double v) {
  static_assert(K >= 0 && K < 16, "lane of a 16-lane row");
  long long x = __builtin_bit_cast(long long, v);
  long long r = __builtin_amdgcn_update_dpp(x, x, 0x150 + K, 0xf, 0xf, true);
  return __builtin_bit_cast(double, r);
}
template <int K>
__device__ __forceinline__ float rg_bcast(float v) { return rg_dpp_f32<0x150 + K>(v); }
template <int K>
__device__ __forceinline__ int rg_bcast(int v) { return rg_dpp_i32<0x150 + K>(v); }

// acc + (lane K's x) * y
template <int K>
__device__ __forceinline__ double rg_fma_bcast(double acc, double x, double y) { return fma(rg_bcast<K>(x), y, acc); }

// sums / maxima over the 16 lanes of the row, result in every lane: xor-1, xor-2 (quad_perm), row_half_mirror, row_mirror
__device__ __forceinline__ double rg_sum(double v) {
  v += rg_dpp_f64<0xB1>(v);
  v += rg_dpp_f64<0x4E>(v);
  v += rg_dpp_f64<0x141>(v);
  v += rg_dpp_f64<0x140>(v);
  return v;
}
__device__ __forceinline__ float rg_sum(float v) {
  v += rg_dpp_f32<0xB1>(v);
  v += rg_dpp_f32<0x4E>(v);
  v += rg_dpp_f32<0x141>(v);
  v += rg_dpp_f32<0x140>(v);
  return v;
}
__device__ __forceinline__ int rg_sum(int v) {
  v += rg_dpp_i32<0xB1>(v);
  v += rg_dpp_i32<0x4E>(v);
  v += rg_dpp_i32<0x141>(v);
  v += rg_dpp_i32<0x140>(v);
  return v;
}
// number of lanes of the row for which p holds: one ballot + a popcount of the row's 16 bits (rows of the wavefront that have
// left a loop are masked off and contribute zeros to their own bits only)
__device__ __forceinline__ int rg_count(bool p) {
  const unsigned long long b = __ballot(p);
  return __popc((unsigned)(b >> (threadIdx.x & 48u)) & 0xffffu);
}
__device__ __forceinline__ float rg_max(float v) {
  v = fmaxf(v, rg_dpp_f32<0xB1>(v));
  v = fmaxf(v, rg_dpp_f32<0x4E>(v));
  v = fmaxf(v, rg_dpp_f32<0x141>(v));
  v = fmaxf(v, rg_dpp_f32<0x140>(v));
  return v;
}

// sum of lanes K0..K1 only (e.g. the lanes that hold the live part of a 9-vector), result in every lane: one
// broadcast-add per lane, cheaper than the 4-step butterfly (12 instructions for a double) for short ranges and no
// masking of the other lanes needed
template <int K0, int K1>
__device__ __forceinline__ double rg_sum_range(double v) {
  if constexpr (K0 >= K1) return rg_bcast<K1>(v);
  else return rg_bcast<K0>(v) + rg_sum_range<K0 + 1, K1>(v);
}

// partner exchange of the reduce-scatter: STEP 8 pairs lane k with 15-k (row_mirror), 4 with k^7 (row_half_mirror),
// 2 with k^2, 1 with k^1 (quad_perm): the partner always sits on the other side of bit STEP and on the same side of
// every higher bit.
template <int STEP>
__device__ __forceinline__ double rg_xchg(double v) {
  static_assert(STEP == 8 || STEP == 4 || STEP == 2 || STEP == 1, "reduce-scatter step");
  constexpr int ctrl = (STEP == 8) ? 0x140 : (STEP == 4) ? 0x141 : (STEP == 2) ? 0x4E : 0xB1;
  return rg_dpp_f64<ctrl>(v);
}

// Forces `x` to be materialised in a vector register at this point of the program: a scheduling fence for one value.  Used to
// make the consumers of early global loads run BEFORE a block that issues stores -- the memory counter is in-order, so a
// consumer scheduled after the stores would wait for the stores to complete, not just for its load.
__device__ __forceinline__ void rg_pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void rg_pin(double& x) { asm volatile("" : "+v"(x)); }

// Orders a row's LDS writes before its later LDS reads.  The lanes of a row belong to one wavefront, whose LDS operations
// execute in issue order: only the compiler has to be stopped from moving memory operations across.
__device__ __forceinline__ void rg_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
