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
// fp64 seeds (v_rsq_f64 / v_rcp_f64): ~2^-23 relative, 16 cycles; dfepe_math.h refines them with Newton-Raphson steps
__device__ __forceinline__ double hw_rsq64(double x) { return __builtin_amdgcn_rsq(x); }
__device__ __forceinline__ double hw_rcp64(double x) { return __builtin_amdgcn_rcp(x); }

__device__ __forceinline__ int rg_lane() { return (int)(threadIdx.x & 15u); }

// Two fp32 values in a register pair, for arithmetic that treats two correspondences of a lane alike: v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 do both in ONE full-rate instruction (gfx950), each half rounded exactly like the scalar instruction -- results are
// bit-identical to the unpacked code, a kernel bound by the issue of a lone wavefront just issues half as many of these.
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2 pk_make(float a, float b) { pk2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ pk2 pk_splat(float a) { pk2 r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ pk2 pk_fma(pk2 a, pk2 b, pk2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ pk2 pk_mul(pk2 a, pk2 b) { return a * b; }
__device__ __forceinline__ pk2 pk_add(pk2 a, pk2 b) { return a + b; }
__device__ __forceinline__ float pk_lo(pk2 a) { return a.x; }
__device__ __forceinline__ float pk_hi(pk2 a) { return a.y; }

// value of the same lane of the NEIGHBOURING row (lane ^ 16) -- for kernels that give a pair two rows of one wavefront; a DPP
// operand cannot cross a row, so this goes through the LDS crossbar (ds_bpermute, no LDS memory)
__device__ __forceinline__ float rg_xrow(float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63u) ^ 16u) << 2), __float_as_int(v)));
}
__device__ __forceinline__ double rg_xrow(double v) {
  union { double d; int i[2]; } a, r;
  a.d = v;
  const int addr = (int)(((threadIdx.x & 63u) ^ 16u) << 2);
  r.i[0] = __builtin_amdgcn_ds_bpermute(addr, a.i[0]);
  r.i[1] = __builtin_amdgcn_ds_bpermute(addr, a.i[1]);
  return r.d;
}

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

// ---- fused broadcast-FMA chains over lanes J0..8 of the row ------------------------------------------------------------------
// `v_fmac_f64_dpp acc, x, y row_newbcast:j` is acc += (lane j's x) * y in ONE instruction; hipcc only ever emits the pair
// v_mov_b64_dpp + v_fma_f64 (the fp64 FMA it selects is VOP3, which has no DPP form on gfx950).  The Householder steps of the
// 9x9 eigenproblem are chains of exactly this operation over the lanes that hold the matrix rows, so the chains are written here
// as ONE asm statement each.  hipcc does not pad hazards inside or around an asm statement: a DPP operand must not have been
// written by the two preceding VALU instructions (which may be a register copy the compiler places right in front of the
// statement), hence the leading s_nop 1; the trailing one covers a compiler-generated DPP read of a result.
#define RG_SFX(j) " row_newbcast:" #j " row_mask:0xf bank_mask:0xf\n\t"
#define RG_DOT_E(j) "v_fmac_f64_dpp %[acc0], %[x], %[a" #j "]" RG_SFX(j)
#define RG_DOT_O(j) "v_fmac_f64_dpp %[acc1], %[x], %[a" #j "]" RG_SFX(j)
#define RG_AX1_L(j) "v_fmac_f64_dpp %[a" #j "], %[x1], %[y1]" RG_SFX(j)
#define RG_AX2_L(j) "v_fmac_f64_dpp %[a" #j "], %[x2], %[y2]" RG_SFX(j)
#define RG_SUM_E(j) "v_fmac_f64_dpp %[acc0], %[x], %[one]" RG_SFX(j)
#define RG_SUM_O(j) "v_fmac_f64_dpp %[acc1], %[x], %[one]" RG_SFX(j)
#define RG_MOV_E(j) "v_mov_b64_dpp %[acc0], %[x]" RG_SFX(j)
#define RG_MOV_O(j) "v_mov_b64_dpp %[acc1], %[x]" RG_SFX(j)
// M applied to lanes J..8 / A and B applied alternately to lanes J..8 (two accumulators: a dependent v_fmac_f64_dpp issues every
// 8 cycles, an independent one every 5 -- scripts/ubench/lat.hip)
#define RG_REP8(M) M(8)
#define RG_REP7(M) M(7) RG_REP8(M)
#define RG_REP6(M) M(6) RG_REP7(M)
#define RG_REP5(M) M(5) RG_REP6(M)
#define RG_REP4(M) M(4) RG_REP5(M)
#define RG_REP3(M) M(3) RG_REP4(M)
#define RG_REP2(M) M(2) RG_REP3(M)
#define RG_REP1(M) M(1) RG_REP2(M)
#define RG_ALT9(A, B)
#define RG_ALT8(A, B) A(8)
#define RG_ALT7(A, B) A(7) RG_ALT8(B, A)
#define RG_ALT6(A, B) A(6) RG_ALT7(B, A)
#define RG_ALT5(A, B) A(5) RG_ALT6(B, A)
#define RG_ALT4(A, B) A(4) RG_ALT5(B, A)
#define RG_ALT3(A, B) A(3) RG_ALT4(B, A)
#define RG_ALT2(A, B) A(2) RG_ALT3(B, A)
#define RG_ALT1(A, B) A(1) RG_ALT2(B, A)
// the same lists without lane 8 (operand lists name a[8] first and the rest behind commas)
#define RG_REQ8(M)
#define RG_REQ7(M) M(7)
#define RG_REQ6(M) M(6) RG_REQ7(M)
#define RG_REQ5(M) M(5) RG_REQ6(M)
#define RG_REQ4(M) M(4) RG_REQ5(M)
#define RG_REQ3(M) M(3) RG_REQ4(M)
#define RG_REQ2(M) M(2) RG_REQ3(M)
#define RG_REQ1(M) M(1) RG_REQ2(M)
#define RG_IN_A(j) , [a##j] "v"(a[j])
#define RG_IO_A(j) , [a##j] "+&v"(a[j])
#define RG_EACH_J0(C) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8)

// sum_{j = J0..8} (lane j's x) * a[j]  (terms J0, J0+2, .. and J0+1, J0+3, .. summed separately, then added)
template <int J0>
__device__ __forceinline__ double rg_dot_bcast(double x, const double* a) {
  static_assert(J0 >= 1 && J0 <= 8, "lanes J0..8 of the row");
  double acc0 = 0.0, acc1 = 0.0;
#define RG_CASE(J) \
  if constexpr (J0 == J) asm volatile("s_nop 1\n\t" RG_ALT##J(RG_DOT_E, RG_DOT_O) : [acc0] "+&v"(acc0), [acc1] "+&v"(acc1) : [x] "v"(x) RG_REP##J(RG_IN_A));
  RG_EACH_J0(RG_CASE)
#undef RG_CASE
  return (J0 == 8) ? acc0 : acc0 + acc1;
}
// a[j] += (lane j's x1) * y1, j = J0..8
template <int J0>
__device__ __forceinline__ void rg_axpy_bcast(double* a, double x1, double y1) {
  static_assert(J0 >= 1 && J0 <= 8, "lanes J0..8 of the row");
#define RG_CASE(J) \
  if constexpr (J0 == J) asm volatile("s_nop 1\n\t" RG_REP##J(RG_AX1_L) "s_nop 1" : [a8] "+&v"(a[8]) RG_REQ##J(RG_IO_A) : [x1] "v"(x1), [y1] "v"(y1));
  RG_EACH_J0(RG_CASE)
#undef RG_CASE
}
// a[j] += (lane j's x1) * y1 + (lane j's x2) * y2, j = J0..8 (in this order, like two rg_fma_bcast)
template <int J0>
__device__ __forceinline__ void rg_axpy2_bcast(double* a, double x1, double y1, double x2, double y2) {
  static_assert(J0 >= 1 && J0 <= 8, "lanes J0..8 of the row");
#define RG_CASE(J) \
  if constexpr (J0 == J) asm volatile("s_nop 1\n\t" RG_REP##J(RG_AX1_L) RG_REP##J(RG_AX2_L) "s_nop 1" : [a8] "+&v"(a[8]) RG_REQ##J(RG_IO_A) : [x1] "v"(x1), [y1] "v"(y1), [x2] "v"(x2), [y2] "v"(y2));
  RG_EACH_J0(RG_CASE)
#undef RG_CASE
}
// sum of lanes J0..8 of the row: two broadcast moves, then fused adds (x * 1.0) into the two partial sums (lanes J0, J0+2, ..
// and J0+1, J0+3, ..).  The results of rg_dot_bcast / rg_sum_to8 feed ordinary arithmetic, so these statements carry no trailing
// s_nop; the axpy forms update the matrix rows that the next step broadcasts, so they do.
template <int J0>
__device__ __forceinline__ double rg_sum_to8(double x) {
  static_assert(J0 >= 0 && J0 <= 8, "lanes J0..8 of the row");
  if constexpr (J0 == 8) return rg_bcast<8>(x);
  if constexpr (J0 == 7) return rg_bcast<7>(x) + rg_bcast<8>(x);
  double acc0, acc1;
  const double one = 1.0;
#define RG_CASE(J0V, J1V, J2V) \
  if constexpr (J0 == J0V) asm volatile("s_nop 1\n\t" RG_MOV_E(J0V) RG_MOV_O(J1V) RG_ALT##J2V(RG_SUM_E, RG_SUM_O) : [acc0] "=&v"(acc0), [acc1] "=&v"(acc1) : [x] "v"(x), [one] "v"(one));
  RG_CASE(0, 1, 2) RG_CASE(1, 2, 3) RG_CASE(2, 3, 4) RG_CASE(3, 4, 5) RG_CASE(4, 5, 6) RG_CASE(5, 6, 7) RG_CASE(6, 7, 8)
#undef RG_CASE
  return acc0 + acc1;
}

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
  const unsigned long long b = __builtin_amdgcn_ballot_w64(p);  // the lane mask of p itself: no 0/1 value is materialised
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
