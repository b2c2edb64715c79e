// Multi-precision primitives on saturated 32-bit limbs for gfx950.
//
// Design notes (measured with tools/ubench/valu_rates.hip on MI355X):
//  * the native integer multiplier is v_mad_u64_u32 (32x32+64 -> 64, carry-out in an SGPR pair);
//    a u64 x u64 product lowers to four of them, so the natural limb is 32 bits and a field
//    element of a 256-bit (384-bit) prime lives in 8 (12) VGPRs of one lane;
//  * v_mad_u64_u32 has a carry-out but no carry-in and 64-bit VGPR operands must be even aligned,
//    so products are accumulated column-wise into a 96-bit accumulator: one mad plus one
//    v_addc_co_u32 per 32x32 multiply-accumulate.  hipcc does not form that pair from C++
//    (it emits mad + 64-bit add + compare + select), hence the two-instruction asm statement.
//
// Everything here is __host__ __device__: the same templates are compiled by the host compiler
// into tests/hosttwin (a test-only library) so that the arithmetic can be checked against the
// oracle in a container without a GPU.  The shipped library (libecgpu.so) only ever launches
// the device instantiation.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ECGPU_HD __host__ __device__ __forceinline__
#else
#define ECGPU_HD inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define ECGPU_ASM 1
#else
#define ECGPU_ASM 0
#endif

// Hook at every read of a precomputed-table entry; empty in the product.  The test-only host build (tests/hosttwin)
// defines it to record the entry index, which is how the tests show that the constant-time schedules touch the same
// entries in the same order whatever the scalar is, and that the throughput schedules do not.
#ifndef ECGPU_TABLE_TOUCH
#define ECGPU_TABLE_TOUCH(idx) ((void)0)
#endif

namespace ecgpu {

typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------------------
// carry chains.  __builtin_addc / __builtin_subc lower to v_add_co_u32 / v_addc_co_u32 chains.
// ---------------------------------------------------------------------------------------------
ECGPU_HD u32 addc(u32 a, u32 b, u32& carry) {
#if defined(__clang__)
  u32 co;
  u32 r = __builtin_addc(a, b, carry, &co);
  carry = co;
  return r;
#else
  u64 t = (u64)a + b + carry;
  carry = (u32)(t >> 32);
  return (u32)t;
#endif
}
ECGPU_HD u32 subb(u32 a, u32 b, u32& borrow) {
#if defined(__clang__)
  u32 bo;
  u32 r = __builtin_subc(a, b, borrow, &bo);
  borrow = bo;
  return r;
#else
  u64 t = (u64)a - b - borrow;
  borrow = (u32)(t >> 63);
  return (u32)t;
#endif
}

// r = a + b, returns carry-out
template <int N>
ECGPU_HD u32 mp_add(u32* r, const u32* a, const u32* b) {
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = addc(a[i], b[i], c);
  return c;
}
// r = a - b, returns borrow-out
template <int N>
ECGPU_HD u32 mp_sub(u32* r, const u32* a, const u32* b) {
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = subb(a[i], b[i], c);
  return c;
}
// a >= b ?
template <int N>
ECGPU_HD bool mp_geq(const u32* a, const u32* b) {
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) (void)subb(a[i], b[i], c);
  return c == 0;
}
template <int N>
ECGPU_HD bool mp_is_zero(const u32* a) {
  u32 t = 0;
#pragma unroll
  for (int i = 0; i < N; i++) t |= a[i];
  return t == 0;
}
template <int N>
ECGPU_HD bool mp_eq(const u32* a, const u32* b) {
  u32 t = 0;
#pragma unroll
  for (int i = 0; i < N; i++) t |= a[i] ^ b[i];
  return t == 0;
}
template <int N>
ECGPU_HD void mp_copy(u32* r, const u32* a) {
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = a[i];
}
template <int N>
ECGPU_HD void mp_zero(u32* r) {
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = 0;
}
// r = cond ? a : b   (lane-wise select, v_cndmask_b32)
template <int N>
ECGPU_HD void mp_select(u32* r, bool cond, const u32* a, const u32* b) {
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = cond ? a[i] : b[i];
}

// ---------------------------------------------------------------------------------------------
// 96-bit column accumulator
// ---------------------------------------------------------------------------------------------
struct Acc96 {
  u64 lo;
  u32 hi;
};

// c += a * b
ECGPU_HD void mac(Acc96& c, u32 a, u32 b) {
#if ECGPU_ASM
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(c.lo), "+v"(c.hi)
      : "v"(a), "v"(b)
      : "vcc");
#else
  u64 p = (u64)a * b;
  c.lo += p;
  c.hi += (c.lo < p);
#endif
}
// c += a * b where the caller guarantees the low 64 bits cannot overflow
ECGPU_HD void mac_nc(Acc96& c, u32 a, u32 b) { c.lo += (u64)a * b; }
// c -= w, the accumulator read as a 96-bit two's-complement number
ECGPU_HD void acc_sub32(Acc96& c, u32 w) {
  u32 l0 = (u32)c.lo, l1 = (u32)(c.lo >> 32), bw = 0;
  l0 = subb(l0, w, bw);
  l1 = subb(l1, 0u, bw);
  c.hi = subb(c.hi, 0u, bw);
  c.lo = ((u64)l1 << 32) | l0;
}

#include "mp32_cols.inc"

// c += sum_{m < M} pa[m] * pb[m], issued as one asm statement.  FRESH: the caller knows c.hi == 0 (the accumulator was
// just popped or zeroed); the first carry then produces c.hi instead of adding to it, which saves zeroing a register
// per column.
#ifndef ECGPU_MAC_FRESH
#define ECGPU_MAC_FRESH 1          // 0: always read c.hi (A/B switch for tools/nopbench)
#endif
template <int M, bool FRESH_ARG = false>
ECGPU_HD void mac_cols(Acc96& c, const u32* pa, const u32* pb) {
  static_assert(M >= 1, "column length");
  constexpr bool FRESH = FRESH_ARG && (ECGPU_MAC_FRESH != 0);
  if constexpr (M > 12) {            // an asm statement takes at most 30 operands: split long columns
    mac_cols<12, FRESH>(c, pa, pb);
    mac_cols<M - 12, false>(c, pa + 12, pb + 12);
  }
  if constexpr (M == 1) { if constexpr (FRESH) mac_col1_f(c, pa[0], pb[0]); else mac_col1(c, pa[0], pb[0]); }
  if constexpr (M == 2) { if constexpr (FRESH) mac_col2_f(c, pa[0], pb[0], pa[1], pb[1]); else mac_col2(c, pa[0], pb[0], pa[1], pb[1]); }
  if constexpr (M == 3) { if constexpr (FRESH) mac_col3_f(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2]); else mac_col3(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2]); }
  if constexpr (M == 4) { if constexpr (FRESH) mac_col4_f(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3]); else mac_col4(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3]); }
  if constexpr (M == 5) { if constexpr (FRESH) mac_col5_f(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4]); else mac_col5(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4]); }
  if constexpr (M == 6) { if constexpr (FRESH) mac_col6_f(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5]); else mac_col6(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5]); }
  if constexpr (M == 7) { if constexpr (FRESH) mac_col7_f(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5], pa[6], pb[6]); else mac_col7(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5], pa[6], pb[6]); }
  if constexpr (M == 8) { if constexpr (FRESH) mac_col8_f(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5], pa[6], pb[6], pa[7], pb[7]); else mac_col8(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5], pa[6], pb[6], pa[7], pb[7]); }
  if constexpr (M == 9) { if constexpr (FRESH) mac_col9_f(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5], pa[6], pb[6], pa[7], pb[7], pa[8], pb[8]); else mac_col9(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5], pa[6], pb[6], pa[7], pb[7], pa[8], pb[8]); }
  if constexpr (M == 10) { if constexpr (FRESH) mac_col10_f(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5], pa[6], pb[6], pa[7], pb[7], pa[8], pb[8], pa[9], pb[9]); else mac_col10(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5], pa[6], pb[6], pa[7], pb[7], pa[8], pb[8], pa[9], pb[9]); }
  if constexpr (M == 11) { if constexpr (FRESH) mac_col11_f(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5], pa[6], pb[6], pa[7], pb[7], pa[8], pb[8], pa[9], pb[9], pa[10], pb[10]); else mac_col11(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5], pa[6], pb[6], pa[7], pb[7], pa[8], pb[8], pa[9], pb[9], pa[10], pb[10]); }
  if constexpr (M == 12) { if constexpr (FRESH) mac_col12_f(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5], pa[6], pb[6], pa[7], pb[7], pa[8], pb[8], pa[9], pb[9], pa[10], pb[10], pa[11], pb[11]); else mac_col12(c, pa[0], pb[0], pa[1], pb[1], pa[2], pb[2], pa[3], pb[3], pa[4], pb[4], pa[5], pb[5], pa[6], pb[6], pa[7], pb[7], pa[8], pb[8], pa[9], pb[9], pa[10], pb[10], pa[11], pb[11]); }
}
// column k of the N x N schoolbook product: sum_{i+j=k} a_i b_j, plus NX extra products (xa, xb)
template <int N, int K, int NX, bool FRESH = false>
ECGPU_HD void mac_product_column(Acc96& c, const u32* a, const u32* b, const u32* xa, const u32* xb) {
  constexpr int LO = (K - (N - 1)) > 0 ? (K - (N - 1)) : 0;
  constexpr int HI = K < (N - 1) ? K : (N - 1);
  constexpr int M = HI - LO + 1;
  u32 pa[M + NX + 1], pb[M + NX + 1];
#pragma unroll
  for (int m = 0; m < M; m++) { pa[m] = a[LO + m]; pb[m] = b[K - LO - m]; }
#pragma unroll
  for (int m = 0; m < NX; m++) { pa[M + m] = xa[m]; pb[M + m] = xb[m]; }
  mac_cols<M + NX, FRESH>(c, pa, pb);
}

// pop the low word and shift the accumulator down by 32 bits
ECGPU_HD u32 acc_pop(Acc96& c) {
  u32 r = (u32)c.lo;
  c.lo = (c.lo >> 32) | ((u64)c.hi << 32);
  c.hi = 0;
  return r;
}

// the same for a signed accumulator (arithmetic shift)
ECGPU_HD u32 acc_pop_signed(Acc96& c) {
  u32 r = (u32)c.lo;
  c.lo = (c.lo >> 32) | ((u64)c.hi << 32);
  c.hi = (u32)((int32_t)c.hi >> 31);
  return r;
}

// r[0..2N) = a * b   (schoolbook, product scanning)
template <int N>
ECGPU_HD void mp_mul_wide(u32* r, const u32* a, const u32* b) {
  Acc96 c{0, 0};
#pragma unroll
  for (int k = 0; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int j = k - i;
      if (j >= 0 && j < N) {
        if (k == 0 || k == 2 * N - 2) mac_nc(c, a[i], b[j]); else mac(c, a[i], b[j]);
      }
    }
    r[k] = acc_pop(c);
  }
  r[2 * N - 1] = (u32)c.lo;
}


// column K of the off-diagonal half of a square: sum_{i < j, i + j = K} a_i a_j
template <int N, int K>
ECGPU_HD void mac_cross_column(Acc96& c, const u32* a) {
  constexpr int LO = (K - (N - 1)) > 0 ? (K - (N - 1)) : 0;
  constexpr int HI = (K - 1) / 2;            // largest i with i < K - i
  constexpr int M = HI - LO + 1;
  if constexpr (M >= 1) {
    u32 pa[M], pb[M];
#pragma unroll
    for (int m = 0; m < M; m++) { pa[m] = a[LO + m]; pb[m] = a[K - LO - m]; }
    mac_cols<M, true>(c, pa, pb);        // every cross column starts from a popped (or zero) accumulator
  }
}
template <int N, int K>
ECGPU_HD void sqr_cross_columns(u32* x, Acc96& c, const u32* a) {
  if constexpr (K < 2 * N - 2) {
    mac_cross_column<N, K>(c, a);
    x[K] = acc_pop(c);
    sqr_cross_columns<N, K + 1>(x, c, a);
  }
}

// r[0..2N) = a * a : off-diagonal products once, doubled by a one-bit funnel shift, plus the squares
template <int N>
ECGPU_HD void mp_sqr_wide(u32* r, const u32* a) {
  u32 x[2 * N];
  Acc96 c{0, 0};
  x[0] = 0;
  sqr_cross_columns<N, 1>(x, c, a);
  x[2 * N - 2] = (u32)c.lo;   // the cross sum is < 2^(64N-1): it fits, top bit clear
  x[2 * N - 1] = (u32)(c.lo >> 32);
  // r = 2*x + sum a_i^2 2^(64 i)
  u32 carry = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const u64 d = (u64)a[i] * a[i];
    const u32 lo2 = (x[2 * i] << 1) | (i ? (x[2 * i - 1] >> 31) : 0);
    const u32 hi2 = (x[2 * i + 1] << 1) | (x[2 * i] >> 31);
    r[2 * i] = addc(lo2, (u32)d, carry);
    r[2 * i + 1] = addc(hi2, (u32)(d >> 32), carry);
  }
}

// byte order helpers: canonical wire format is big-endian (to_bytes / to_repr of the reference)
ECGPU_HD u32 bswap32(u32 x) { return __builtin_bswap32(x); }

}  // namespace ecgpu
