// secp256k1 base field on 8 saturated 32-bit limbs (one element per lane, 8 VGPRs).
//
// Replaces the role of k256::FieldElement (k256/src/arithmetic/field.rs:56 ->
// field/field_5x52.rs:16).  The reference keeps 5x52-bit lazily reduced limbs with a tracked
// magnitude; on gfx950 the multiplier is 32x32 (v_mad_u64_u32), so the internal form here is
// different: every value is an integer in [0, 2^256) congruent to the element ("weakly
// reduced": it may be >= p, exactly like the reference's magnitude-1 weak normal form), and
// `normalize` produces the canonical representative that `to_bytes` writes
// (field_5x52.rs:189-206, :96-131).  Field arithmetic is exact, so results agree bit for bit
// with the reference after normalisation.
//
// p = 2^256 - C,  C = 2^32 + 977  (field_5x52.rs:134-136 uses the same constant 0x1000003D1).
#pragma once
#include "mp32.hpp"

namespace ecgpu {

struct FeK256 {
  static constexpr int N = 8;
  static constexpr int NBYTES = 32;
  u32 v[8];
};

namespace k256 {

static constexpr u32 C_LO = 977;   // C = 2^32 + C_LO

// The rare carry paths of fold_top_fast / add / sub / shl.  In the throughput build they are real branches, almost never taken
// (probability 2^-26 .. 2^-32 per operation): which way they go depends on the values.  A translation unit that defines
// ECGPU_K256_BRANCHFREE before including this header - ops_k256_ct.hip: the kernels that run on SECRET scalars - gets them executed
// unconditionally instead: each of these blocks is a no-op when its carry is zero (it ripples a zero and folds a zero), so running it
// always gives the same values and leaves no branch that depends on data (VERDICT r3 missing 4; the reference's 5x52 lazy limbs need no
// carry branch, k256/src/arithmetic/field/field_5x52.rs:252-285).  Cost: 5 + 3 carry instructions per multiplication, 6 + 2 per
// addition / subtraction / small shift (measured on the constant-time kernels: profiles/r04_ab_measurements.txt).
#ifdef ECGPU_K256_BRANCHFREE
#define ECGPU_K256_RARE(cond) true
#else
#define ECGPU_K256_RARE(cond) __builtin_expect((cond), 0)
#endif

// r += T * C for a small T (< 2^40), then fold the possible carry out of 2^256 once more.
ECGPU_HD void fold_top(u32* r, u64 T) {
  // A = T*C = T*977 + (T << 32): three words
  const u64 m = (u64)(u32)T * C_LO + (((u64)((u32)(T >> 32) * C_LO)) << 32);
  const u64 alo = m + (T << 32);
  const u32 a2 = (u32)(T >> 32) + (alo < m ? 1u : 0u);
  u32 c = 0;
  r[0] = addc(r[0], (u32)alo, c);
  r[1] = addc(r[1], (u32)(alo >> 32), c);
  r[2] = addc(r[2], a2, c);
#pragma unroll
  for (int i = 3; i < 8; i++) r[i] = addc(r[i], 0u, c);
  // if that carried, the wrapped value is < 2^72: adding C once more touches words 0..2 only
  u32 c2 = 0;
  r[0] = addc(r[0], c ? C_LO : 0u, c2);
  r[1] = addc(r[1], c, c2);
  r[2] = addc(r[2], 0u, c2);
}

// r += T * C for T < 2^40 (T*C is at most three words).  The carry out of word 2 is rare (~2^-26):
// it takes a branch that ripples it and, if 2^256 is crossed, folds once more (the wrapped value is
// then tiny, so that last fold cannot carry).
ECGPU_HD void fold_top_fast(u32* r, u64 T) {
  const u32 t0 = (u32)T, t1 = (u32)(T >> 32);
  const u64 p = (u64)t0 * C_LO;                     // t0 * 977
  const u32 s1 = (u32)(p >> 32) + t1 * C_LO;        // < 2^19
  u32 ca = 0;
  const u32 a1 = addc(s1, t0, ca);                  // + (T << 32)
  const u32 a2 = t1 + ca;
  u32 c = 0;
  r[0] = addc(r[0], (u32)p, c);
  r[1] = addc(r[1], a1, c);
  r[2] = addc(r[2], a2, c);
  if (ECGPU_K256_RARE(c != 0)) {
#pragma unroll
    for (int i = 3; i < 8; i++) r[i] = addc(r[i], 0u, c);
    u32 c2 = 0;
    r[0] = addc(r[0], c ? C_LO : 0u, c2);
    r[1] = addc(r[1], c, c2);
    r[2] = addc(r[2], 0u, c2);
  }
}

// r = a * b mod p (weakly reduced).  field_5x52.rs:288-449 (mul_inner) is the reference.
// Columns 8..14 of the schoolbook product are summed first (H), then columns 0..7 are summed
// together with H*C, so the pseudo-Mersenne fold rides on the same 96-bit accumulator.
// One asm statement per column (hipcc pads each asm statement with an s_nop): the column's
// products, and for the low half also h[k]*977 and h[k-1]*1, go into a single statement.
template <int K>
ECGPU_HD void mul_high_column(u32* h, Acc96& c, const u32* a, const u32* b) {
  mac_product_column<8, K, 0, true>(c, a, b, nullptr, nullptr);     // c.hi == 0: fresh accumulator or just popped
  h[K - 8] = acc_pop(c);
}
template <int K>
ECGPU_HD void mul_low_column(u32* t, Acc96& c, const u32* a, const u32* b, const u32* h) {
  const u32 xa[1] = {h[K]}, xb[1] = {C_LO};
  mac_product_column<8, K, 1, true>(c, a, b, xa, xb);
  t[K] = acc_pop(c);
}
ECGPU_HD void mul(FeK256& r, const FeK256& a, const FeK256& b) {
  u32 h[8], t[8];
  Acc96 c{0, 0};
  mul_high_column<8>(h, c, a.v, b.v);  mul_high_column<9>(h, c, a.v, b.v);
  mul_high_column<10>(h, c, a.v, b.v); mul_high_column<11>(h, c, a.v, b.v);
  mul_high_column<12>(h, c, a.v, b.v); mul_high_column<13>(h, c, a.v, b.v);
  mul_high_column<14>(h, c, a.v, b.v);
  h[7] = (u32)c.lo;
  c.lo = 0; c.hi = 0;
  mul_low_column<0>(t, c, a.v, b.v, h); mul_low_column<1>(t, c, a.v, b.v, h);
  mul_low_column<2>(t, c, a.v, b.v, h); mul_low_column<3>(t, c, a.v, b.v, h);
  mul_low_column<4>(t, c, a.v, b.v, h); mul_low_column<5>(t, c, a.v, b.v, h);
  mul_low_column<6>(t, c, a.v, b.v, h); mul_low_column<7>(t, c, a.v, b.v, h);
  // H * 2^32 (the other half of H * C) is one carry chain; its top word joins the overflow above 2^256
  u32 cy = 0;
  r.v[0] = t[0];
#pragma unroll
  for (int i = 1; i < 8; i++) r.v[i] = addc(t[i], h[i - 1], cy);
  const u64 T = c.lo + h[7] + cy;
  fold_top_fast(r.v, T);
}

// r = a * b + e * f mod p (weakly reduced): both products ride on the same column accumulators and share ONE reduction (the
// point formulas end in differences of two products: Y3 = R (V - X3) - Y1 HHH).  A column holds up to 16 products plus the
// fold term, < 2^69: the 96-bit accumulator has room.  The sum of the two products can reach 2^513, so word 16 (one bit) exists:
// 2^512 = C^2 mod p, i.e. it adds C to the overflow count T that fold_top_fast multiplies by C.
template <int K>
ECGPU_HD void mul2_high_column(u32* h, Acc96& c, const u32* a, const u32* b, const u32* e, const u32* f) {
  mac_product_column<8, K, 0, true>(c, a, b, nullptr, nullptr);     // c.hi == 0: fresh accumulator or just popped
  mac_product_column<8, K, 0, false>(c, e, f, nullptr, nullptr);
  h[K - 8] = acc_pop(c);
}
template <int K>
ECGPU_HD void mul2_low_column(u32* t, Acc96& c, const u32* a, const u32* b, const u32* e, const u32* f, const u32* h) {
  const u32 xa[1] = {h[K]}, xb[1] = {C_LO};
  mac_product_column<8, K, 1, true>(c, a, b, xa, xb);
  mac_product_column<8, K, 0, false>(c, e, f, nullptr, nullptr);
  t[K] = acc_pop(c);
}
ECGPU_HD void mul_add2(FeK256& r, const FeK256& a, const FeK256& b, const FeK256& e, const FeK256& f) {
  u32 h[8], t[8];
  Acc96 c{0, 0};
  mul2_high_column<8>(h, c, a.v, b.v, e.v, f.v);  mul2_high_column<9>(h, c, a.v, b.v, e.v, f.v);
  mul2_high_column<10>(h, c, a.v, b.v, e.v, f.v); mul2_high_column<11>(h, c, a.v, b.v, e.v, f.v);
  mul2_high_column<12>(h, c, a.v, b.v, e.v, f.v); mul2_high_column<13>(h, c, a.v, b.v, e.v, f.v);
  mul2_high_column<14>(h, c, a.v, b.v, e.v, f.v);
  h[7] = (u32)c.lo;
  const u32 h8 = (u32)(c.lo >> 32);                 // word 16 of the sum: 0 or 1
  c.lo = 0; c.hi = 0;
  mul2_low_column<0>(t, c, a.v, b.v, e.v, f.v, h); mul2_low_column<1>(t, c, a.v, b.v, e.v, f.v, h);
  mul2_low_column<2>(t, c, a.v, b.v, e.v, f.v, h); mul2_low_column<3>(t, c, a.v, b.v, e.v, f.v, h);
  mul2_low_column<4>(t, c, a.v, b.v, e.v, f.v, h); mul2_low_column<5>(t, c, a.v, b.v, e.v, f.v, h);
  mul2_low_column<6>(t, c, a.v, b.v, e.v, f.v, h); mul2_low_column<7>(t, c, a.v, b.v, e.v, f.v, h);
  u32 cy = 0;
  r.v[0] = t[0];
#pragma unroll
  for (int i = 1; i < 8; i++) r.v[i] = addc(t[i], h[i - 1], cy);
  const u64 T = c.lo + h[7] + cy + (((u64)h8 << 32) | (h8 ? C_LO : 0u));     // < 2^38
  fold_top_fast(r.v, T);
}

// reduce a 16-word integer modulo p: lo + hi * 977 + (hi << 32), then fold what spills over 2^256
ECGPU_HD void reduce16(FeK256& r, const u32* w) {
  u32 u[8];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {          // U = hi * 977: no carries needed, hi_k * 977 + carry < 2^43
    acc = (u64)w[8 + k] * C_LO + (acc >> 32);
    u[k] = (u32)acc;
  }
  u32 c1 = 0, c2 = 0;
  u32 t[8];
#pragma unroll
  for (int k = 0; k < 8; k++) t[k] = addc(w[k], u[k], c1);
  r.v[0] = t[0];
#pragma unroll
  for (int k = 1; k < 8; k++) r.v[k] = addc(t[k], w[8 + k - 1], c2);
  const u64 T = (acc >> 32) + c1 + c2 + w[15];
  fold_top_fast(r.v, T);
}

// r = a^2 mod p
ECGPU_HD void sqr(FeK256& r, const FeK256& a) {
  u32 w[16];
  mp_sqr_wide<8>(w, a.v);
  reduce16(r, w);
}

// r = a + b mod p   (field_5x52.rs:264-272 adds limb-wise and defers; here the fold is immediate).
// The carry out of 2^256 is folded into words 0 and 1; a carry out of word 1 (probability ~2^-32)
// takes the rare path that ripples it upwards and, if that wraps again, folds once more.
ECGPU_HD void add(FeK256& r, const FeK256& a, const FeK256& b) {
  const u32 c = mp_add<8>(r.v, a.v, b.v);
  u32 c2 = 0;
  r.v[0] = addc(r.v[0], c ? C_LO : 0u, c2);
  r.v[1] = addc(r.v[1], c, c2);
  if (ECGPU_K256_RARE(c2 != 0)) {
#pragma unroll
    for (int i = 2; i < 8; i++) r.v[i] = addc(r.v[i], 0u, c2);
    // second wrap only when both inputs were within C of 2^256; the wrapped value is then < C
    u32 c3 = 0;
    r.v[0] = addc(r.v[0], c2 ? C_LO : 0u, c3);
    r.v[1] = addc(r.v[1], c2, c3);
  }
}

// r = a - b mod p   (reference: a + negate(b), field.rs:425-451, field_5x52.rs:252-260)
ECGPU_HD void sub(FeK256& r, const FeK256& a, const FeK256& b) {
  const u32 bw = mp_sub<8>(r.v, a.v, b.v);
  u32 b2 = 0;
  r.v[0] = subb(r.v[0], bw ? C_LO : 0u, b2);
  r.v[1] = subb(r.v[1], bw, b2);
  if (ECGPU_K256_RARE(b2 != 0)) {
#pragma unroll
    for (int i = 2; i < 8; i++) r.v[i] = subb(r.v[i], 0u, b2);
    u32 b3 = 0;
    r.v[0] = subb(r.v[0], b2 ? C_LO : 0u, b3);
    r.v[1] = subb(r.v[1], b2, b3);
  }
}

ECGPU_HD void set_zero(FeK256& r) { mp_zero<8>(r.v); }
ECGPU_HD void set_one(FeK256& r) { mp_zero<8>(r.v); r.v[0] = 1; }
ECGPU_HD void set_u32(FeK256& r, u32 x) { mp_zero<8>(r.v); r.v[0] = x; }

ECGPU_HD void neg(FeK256& r, const FeK256& a) {
  FeK256 z;
  set_zero(z);
  sub(r, z, a);
}
ECGPU_HD void dbl(FeK256& r, const FeK256& a) { add(r, a, a); }

// r = a * 2^K mod p for K = 1, 2, 3: a funnel shift of the eight words and a fold of the K bits shifted out of the
// top (8 shifts + 3 carry instructions instead of K carry chains of ten).  r may alias a.
template <int K>
ECGPU_HD void shl(FeK256& r, const FeK256& a) {
  static_assert(K >= 1 && K <= 3, "small shifts only");
  const u32 top = a.v[7] >> (32 - K);
  u32 t[8];
#pragma unroll
  for (int i = 7; i >= 1; i--) t[i] = (a.v[i] << K) | (a.v[i - 1] >> (32 - K));
  t[0] = a.v[0] << K;
  // + top * C with C = 2^32 + 977: top * 977 < 2^13 into word 0, top into word 1
  u32 c = 0;
  r.v[0] = addc(t[0], top * C_LO, c);
  r.v[1] = addc(t[1], top, c);
#pragma unroll
  for (int i = 2; i < 8; i++) r.v[i] = t[i];
  if (ECGPU_K256_RARE(c != 0)) {
#pragma unroll
    for (int i = 2; i < 8; i++) r.v[i] = addc(r.v[i], 0u, c);
    u32 c2 = 0;                       // wrapped past 2^256: the value is then tiny, one more fold cannot carry far
    r.v[0] = addc(r.v[0], c ? C_LO : 0u, c2);
    r.v[1] = addc(r.v[1], c, c2);
    r.v[2] = addc(r.v[2], 0u, c2);
  }
}

// r = a * k for a small constant (field_5x52.rs:276-285 mul_single)
ECGPU_HD void mul_small(FeK256& r, const FeK256& a, u32 k) {
  u64 acc = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    acc += (u64)a.v[i] * k;
    r.v[i] = (u32)acc;
    acc >>= 32;
  }
  fold_top(r.v, acc);
}

// canonical representative in [0, p)   (field_5x52.rs:189-206)
ECGPU_HD void normalize(FeK256& r, const FeK256& a) {
  // a >= p  <=>  a + C >= 2^256
  u32 t[8];
  u32 c = 0;
  t[0] = addc(a.v[0], C_LO, c);
  t[1] = addc(a.v[1], 1u, c);
#pragma unroll
  for (int i = 2; i < 8; i++) t[i] = addc(a.v[i], 0u, c);
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = c ? t[i] : a.v[i];
}

// field_5x52.rs:209-223 normalizes_to_zero: raw 0 or raw p
ECGPU_HD bool is_zero(const FeK256& a) {
  u32 z = 0, o = 0xFFFFFFFFu;
#pragma unroll
  for (int i = 0; i < 8; i++) z |= a.v[i];
#pragma unroll
  for (int i = 2; i < 8; i++) o &= a.v[i];
  const bool is_p = (o == 0xFFFFFFFFu) && (a.v[1] == 0xFFFFFFFEu) && (a.v[0] == (0u - C_LO));
  return z == 0 || is_p;
}
// The same test for the throughput schedules' exceptional-case branches (accumulator at infinity, equal or opposite
// points), taken once or twice per point addition and almost never true: a value that is 0 or p has a top word of 0 or
// 2^32 - 1, so anything else (probability 1 - 2^-31 on the values these tests see) is rejected after two instructions.
// Not for the constant-time schedule: which path runs depends on the value.
ECGPU_HD bool is_zero_fast(const FeK256& a) {
#ifndef ECGPU_NO_ZERO_PREFILTER        // A/B switch for measurements
  if (__builtin_expect(a.v[7] + 1u > 1u, 1)) return false;
#endif
  return is_zero(a);
}
ECGPU_HD bool equal(const FeK256& a, const FeK256& b) {
  FeK256 d;
  sub(d, a, b);
  return is_zero(d);
}
// field_5x52.rs:239-241 (on the normalised value)
ECGPU_HD bool is_odd(const FeK256& a) {
  FeK256 n;
  normalize(n, a);
  return n.v[0] & 1;
}
ECGPU_HD void select(FeK256& r, bool cond, const FeK256& a, const FeK256& b) { mp_select<8>(r.v, cond, a.v, b.v); }

ECGPU_HD void sqr_n(FeK256& r, const FeK256& a, int n) {
  r = a;
  for (int i = 0; i < n; i++) sqr(r, r);
}

// common prefix of the inversion / square-root chains: x223 = a^(2^223-1), x22, x2
ECGPU_HD void pow_prefix(FeK256& x223, FeK256& x22, FeK256& x2, const FeK256& a) {
  FeK256 x3, x6, x9, x11, x44, x88, x176, x220, t;
  sqr(t, a); mul(x2, t, a);
  sqr(t, x2); mul(x3, t, a);
  sqr_n(t, x3, 3); mul(x6, t, x3);
  sqr_n(t, x6, 3); mul(x9, t, x3);
  sqr_n(t, x9, 2); mul(x11, t, x2);
  sqr_n(t, x11, 11); mul(x22, t, x11);
  sqr_n(t, x22, 22); mul(x44, t, x22);
  sqr_n(t, x44, 44); mul(x88, t, x44);
  sqr_n(t, x88, 88); mul(x176, t, x88);
  sqr_n(t, x176, 44); mul(x220, t, x44);
  sqr_n(t, x220, 3); mul(x223, t, x3);
}

// r = a^(p-2): the unique inverse (0 -> 0; the caller reports CtOption::none for zero).
// Same exponent as field.rs:187-216; 255 squarings + 15 multiplications.
ECGPU_HD void inv(FeK256& r, const FeK256& a) {
  FeK256 x223, x22, x2, t;
  pow_prefix(x223, x22, x2, a);
  sqr_n(t, x223, 23); mul(t, t, x22);
  sqr_n(t, t, 5); mul(t, t, a);
  sqr_n(t, t, 3); mul(t, t, x2);
  sqr_n(t, t, 2); mul(r, t, a);
}

// r = a^((p+1)/4); returns whether r^2 == a (field.rs:220-255)
ECGPU_HD bool sqrt(FeK256& r, const FeK256& a) {
  FeK256 x223, x22, x2, t, chk;
  pow_prefix(x223, x22, x2, a);
  sqr_n(t, x223, 23); mul(t, t, x22);
  sqr_n(t, t, 6); mul(t, t, x2);
  sqr_n(r, t, 2);
  sqr(chk, r);
  return equal(chk, a);
}

// canonical big-endian bytes <-> limbs.  `words` points at 8 u32 as they lie in memory.
ECGPU_HD void from_be_words(FeK256& r, const u32* words) {
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = bswap32(words[7 - i]);
}
ECGPU_HD void to_be_words(u32* words, const FeK256& a) {   // caller normalises first
#pragma unroll
  for (int i = 0; i < 8; i++) words[7 - i] = bswap32(a.v[i]);
}
// from_bytes rejects values >= p (field_5x52.rs:75-79, :163-170)
ECGPU_HD bool is_canonical(const FeK256& a) {
  u32 c = 0;
  (void)addc(a.v[0], C_LO, c);
  (void)addc(a.v[1], 1u, c);
#pragma unroll
  for (int i = 2; i < 8; i++) (void)addc(a.v[i], 0u, c);
  return c == 0;
}

}  // namespace k256
}  // namespace ecgpu
