// Montgomery base fields for NIST P-256 and P-384 on saturated 32-bit limbs (8 / 12 VGPRs).
//
// Same internal form as the reference: elements are kept as a*R mod p with R = 2^(32 N), always
// fully reduced (p256/src/arithmetic/field.rs:43 "always in Montgomery form", R = 2^256;
// p384/src/arithmetic/field.rs:47-63 via fiat_p384_* with R = 2^384).  Both primes are
// congruent to -1 modulo 2^32, so the Montgomery constant is 1 for 32-bit words and the
// reduction needs no multiplication to find its quotient digits: in the finely integrated
// product scanning below the quotient digit of column k is simply the low word of that column.
// Multiples of the modulus are added column-wise with the same 96-bit accumulator as the
// product; limbs of p that are zero are skipped at compile time (p256: 4 of 8, p384: 2 of 12).
// Reference algorithms: p256 field.rs:240-277 (montgomery_reduce) + :293-319 (multiply);
// p384 p384_64.rs:146 (fiat_p384_mul: the same word-by-word Montgomery, 64-bit words).
#pragma once
#include "mp32.hpp"

namespace ecgpu {

struct P256Mod {
  static constexpr int N = 8;
  // p = 2^256 - 2^224 + 2^192 + 2^96 - 1   (p256/src/arithmetic/field.rs:22)
  static constexpr u32 P[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0xFFFFFFFFu};
  // R^2 mod p (field.rs:33-36), R mod p (field.rs:28-31)
  static constexpr u32 R2[8] = {0x00000003u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFBu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFDu, 0x00000004u};
  static constexpr u32 ONE[8] = {0x00000001u, 0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFEu, 0x00000000u};
  // p + 1 as a sum of word multiples: p = -1 + 2^96 + 2^192 + (2^32 - 1) 2^224.  The reduction adds m_i * p * 2^(32 i);
  // the "-1" cancels the low word of column i (that is what defines m_i), the rest are these three terms.
  static constexpr int NTERM = 3;
  static constexpr int TERM_OFF[3] = {3, 6, 7};
  static constexpr u32 TERM_MUL[3] = {1u, 1u, 0xFFFFFFFFu};
  // no subtracted terms: the one negative word of p + 1 (-2^224) is cheaper as the multiplier 2^32 - 1 above
  static constexpr int NNEG = 0;
  static constexpr int NEG_OFF[1] = {0};
  // a separate squaring (28 instead of 64 products) pays for the extra pass over the columns only when the
  // product dominates the reduction: counted out for 8 words (905 vs 920 cycles), taken for 12 (P384Mod)
#ifdef ECGPU_P256_DEDICATED_SQR                    // A/B switch (tools/ab_round3e.sh)
  static constexpr bool DEDICATED_SQR = true;
#else
  static constexpr bool DEDICATED_SQR = false;
#endif
};
struct P384Mod {
  static constexpr int N = 12;
  // p = 2^384 - 2^128 - 2^96 + 2^32 - 1   (p384/src/arithmetic/field.rs:43-45)
  static constexpr u32 P[12] = {0xFFFFFFFFu, 0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFEu, 0xFFFFFFFFu,
                                0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  // R^2 mod p = 2^768 mod p, R mod p = 2^384 mod p = 2^128 + 2^96 - 2^32 + 1
  static constexpr u32 R2[12] = {0x00000001u, 0xFFFFFFFEu, 0x00000000u, 0x00000002u, 0x00000000u, 0xFFFFFFFEu,
                                 0x00000000u, 0x00000002u, 0x00000001u, 0x00000000u, 0x00000000u, 0x00000000u};
  static constexpr u32 ONE[12] = {0x00000001u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0x00000001u, 0x00000000u,
                                  0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u};
  // p + 1 = 2^384 - 2^128 - 2^96 + 2^32: quotient digit m_i is added at words i + 1 and i + 12 and subtracted at
  // words i + 3 and i + 4.  As all-positive words this would be ten terms (word 1 and the nine words 3..11 with
  // multipliers 2^32 - 1 / 2^32 - 2), each a multiply-accumulate; the signed form is two additions and two
  // 96-bit subtractions per digit (10 instead of 20 VALU instructions), the column accumulator being read as a
  // two's-complement number while columns are in flight.
  static constexpr int NTERM = 2;
  static constexpr int TERM_OFF[2] = {1, 12};
  static constexpr u32 TERM_MUL[2] = {1u, 1u};
  static constexpr int NNEG = 2;
  static constexpr int NEG_OFF[2] = {3, 4};
  static constexpr bool DEDICATED_SQR = true;     // 66 + 12 instead of 144 products, then a reduction-only pass
};

template <class M>
struct FeMont {
  static constexpr int N = M::N;
  u32 v[M::N];
};

namespace mont {

// number of reduction terms that land in column K: term (off, mul) of quotient digit m_i lands in column i + off
template <class M>
constexpr int term_count(int K) {
  int c = 0;
  for (int t = 0; t < M::NTERM; t++) {
    const int i = K - M::TERM_OFF[t];
    if (i >= 0 && i < M::N) c++;
  }
  return c;
}

// One column of the finely integrated product scanning:
//   c += sum_{i+j=K} a_i b_j  +  sum_{terms, i = K - off in [0, N)} m_i * mul
template <class M, int K>
ECGPU_HD void fips_column(Acc96& c, const u32* a, const u32* b, const u32* m) {
  constexpr int N = M::N;
  constexpr int LO = (K - (N - 1)) > 0 ? (K - (N - 1)) : 0;     // first i of the a*b products
  constexpr int HI = K < (N - 1) ? K : (N - 1);
  constexpr int NPROD = (HI >= LO) ? HI - LO + 1 : 0;
  constexpr int NRED = term_count<M>(K);
  constexpr int TOT = NPROD + NRED;
  if constexpr (TOT > 0) {
    u32 pa[TOT], pb[TOT];
    int n = 0;
#pragma unroll
    for (int i = LO; i <= HI; i++) { pa[n] = a[i]; pb[n] = b[K - i]; n++; }
#pragma unroll
    for (int t = 0; t < M::NTERM; t++) {
      const int i = K - M::TERM_OFF[t];
      if (i >= 0 && i < N) { pa[n] = m[i]; pb[n] = M::TERM_MUL[t]; n++; }
    }
    mac_cols<TOT, (M::NNEG == 0)>(c, pa, pb);      // unsigned accumulators are always popped to c.hi == 0
  }
#pragma unroll
  for (int t = 0; t < M::NNEG; t++) {
    const int i = K - M::NEG_OFF[t];
    if (i >= 0 && i < N) acc_sub32(c, m[i]);
  }
}
// shift the accumulator down one word; with subtracted terms it is signed while columns are in flight
template <class M>
ECGPU_HD u32 fips_pop(Acc96& c) {
  if constexpr (M::NNEG > 0) return acc_pop_signed(c);
  else return acc_pop(c);
}

template <class M, int K>
ECGPU_HD void fips_low(Acc96& c, const u32* a, const u32* b, u32* m) {
  if constexpr (K < M::N) {
    fips_column<M, K>(c, a, b, m);
    // m_K = low word of the column; the "-1" of p cancels it exactly, so popping it is all that happens here
    m[K] = fips_pop<M>(c);
    fips_low<M, K + 1>(c, a, b, m);
  }
}
template <class M, int K>
ECGPU_HD void fips_high(Acc96& c, const u32* a, const u32* b, const u32* m, u32* t) {
  if constexpr (K < 2 * M::N - 1) {
    fips_column<M, K>(c, a, b, m);
    t[K - M::N] = fips_pop<M>(c);
    fips_high<M, K + 1>(c, a, b, m, t);
  }
}

// t (N + 1 words) < 2p -> r = t mod p: subtract p once if needed (p256 field.rs:272-276 sub_inner)
template <class M>
ECGPU_HD void final_subtract(FeMont<M>& r, const u32* t) {
  constexpr int N = M::N;
  u32 d[N], bw = 0;
#pragma unroll
  for (int i = 0; i < N; i++) d[i] = subb(t[i], M::P[i], bw);
  const bool use_d = (t[N] != 0) || (bw == 0);
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = use_d ? d[i] : t[i];
}

// r = a * b * R^-1 mod p
template <class M>
ECGPU_HD void mul(FeMont<M>& r, const FeMont<M>& a, const FeMont<M>& b) {
  constexpr int N = M::N;
  u32 m[N], t[N + 1];
  Acc96 c{0, 0};
  fips_low<M, 0>(c, a.v, b.v, m);
  fips_high<M, N>(c, a.v, b.v, m, t);
  fips_column<M, 2 * N - 1>(c, a.v, b.v, m);      // no products left; reduction terms that reach the top word (offset N)
  t[N - 1] = (u32)c.lo;
  t[N] = (u32)(c.lo >> 32);
  final_subtract<M>(r, t);
}
// Reduction-only columns for a product that is already on the table as 2N words: column K receives w[K] and the
// reduction terms of the earlier quotient digits.
template <class M, int K>
ECGPU_HD void redc_column(Acc96& c, const u32* w, const u32* m) {
  constexpr int N = M::N;
  constexpr int TOT = 1 + term_count<M>(K);
  u32 pa[TOT], pb[TOT];
  pa[0] = w[K]; pb[0] = 1u;
  int n = 1;
#pragma unroll
  for (int t = 0; t < M::NTERM; t++) {
    const int i = K - M::TERM_OFF[t];
    if (i >= 0 && i < N) { pa[n] = m[i]; pb[n] = M::TERM_MUL[t]; n++; }
  }
  mac_cols<TOT, (M::NNEG == 0)>(c, pa, pb);
#pragma unroll
  for (int t = 0; t < M::NNEG; t++) {
    const int i = K - M::NEG_OFF[t];
    if (i >= 0 && i < N) acc_sub32(c, m[i]);
  }
}
template <class M, int K>
ECGPU_HD void redc_low(Acc96& c, const u32* w, u32* m) {
  if constexpr (K < M::N) {
    redc_column<M, K>(c, w, m);
    m[K] = fips_pop<M>(c);
    redc_low<M, K + 1>(c, w, m);
  }
}
template <class M, int K>
ECGPU_HD void redc_high(Acc96& c, const u32* w, const u32* m, u32* t) {
  if constexpr (K < 2 * M::N - 1) {
    redc_column<M, K>(c, w, m);
    t[K - M::N] = fips_pop<M>(c);
    redc_high<M, K + 1>(c, w, m, t);
  }
}
// r = a^2 * R^-1 mod p
template <class M>
ECGPU_HD void sqr(FeMont<M>& r, const FeMont<M>& a) {
  if constexpr (M::DEDICATED_SQR) {
    constexpr int N = M::N;
    u32 w[2 * N], m[N], t[N + 1];
    mp_sqr_wide<N>(w, a.v);
    Acc96 c{0, 0};
    redc_low<M, 0>(c, w, m);
    redc_high<M, N>(c, w, m, t);
    redc_column<M, 2 * N - 1>(c, w, m);
    t[N - 1] = (u32)c.lo;
    t[N] = (u32)(c.lo >> 32);
    final_subtract<M>(r, t);
  } else {
    mul(r, a, a);
  }
}

template <class M>
ECGPU_HD void add(FeMont<M>& r, const FeMont<M>& a, const FeMont<M>& b) {   // p256 field.rs:118-134
  constexpr int N = M::N;
  u32 t[N], d[N];
  const u32 c = mp_add<N>(t, a.v, b.v);
  u32 bw = 0;
#pragma unroll
  for (int i = 0; i < N; i++) d[i] = subb(t[i], M::P[i], bw);
  const bool use_d = (c != 0) || (bw == 0);
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = use_d ? d[i] : t[i];
}
template <class M>
ECGPU_HD void sub(FeMont<M>& r, const FeMont<M>& a, const FeMont<M>& b) {   // p256 field.rs:142-197
  constexpr int N = M::N;
  u32 t[N];
  const u32 bw = mp_sub<N>(t, a.v, b.v);
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = addc(t[i], bw ? M::P[i] : 0u, c);
}
template <class M> ECGPU_HD void set_zero(FeMont<M>& r) { mp_zero<M::N>(r.v); }
template <class M> ECGPU_HD void set_one(FeMont<M>& r) {
#pragma unroll
  for (int i = 0; i < M::N; i++) r.v[i] = M::ONE[i];
}
template <class M> ECGPU_HD void neg(FeMont<M>& r, const FeMont<M>& a) { FeMont<M> z; set_zero(z); sub(r, z, a); }
template <class M> ECGPU_HD void dbl(FeMont<M>& r, const FeMont<M>& a) { add(r, a, a); }
// r = a / 2: (a + p) >> 1 for odd a (halving acts on the represented element whatever the Montgomery factor)
template <class M>
ECGPU_HD void half(FeMont<M>& r, const FeMont<M>& a) {
  constexpr int N = M::N;
  const bool odd = (a.v[0] & 1u) != 0;
  u32 t[N], c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) t[i] = addc(a.v[i], odd ? M::P[i] : 0u, c);
#pragma unroll
  for (int i = 0; i < N - 1; i++) r.v[i] = (t[i] >> 1) | (t[i + 1] << 31);
  r.v[N - 1] = (t[N - 1] >> 1) | (c << 31);
}
template <class M> ECGPU_HD bool is_zero(const FeMont<M>& a) { return mp_is_zero<M::N>(a.v); }
// top-word prefilter for the exceptional-case branches of the throughput schedules (see k256::is_zero_fast)
template <class M> ECGPU_HD bool is_zero_fast(const FeMont<M>& a) {
  if (__builtin_expect(a.v[M::N - 1] != 0, 1)) return false;
  return mp_is_zero<M::N>(a.v);
}
template <class M> ECGPU_HD bool equal(const FeMont<M>& a, const FeMont<M>& b) { return mp_eq<M::N>(a.v, b.v); }
template <class M> ECGPU_HD void select(FeMont<M>& r, bool c, const FeMont<M>& a, const FeMont<M>& b) { mp_select<M::N>(r.v, c, a.v, b.v); }

// canonical integer (little-endian limbs, < p) <-> Montgomery form
template <class M>
ECGPU_HD void to_mont(FeMont<M>& r, const u32* canon) {   // p256 field.rs:288-290
  FeMont<M> x, r2;
#pragma unroll
  for (int i = 0; i < M::N; i++) { x.v[i] = canon[i]; r2.v[i] = M::R2[i]; }
  mul(r, x, r2);
}
template <class M>
ECGPU_HD void from_mont(u32* canon, const FeMont<M>& a) {   // p256 field.rs:281-284
  FeMont<M> one, t;
  set_zero(one); one.v[0] = 1;
  mul(t, a, one);
#pragma unroll
  for (int i = 0; i < M::N; i++) canon[i] = t.v[i];
}

template <class M>
ECGPU_HD void sqr_n(FeMont<M>& r, const FeMont<M>& a, int n) {
  r = a;
  for (int i = 0; i < n; i++) sqr(r, r);
}
// acc = acc^(2^k) * m: the step every addition chain below is made of
template <class M>
ECGPU_HD void pow2k_mul(FeMont<M>& acc, int k, const FeMont<M>& m) {
#pragma unroll 1
  for (int i = 0; i < k; i++) sqr(acc, acc);
  mul(acc, acc, m);
}
template <class M>
ECGPU_HD void pow2k(FeMont<M>& acc, int k) {
#pragma unroll 1
  for (int i = 0; i < k; i++) sqr(acc, acc);
}
// Fixed addition chains for the two exponents every caller needs, a^(p-2) and a^((p+1)/4).  x<k> stands for
// a^(2^k - 1), a run of k one bits; x_{j+k} = x_j^(2^k) * x_k.  The exponents written as runs (top bit first):
//   P-256  p - 2     = [32 ones][31 zeros][1][96 zeros][94 ones][0][1]              255 S + 12 M
//          (p + 1)/4 = [32 ones][31 zeros][1][95 zeros][1][94 zeros]                253 S +  7 M
//   P-384  p - 2     = [255 ones][0][32 ones][64 zeros][30 ones][0][1]              385 S + 14 M
//          (p + 1)/4 = [255 ones][0][32 ones][63 zeros][1][30 zeros]                383 S + 13 M
// (the generic square-and-multiply they replace cost 256 S + 128 M and 384 S + ~350 M).  Same exponents as
// p256 field.rs:357-411 and p384 field.rs:95-117; p384's invert is Bernstein-Yang in the reference (field.rs:67-91),
// the inverse is unique so the field element is the same.
template <class M>
ECGPU_HD void p384_runs(FeMont<M>& x255, FeMont<M>& x32, FeMont<M>& x30, const FeMont<M>& a) {
  FeMont<M> x2, x3, x15, t;
  sqr(x2, a); mul(x2, x2, a);
  sqr(x3, x2); mul(x3, x3, a);
  t = x3; pow2k_mul(t, 3, x3);            // x6
  x15 = t; pow2k_mul(x15, 6, t);          // x12
  pow2k_mul(x15, 3, x3);                  // x15
  x30 = x15; pow2k_mul(x30, 15, x15);
  x32 = x30; pow2k_mul(x32, 2, x2);
  t = x30; pow2k_mul(t, 30, x30);         // x60
  x255 = t; pow2k_mul(x255, 60, t);       // x120
  t = x255; pow2k_mul(x255, 120, t);      // x240
  pow2k_mul(x255, 15, x15);
}
// a^(p-2): the unique inverse (0 -> 0)
template <class M>
ECGPU_HD void inv(FeMont<M>& r, const FeMont<M>& a) {
  if constexpr (M::N == 8) {
    FeMont<M> x2, x3, x15, x32, x47, t;
    sqr(x2, a); mul(x2, x2, a);
    sqr(x3, x2); mul(x3, x3, a);
    t = x3; pow2k_mul(t, 3, x3);          // x6
    x15 = t; pow2k_mul(x15, 6, t);        // x12
    pow2k_mul(x15, 3, x3);                // x15
    x32 = x15; pow2k_mul(x32, 1, a);      // x16
    t = x32; pow2k_mul(x32, 16, t);       // x32
    t = x32; pow2k(t, 15);                // x32 << 15 (still on the main chain)
    mul(x47, t, x15);                     // x47 beside it
    pow2k_mul(t, 17, a);                  // [32 ones][31 zeros][1]
    pow2k_mul(t, 143, x47);               // [96 zeros][47 ones]
    pow2k_mul(t, 47, x47);                // [47 ones]
    pow2k_mul(t, 2, a);                   // [0][1]
    r = t;
  } else {
    FeMont<M> x255, x32, x30;
    p384_runs(x255, x32, x30, a);
    pow2k_mul(x255, 33, x32);             // [0][32 ones]
    pow2k_mul(x255, 94, x30);             // [64 zeros][30 ones]
    pow2k_mul(x255, 2, a);                // [0][1]
    r = x255;
  }
}
// a^((p+1)/4) (both primes are 3 mod 4); returns whether it is a square root
template <class M>
ECGPU_HD bool sqrt(FeMont<M>& r, const FeMont<M>& a) {
  FeMont<M> t;
  if constexpr (M::N == 8) {
    FeMont<M> u;
    sqr(t, a); mul(t, t, a);              // x2
    u = t; pow2k_mul(t, 2, u);            // x4
    u = t; pow2k_mul(t, 4, u);            // x8
    u = t; pow2k_mul(t, 8, u);            // x16
    u = t; pow2k_mul(t, 16, u);           // x32
    pow2k_mul(t, 32, a);                  // [31 zeros][1]
    pow2k_mul(t, 96, a);                  // [95 zeros][1]
    pow2k(t, 94);
  } else {
    FeMont<M> x32, x30;
    p384_runs(t, x32, x30, a);
    pow2k_mul(t, 33, x32);                // [0][32 ones]
    pow2k_mul(t, 64, a);                  // [63 zeros][1]
    pow2k(t, 30);
  }
  r = t;
  FeMont<M> chk;
  sqr(chk, r);
  return equal(chk, a);
}

}  // namespace mont
}  // namespace ecgpu
