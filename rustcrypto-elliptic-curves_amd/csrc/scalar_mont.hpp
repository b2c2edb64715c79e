// Scalar fields (integers modulo the group order n) of the three curves, for the ECDSA batch kernels.
//
// The reference keeps k256 / p256 scalars as plain integers with a Barrett-style wide reduction
// (k256/src/arithmetic/scalar.rs:114-124, scalar/wide64.rs:121-212; p256/src/arithmetic/scalar.rs:99-117) and
// p384 scalars in Montgomery form (p384/src/arithmetic/scalar.rs:60-75); values are what is specified, so
// all three use one dense Montgomery multiplication here (R = 2^(32 L), finely integrated product scanning
// with the 96-bit column accumulator of mp32.hpp).  The group orders have no sparse structure worth
// exploiting and these operations are <2 % of an ECDSA verification, so this code is written for size,
// not for the last instruction.
#pragma once
#include "mp32.hpp"

namespace ecgpu {

struct K256Order {
  static constexpr int L = 8;
  // n (k256/src/lib.rs:76-79), R^2 mod n, R mod n, -n^-1 mod 2^32, (n - 1) / 2
  static constexpr u32 N[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  static constexpr u32 R2[8] = {0x67D7D140u, 0x896CF214u, 0x0E7CF878u, 0x741496C2u, 0x5BCD07C6u, 0xE697F5E4u, 0x81C69BC5u, 0x9D671CD5u};
  static constexpr u32 ONE[8] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 0x00000001u, 0x00000000u, 0x00000000u, 0x00000000u};
  static constexpr u32 N0INV = 0x5588B13Fu;
  static constexpr u32 HALF[8] = {0x681B20A0u, 0xDFE92F46u, 0x57A4501Du, 0x5D576E73u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu};
};
struct P256Order {
  static constexpr int L = 8;
  // n (p256/src/lib.rs:74-108)
  static constexpr u32 N[8] = {0xFC632551u, 0xF3B9CAC2u, 0xA7179E84u, 0xBCE6FAADu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0xFFFFFFFFu};
  static constexpr u32 R2[8] = {0xBE79EEA2u, 0x83244C95u, 0x49BD6FA6u, 0x4699799Cu, 0x2B6BEC59u, 0x2845B239u, 0xF3D95620u, 0x66E12D94u};
  static constexpr u32 ONE[8] = {0x039CDAAFu, 0x0C46353Du, 0x58E8617Bu, 0x43190552u, 0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0x00000000u};
  static constexpr u32 N0INV = 0xEE00BC4Fu;
  static constexpr u32 HALF[8] = {0x7E3192A8u, 0x79DCE561u, 0xD38BCF42u, 0xDE737D56u, 0xFFFFFFFFu, 0x7FFFFFFFu, 0x80000000u, 0x7FFFFFFFu};
};
struct P384Order {
  static constexpr int L = 12;
  // n (p384/src/lib.rs:50-64)
  static constexpr u32 N[12] = {0xCCC52973u, 0xECEC196Au, 0x48B0A77Au, 0x581A0DB2u, 0xF4372DDFu, 0xC7634D81u,
                                0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  static constexpr u32 R2[12] = {0x19B409A9u, 0x2D319B24u, 0xDF1AA419u, 0xFF3D81E5u, 0xFCB82947u, 0xBC3E483Au,
                                 0x4AAB1CC5u, 0xD40D4917u, 0x28266895u, 0x3FB05B7Au, 0x2B39BF21u, 0x0C84EE01u};
  static constexpr u32 ONE[12] = {0x333AD68Du, 0x1313E695u, 0xB74F5885u, 0xA7E5F24Du, 0x0BC8D220u, 0x389CB27Eu,
                                  0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u};
  static constexpr u32 N0INV = 0xE88FDC45u;
  static constexpr u32 HALF[12] = {0x666294B9u, 0x76760CB5u, 0x245853BDu, 0xAC0D06D9u, 0xFA1B96EFu, 0xE3B1A6C0u,
                                   0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu};
};

namespace smont {

template <class O>
ECGPU_HD void order(u32* n) {
#pragma unroll
  for (int i = 0; i < O::L; i++) n[i] = O::N[i];
}
// x in [0, 2n) -> [0, n)
template <class O>
ECGPU_HD void reduce_once(u32* x) {
  u32 n[O::L], d[O::L];
  order<O>(n);
  const u32 bw = mp_sub<O::L>(d, x, n);
  mp_select<O::L>(x, bw == 0, d, x);
}
// r = a * b * R^-1 mod n; a, b < n
template <class O>
ECGPU_HD void mul(u32* r, const u32* a, const u32* b) {
  constexpr int L = O::L;
  u32 n[L], m[L], t[L + 1];
  order<O>(n);
  Acc96 c{0, 0};
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) mac(c, a[i], b[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) mac(c, m[i], n[k - i]);
    m[k] = (u32)c.lo * O::N0INV;
    mac(c, m[k], n[0]);
    (void)acc_pop(c);
  }
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = k - L + 1; i < L; i++) { mac(c, a[i], b[k - i]); mac(c, m[i], n[k - i]); }
    t[k - L] = acc_pop(c);
  }
  t[L - 1] = (u32)c.lo;
  t[L] = (u32)(c.lo >> 32);
  // t < 2n: one conditional subtraction
  u32 d[L], bw = 0;
#pragma unroll
  for (int i = 0; i < L; i++) d[i] = subb(t[i], n[i], bw);
  const bool ge = (t[L] != 0) || (bw == 0);
  mp_select<L>(r, ge, d, t);
}
template <class O>
ECGPU_HD void to_mont(u32* r, const u32* a) {
  u32 r2[O::L];
#pragma unroll
  for (int i = 0; i < O::L; i++) r2[i] = O::R2[i];
  mul<O>(r, a, r2);
}
template <class O>
ECGPU_HD void from_mont(u32* r, const u32* a) {
  u32 one[O::L];
  mp_zero<O::L>(one);
  one[0] = 1;
  mul<O>(r, a, one);
}
template <class O>
ECGPU_HD void add(u32* r, const u32* a, const u32* b) {
  u32 t[O::L], d[O::L], n[O::L];
  order<O>(n);
  const u32 cy = mp_add<O::L>(t, a, b);
  const u32 bw = mp_sub<O::L>(d, t, n);
  mp_select<O::L>(r, cy != 0 || bw == 0, d, t);
}
// a^(n-2) in Montgomery form: public exponent, 4-bit fixed window (the reference inverts scalars with an
// addition chain, k256/src/arithmetic/scalar.rs:161-209, p256 scalar.rs:140-158; same value)
template <class O>
ECGPU_HD void inv(u32* r, const u32* a) {
  constexpr int L = O::L;
  u32 tab[15][L];
  mp_copy<L>(tab[0], a);
#pragma unroll 1
  for (int i = 1; i < 15; i++) mul<O>(tab[i], tab[i - 1], a);
  u32 e[L];
  order<O>(e);
  e[0] -= 2;                       // every order here ends in a word >= 2
  u32 acc[L];
#pragma unroll
  for (int i = 0; i < L; i++) acc[i] = O::ONE[i];
#pragma unroll 1
  for (int j = 8 * L - 1; j >= 0; j--) {
#pragma unroll 1
    for (int s = 0; s < 4; s++) mul<O>(acc, acc, acc);
    u32 w = e[0];
#pragma unroll
    for (int q = 1; q < L; q++) w = (j >> 3) == q ? e[q] : w;
    const u32 d = (w >> (4 * (j & 7))) & 15u;
    if (d) mul<O>(acc, acc, tab[d - 1]);
  }
  mp_copy<L>(r, acc);
}

}  // namespace smont
}  // namespace ecgpu
