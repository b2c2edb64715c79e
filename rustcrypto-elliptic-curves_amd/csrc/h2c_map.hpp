// Hash to curve, the field-arithmetic half of GroupDigest::hash_from_bytes:
// map_to_curve = simplified SWU for p = 3 (mod 4) in the straight-line form of
// k256/src/arithmetic/hash2curve.rs:100-143 (the external crate's generic osswu gives the same point, see
// tools/gen_h2c_constants.py), followed for secp256k1 by the 3-isogeny E' -> E (hash2curve.rs:186-262), and the sum
// Q0 + Q1 of the two mapped points (cofactor 1).  expand_message_xmd / hash_to_field are host glue
// (ecgpu/hash2curve.py).
//
// One inversion per OUTPUT instead of one per fraction (the first version inverted the SWU denominator, the isogeny's two
// denominators and the sum's Z separately: five inversions per hashed element on secp256k1): the map returns the point as
// a homogeneous projective triple - the SWU x stays the fraction xn / xd, the isogeny's polynomials are evaluated in
// homogeneous form on (xn, xd) - the two mapped points are added with the complete formulas and only the sum is brought
// to affine.  The one exponentiation of the map, a^((p-3)/4), runs on an addition chain per prime.
// Host + device (tests/hosttwin checks it against the oracle on the RFC 9380 vectors without a GPU).
#pragma once
#include "traits.hpp"

namespace ecgpu {

template <int ID> struct H2cParams;
#include "h2c_constants.inc"

namespace h2c {

// canonical little-endian words -> the curve's field form
template <class C>
ECGPU_HD void konst(typename C::Fe& r, const u32* w) {
  u32 be[C::NW];
  words_store_be<C::NW>(be, w);
  C::fe_load(r, be);
}

// r = a^((p-3)/4).  Exponents as runs of bits (x<k> = k ones), verified against big integers:
//   secp256k1  [223 ones][0][22 ones][0000][1][0][11]
//   P-256      [32 ones][31 zeros][1][96 zeros][94 ones]
//   P-384      [255 ones][0][32 ones][64 zeros][30 ones]
template <class C>
ECGPU_HD void pow_c1(typename C::Fe& r, const typename C::Fe& a) {
  if constexpr (C::ID == 0) {
    FeK256 x223, x22, x2, t;
    k256::pow_prefix(x223, x22, x2, a);
    k256::sqr_n(t, x223, 23); k256::mul(t, t, x22);
    k256::sqr_n(t, t, 5); k256::mul(t, t, a);
    k256::sqr_n(t, t, 3); k256::mul(r, t, x2);
  } else if constexpr (C::NW == 8) {
    using Fe = typename C::Fe;
    Fe x2, x3, x15, x32, x47, t;
    mont::sqr(x2, a); mont::mul(x2, x2, a);
    mont::sqr(x3, x2); mont::mul(x3, x3, a);
    t = x3; mont::pow2k_mul(t, 3, x3);            // x6
    x15 = t; mont::pow2k_mul(x15, 6, t);          // x12
    mont::pow2k_mul(x15, 3, x3);                  // x15
    x32 = x15; mont::pow2k_mul(x32, 1, a);        // x16
    t = x32; mont::pow2k_mul(x32, 16, t);         // x32
    t = x32; mont::pow2k(t, 15);
    mont::mul(x47, t, x15);                       // x47
    t = x32; mont::pow2k_mul(t, 32, a);           // [32 ones][31 zeros][1]
    mont::pow2k_mul(t, 143, x47);                 // [96 zeros][47 ones]
    mont::pow2k_mul(t, 47, x47);                  // [47 ones]
    r = t;
  } else {
    typename C::Fe x255, x32, x30;
    mont::p384_runs(x255, x32, x30, a);
    mont::pow2k_mul(x255, 33, x32);               // [0][32 ones]
    mont::pow2k_mul(x255, 94, x30);               // [64 zeros][30 ones]
    r = x255;
  }
}

// f(xn / xd) * xd^(N-1) for a polynomial with N coefficients (ascending powers), xdp[i] = xd^i
template <class C, int N>
ECGPU_HD void horner_h(typename C::Fe& r, const u32 (*co)[C::NW], const typename C::Fe& xn, const typename C::Fe* xdp) {
  typename C::Fe k, t;
  konst<C>(r, co[N - 1]);
#pragma unroll 1
  for (int i = N - 2; i >= 0; i--) {
    C::fe_mul(r, r, xn);
    konst<C>(k, co[i]);
    C::fe_mul(t, k, xdp[N - 1 - i]);
    C::fe_add(r, r, t);
  }
}

// map_to_curve(u) as a homogeneous projective point (X : Y : Z) of the suite's target curve
template <class C>
ECGPU_HD void map_to_curve(typename C::Pt& out, const typename C::Fe& u) {
  using Fe = typename C::Fe;
  using P = H2cParams<C::ID>;
  Fe Z, A, B, c2, one;
  konst<C>(Z, P::Z); konst<C>(A, P::A); konst<C>(B, P::B); konst<C>(c2, P::C2);
  C::fe_one(one);
  Fe tv1, tv2, tv3, tv4, xd, x1n, gxd, gx1, y1, y2, x2n, t;
  C::fe_sqr(tv1, u);                              // u^2
  C::fe_mul(tv3, Z, tv1);                         // Z u^2
  C::fe_sqr(tv2, tv3);
  C::fe_add(xd, tv2, tv3);                        // tv3^2 + tv3
  C::fe_add(t, xd, one);
  C::fe_mul(x1n, B, t);                           // B (xd + 1)
  C::fe_neg(t, A);
  C::fe_mul(xd, xd, t);                           // -A xd
  if (C::fe_is_zero(xd)) C::fe_mul(xd, Z, A);
  C::fe_sqr(tv2, xd);
  C::fe_mul(gxd, tv2, xd);                        // xd^3
  C::fe_mul(tv2, tv2, A);                         // A xd^2
  C::fe_sqr(t, x1n);
  C::fe_add(t, t, tv2);
  C::fe_mul(gx1, x1n, t);                         // x1n (A xd^2 + x1n^2)
  C::fe_mul(tv2, gxd, B);
  C::fe_add(gx1, gx1, tv2);                       // + B xd^3
  C::fe_sqr(tv4, gxd);
  C::fe_mul(tv2, gx1, gxd);
  C::fe_mul(tv4, tv4, tv2);                       // gx1 gxd^3
  pow_c1<C>(y1, tv4);
  C::fe_mul(y1, y1, tv2);                         // tv4^c1 tv2
  C::fe_mul(x2n, tv3, x1n);
  C::fe_mul(y2, y1, c2);
  C::fe_mul(y2, y2, tv1);
  C::fe_mul(y2, y2, u);                           // y1 c2 u^3
  C::fe_sqr(t, y1);
  C::fe_mul(t, t, gxd);
  Fe d;
  C::fe_sub(d, t, gx1);
  const bool e2 = C::fe_is_zero(d);               // y1^2 gxd == gx1
  Fe xn, y;
  C::fe_select(xn, e2, x1n, x2n);
  C::fe_select(y, e2, y1, y2);
  if (C::fe_is_odd(u) != C::fe_is_odd(y)) C::fe_neg(y, y);      // sgn0(u) == sgn0(y)
  if constexpr (C::ID == 0) {
    // 3-isogeny to secp256k1 on x = xn / xd: x' = xnum(x) / xden(x), y' = y ynum(x) / yden(x).  With the polynomials in
    // homogeneous form (degrees 3, 2, 3, 3): x' = XN / (XD xd), y' = y YN / YD, so (X : Y : Z) = (XN YD : y YN XD xd : XD xd YD)
    Fe xdp[4], XN, XD, YN, YD, dx;
    C::fe_one(xdp[0]); xdp[1] = xd; C::fe_sqr(xdp[2], xd); C::fe_mul(xdp[3], xdp[2], xd);
    horner_h<C, P::N_XNUM>(XN, P::XNUM, xn, xdp);
    horner_h<C, P::N_XDEN>(XD, P::XDEN, xn, xdp);
    horner_h<C, P::N_YNUM>(YN, P::YNUM, xn, xdp);
    horner_h<C, P::N_YDEN>(YD, P::YDEN, xn, xdp);
    C::fe_mul(dx, XD, xd);
    C::fe_mul(out.x, XN, YD);
    C::fe_mul(t, y, YN); C::fe_mul(out.y, t, dx);
    C::fe_mul(out.z, dx, YD);
  } else {
    out.x = xn;                                    // (xn / xd, y) = (xn : y xd : xd)
    C::fe_mul(out.y, y, xd);
    out.z = xd;
  }
}

}  // namespace h2c
}  // namespace ecgpu
