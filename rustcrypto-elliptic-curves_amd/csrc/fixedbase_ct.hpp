// Constant-time fixed-base multiplication, the per-result body (host + device: the host twin walks it with a trace of the
// table entries it reads).  See fixedbase.hpp (mul_ct_kernel) for the schedule and its place in the library.
#pragma once
#include "jacobian.hpp"
#include "varbase_ct.hpp"        // fe_mask_select
#include "msm.hpp"               // Xyzz

namespace ecgpu {
namespace fb {

constexpr int CT_WB = 5;                                        // signed 5-bit digits
constexpr int CT_ENTRIES = 1 << (CT_WB - 1);                    // T[j][d-1] = d 2^(5j) G, d = 1..16
template <class C> constexpr int ct_nwin() { return (8 * C::NB + CT_WB - 1) / CT_WB; }

// p += (x2, y2) in XYZZ coordinates (madd-2008-s, 8M + 2S), NO exceptional-case handling: valid iff p is finite and
// p != +-(x2, y2); anything else gives garbage that the caller masks away.
template <class C>
ECGPU_HD void xyzz_add_mixed_raw(msm::Xyzz<C>& p, const typename C::Fe& x2, const typename C::Fe& y2) {
  using Fe = typename C::Fe;
  Fe pp, r, t, q;
  C::fe_mul(pp, x2, p.zz); C::fe_sub(pp, pp, p.x);           // P = U2 - X1
  C::fe_mul(r, y2, p.zzz); C::fe_sub(r, r, p.y);             // R = S2 - Y1
  C::fe_sqr(t, pp);                                          // PP
  C::fe_mul(q, p.x, t);                                      // Q = X1 PP
  C::fe_mul(p.zz, p.zz, t);                                  // ZZ3 = ZZ1 PP
  C::fe_mul(t, t, pp);                                       // PPP
  C::fe_mul(p.zzz, p.zzz, t);                                // ZZZ3 = ZZZ1 PPP
  C::fe_sqr(pp, r);
  C::fe_sub(pp, pp, t); C::fe_sub(pp, pp, q); C::fe_sub(p.x, pp, q);   // X3 = R^2 - PPP - 2Q
  C::fe_mul(t, p.y, t);                                      // Y1 PPP
  C::fe_sub(q, q, p.x); C::fe_mul(q, r, q);                  // R (Q - X3)
  C::fe_sub(p.y, q, t);                                      // (the fused difference of products, C::fe_mul_sub2, costs this kernel registers: k256 signing 4.79 against 4.53 ms per 2^20)
}

// acc = k G for a scalar k < n (reduced by the caller), returned in the reference's homogeneous projective coordinates.
// Nothing but data depends on k: the digits come from branch-free recoding, EVERY entry of window j is read (the
// address depends on j and the entry number only) and the digit's one is kept by AND / OR masks - arithmetic masks, not
// selects: a select of a loaded value lets the compiler load only under the condition, and it did (the VMEM instruction
// count followed the digits until this was an AND / OR) -, the sign is a masked negation.
//
// The addition is the plain mixed addition in XYZZ coordinates (8M + 2S; the Jacobian one, 8M + 3S, until late in round 3 -
// there is no doubling in this loop to make the fourth coordinate expensive), executed for EVERY digit and resolved by masks for a zero
// digit (the accumulator is kept) and an empty accumulator (the result is the entry itself).  Round 2 used the reference's
// complete mixed addition here (11M + 2 multiplications by b + 29 field additions on the a = -3 curves); the incomplete
// one is exception-free on the operands this loop meets (its exceptional operands are group elements - acc = O, acc = +-Q -
// whatever the coordinates).  Let S_j = sum_{i < j} d_i 32^i, so the accumulator is S_j G when
// Q = d_j 32^j G is added; -16 <= d_i <= 16 gives |S_j| <= 16 (32^j - 1) / 31 < 0.52 * 32^j, and S_NWIN = k in [0, n).
//   acc = O   iff n | S_j iff S_j = 0 (|S_j| < 32^j < n) iff all lower digits are zero (at the highest non-zero one,
//             |S_i| < 32^i <= |d_i| 32^i contradicts S_(i+1) = 0): the `empty` mask, updated from the digits, no test of Z;
//   acc = -Q  iff n | S_(j+1): S_(j+1) = 0 as an integer (it is k < n after the top window, smaller than n before), which
//             the same contradiction excludes for d_j != 0;
//   acc = +Q  iff n | D, D = S_j - d_j 32^j.  D != 0 since |S_j| < 32^j <= |d_j| 32^j.  Below the top window
//             |D| < 17 * 32^(NWIN-2) < n.  In the top window (0 <= d <= 16, 32^j = 2^(5 (NWIN-1))) |D| < 2n and D = 2 S_j - k,
//             so D = -n needs S_j < 0 and n = d 32^j + |S_j|, i.e. |S_j| = n mod 32^j: 2^255 (1 - 2^-128) for secp256k1,
//             2^255 (1 - 2^-32) for P-256, 2^380 (1 - 2^-252) for P-384 - all above the 0.52 * 32^j that bounds |S_j|
//             (tests/test_oracle_golden.py::test_fixed_base_ct_top_window checks the three moduli).
// So the only special operands are the empty accumulator and a zero digit, both masks.
template <class C>
ECGPU_HD void mul_ct_one(typename C::Pt& out, const u32* k, const AffEntry<C>* table) {
  constexpr int NW = C::NW, NWIN = ct_nwin<C>();
  static_assert((8 * C::NB) % CT_WB != 0, "the top window must have room for the last carry");
  using Fe = typename C::Fe;
  Fe one;
  C::fe_one(one);
  msm::Xyzz<C> acc;
  msm::xyzz_set_infinity<C>(acc);
  u32 empty = 0xFFFFFFFFu;                                        // all ones while every digit so far was zero
  u32 carry = 0;
#pragma unroll 1
  for (int j = 0; j < NWIN; j++) {
    const int wi = (CT_WB * j) >> 5, sh = (CT_WB * j) & 31;       // public: the window number
    u32 w0 = 0, w1 = 0;
#pragma unroll
    for (int q = 0; q < NW; q++) { w0 = (wi == q) ? k[q] : w0; w1 = (wi + 1 == q) ? k[q] : w1; }
    const u64 pair = ((u64)w1 << 32) | w0;
    const u32 v = ((u32)(pair >> sh) & ((1u << CT_WB) - 1u)) + carry;            // 0 .. 32
    carry = (j == NWIN - 1) ? 0u : ((v + (1u << (CT_WB - 1))) >> CT_WB);         // v >= 16 -> v - 32 and a carry; the top window keeps v
    const int d = (int)v - (int)(carry << CT_WB);                                // -16 .. 16
    const u32 sgn = (u32)(d >> 31), mag = ((u32)d ^ sgn) - sgn;                  // |d| without a branch
    Fe qx, qy;
    C::fe_zero(qx);
    C::fe_zero(qy);
    const AffEntry<C>* row = table + (size_t)j * CT_ENTRIES;
#pragma unroll 4
    for (int e = 0; e < CT_ENTRIES; e++) {
      ECGPU_TABLE_TOUCH(j * CT_ENTRIES + e);
      const u32 mk = 0u - (((mag ^ (u32)(e + 1)) - 1u) >> 31);                   // all ones iff mag == e + 1
      const AffEntry<C> t = row[e];
#pragma unroll
      for (int w = 0; w < NW; w++) { qx.v[w] |= t.x.v[w] & mk; qy.v[w] |= t.y.v[w] & mk; }
    }
    Fe ny;
    C::fe_neg(ny, qy);
    vbct::fe_mask_select<C>(qy, sgn, ny, qy);
    const u32 zd = 0u - ((mag - 1u) >> 31);                                      // all ones iff the digit is zero
    msm::Xyzz<C> t = acc;
    xyzz_add_mixed_raw<C>(t, qx, qy);                                            // garbage for an empty accumulator or a zero digit
    const u32 take_q = empty & ~zd;                                              // first non-zero digit: the entry itself
    vbct::fe_mask_select<C>(t.x, take_q, qx, t.x);
    vbct::fe_mask_select<C>(t.y, take_q, qy, t.y);
    vbct::fe_mask_select<C>(t.zz, take_q, one, t.zz);
    vbct::fe_mask_select<C>(t.zzz, take_q, one, t.zzz);
    vbct::fe_mask_select<C>(acc.x, zd, acc.x, t.x);
    vbct::fe_mask_select<C>(acc.y, zd, acc.y, t.y);
    vbct::fe_mask_select<C>(acc.zz, zd, acc.zz, t.zz);
    vbct::fe_mask_select<C>(acc.zzz, zd, acc.zzz, t.zzz);
    empty &= zd;
  }
  // XYZZ (X, Y, ZZ, ZZZ) = homogeneous (X ZZZ : Y ZZ : ZZ ZZZ); k = 0 (all coordinates still zero): the identity (0 : 1 : 0)
  Fe y;
  C::fe_mul(out.x, acc.x, acc.zzz);
  C::fe_mul(y, acc.y, acc.zz);
  C::fe_mul(out.z, acc.zz, acc.zzz);
  vbct::fe_mask_select<C>(out.y, empty, one, y);
}

}  // namespace fb
}  // namespace ecgpu
