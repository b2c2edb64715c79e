// Multi-scalar multiplication: sum_i k_i * P_i for one large n (Pippenger bucket method), all three curves.
//
// The reference's form of this operation is lincomb_ext over a slice (k256/src/arithmetic/mul.rs:325-393), a Straus
// interleaving with two 8-point tables per term - O(n) table memory and 128 shared doublings, unusable at 2^20..2^26
// terms; the primeorder curves only have the two-term default (primeorder/src/projective.rs:415-420).  The result is
// the same group element; the schedule here is
//   0. secp256k1: the GLV split of every scalar (decompose_scalar, mul.rs:260-268: k = k1 + k2 lambda, |k1|, |k2| < 2^128),
//      so a term is two half-terms (k1, P) and (k2, lambda P = (beta x, y)) and the windows cover 128 instead of 256 bits:
//      half the buckets to reduce and half the doublings in the serial tail, for the same number of bucket additions.
//      P-256 / P-384 have no such endomorphism: one half; k > n/2 is replaced by n - k and -P so that the top digit stays
//      below half the window range.
//   1. signed c-bit digits of every (half-)scalar, computed once; c = 16 (2^15 buckets per window, 8 windows and a
//      carry window for secp256k1) or c = 19 for large sums (2^18 buckets, 7 windows, no carry window: 18 % fewer bucket
//      additions).  The points are brought to the field's internal form once as well.
//   2. two-level counting sort of the (half-term, window) entries by bucket (coarse bins, then the buckets within a bin;
//      LDS-privatised counters at both levels, msm_kernels.hpp)
//   3. bucket sums in XYZZ coordinates (mixed addition 8M + 2S) over EQUAL runs of the sorted entries: a lane sums the same
//      number of consecutive entries whatever buckets they belong to, writes the buckets that end inside its run and leaves
//      the two it shares with its neighbours as pieces; exceptional cases of the addition handled
//   4. per window sum_j j*B_j by a tree of running sums, then Horner over the windows
// Every stage is a kernel over device memory; no host round trips until the final point.
#pragma once
#include "jacobian.hpp"
#include "mulfast_k256.hpp"

namespace ecgpu {
namespace msm {

// doubling and general addition on Jacobian triples: secp256k1 keeps its own doubling (funnel-shift small multiples)
template <class C>
ECGPU_HD void pt_dbl(Jac<C>& p) { jac::dbl<C>(p); }
template <>
ECGPU_HD void pt_dbl<CurveK256>(Jac<CurveK256>& p) {
  JacK256 t;
  t.x = p.x; t.y = p.y; t.z = p.z;
  k256::jac_double(t);
  p.x = t.x; p.y = t.y; p.z = t.z;
}
template <class C>
ECGPU_HD void pt_add(Jac<C>& r, const Jac<C>& p, const Jac<C>& q) { jac::add<C>(r, p, q); }

// XYZZ coordinates for the bucket accumulators: x = X / ZZ, y = Y / ZZZ with ZZ^3 = ZZZ^2, infinity <=> ZZ = 0.
// Adding an affine point costs 8M + 2S (madd-2008-s, independent of the curve coefficients), one squaring less than the
// Jacobian mixed addition; the 10^8 bucket additions of a 2^23-term sum are where that squaring counts.  Buckets leave the
// accumulation as Jacobian triples (X ZZ, Y ZZZ, ZZ) - two multiplications - because the reduction tree doubles, and
// doubling is cheaper there.
template <class C>
struct Xyzz {
  typename C::Fe x, y, zz, zzz;
};
template <class C>
ECGPU_HD void xyzz_set_infinity(Xyzz<C>& p) { C::fe_zero(p.x); C::fe_zero(p.y); C::fe_zero(p.zz); C::fe_zero(p.zzz); }
// p += (x2, y2), an affine point (internal field form) that is not the identity.  Exceptional cases by control flow:
// p at infinity, the same point (doubling), opposite points (infinity).
template <class C>
ECGPU_HD void xyzz_add_mixed(Xyzz<C>& p, const typename C::Fe& x2, const typename C::Fe& y2) {
  using Fe = typename C::Fe;
  if (C::fe_is_zero_fast(p.zz)) {
    p.x = x2; p.y = y2; C::fe_one(p.zz); C::fe_one(p.zzz);
    return;
  }
  Fe pp, r, t, q;
  C::fe_mul(pp, x2, p.zz); C::fe_sub(pp, pp, p.x);           // P = U2 - X1
  C::fe_mul(r, y2, p.zzz); C::fe_sub(r, r, p.y);             // R = S2 - Y1
  if (__builtin_expect(C::fe_is_zero_fast(pp), 0)) {
    if (C::fe_is_zero(r)) {                       // same point: 2 (x2, y2), brought from Jacobian (X, Y, Z) to (X, Y, Z^2, Z^3)
      Jac<C> d;
      d.x = x2; d.y = y2; C::fe_one(d.z);
      pt_dbl<C>(d);
      p.x = d.x; p.y = d.y;
      C::fe_sqr(p.zz, d.z); C::fe_mul(p.zzz, p.zz, d.z);
    } else {
      xyzz_set_infinity<C>(p);
    }
    return;
  }
  C::fe_sqr(t, pp);                                          // PP
  C::fe_mul(q, p.x, t);                                      // Q = X1 PP
  C::fe_mul(p.zz, p.zz, t);                                  // ZZ3 = ZZ1 PP
  C::fe_mul(t, t, pp);                                       // PPP
  C::fe_mul(p.zzz, p.zzz, t);                                // ZZZ3 = ZZZ1 PPP
  C::fe_sqr(pp, r);
  C::fe_sub(pp, pp, t); C::fe_sub(pp, pp, q); C::fe_sub(p.x, pp, q);   // X3 = R^2 - PPP - 2Q
  C::fe_sub(q, q, p.x);                                      // Q - X3
  C::fe_mul_sub2(p.y, r, q, p.y, t);                         // Y3 = R (Q - X3) - Y1 PPP
}
// p += (x2, y2) for p = (X1, Y1, 1, 1) AFFINE and finite (the caller knows: the first addition after a set): 4M + 2S, ZZ3 = PP and
// ZZZ3 = PPP fall out of the formula.  Same special cases by control flow.
template <class C>
ECGPU_HD void xyzz_add_affine(Xyzz<C>& p, const typename C::Fe& x2, const typename C::Fe& y2) {
  using Fe = typename C::Fe;
  Fe pp, r, t, q;
  C::fe_sub(pp, x2, p.x);
  C::fe_sub(r, y2, p.y);
  if (__builtin_expect(C::fe_is_zero_fast(pp), 0)) {
    if (C::fe_is_zero(r)) {
      Jac<C> d;
      d.x = x2; d.y = y2; C::fe_one(d.z);
      pt_dbl<C>(d);
      p.x = d.x; p.y = d.y;
      C::fe_sqr(p.zz, d.z); C::fe_mul(p.zzz, p.zz, d.z);
    } else {
      xyzz_set_infinity<C>(p);
    }
    return;
  }
  C::fe_sqr(p.zz, pp);                                       // ZZ3 = PP
  C::fe_mul(q, p.x, p.zz);                                   // Q = X1 PP
  C::fe_mul(p.zzz, p.zz, pp);                                // ZZZ3 = PPP
  C::fe_sqr(t, r);
  C::fe_sub(t, t, p.zzz); C::fe_sub(t, t, q); C::fe_sub(p.x, t, q);    // X3 = R^2 - PPP - 2Q
  C::fe_sub(q, q, p.x);                                      // Q - X3
  C::fe_mul_sub2(p.y, r, q, p.y, p.zzz);                     // Y3 = R (Q - X3) - Y1 PPP
}
// p += q, both in XYZZ coordinates (add-2008-s: 12M + 2S; a bucket left in pieces by the equal-run bucket sums is the sum of
// its pieces).  Exceptional cases by control flow: either operand at infinity, the same point (doubling, through the
// Jacobian doubling), opposite points (infinity).
template <class C>
ECGPU_HD void jacobian_to_xyzz(Xyzz<C>& r, const Jac<C>& p);
template <class C>
ECGPU_HD void xyzz_add(Xyzz<C>& p, const Xyzz<C>& q) {
  using Fe = typename C::Fe;
  if (C::fe_is_zero_fast(q.zz)) return;
  if (C::fe_is_zero_fast(p.zz)) { p = q; return; }
  Fe u1, u2, s1, s2, pp, t;
  C::fe_mul(u1, p.x, q.zz); C::fe_mul(u2, q.x, p.zz);
  C::fe_mul(s1, p.y, q.zzz); C::fe_mul(s2, q.y, p.zzz);
  C::fe_sub(u2, u2, u1);                                     // P = U2 - U1
  C::fe_sub(s2, s2, s1);                                     // R = S2 - S1
  if (__builtin_expect(C::fe_is_zero_fast(u2), 0)) {
    if (C::fe_is_zero(s2)) {                                 // the same point: 2 p through (X ZZ, Y ZZZ, ZZ)
      Jac<C> d;
      C::fe_mul(d.x, p.x, p.zz); C::fe_mul(d.y, p.y, p.zzz); d.z = p.zz;
      pt_dbl<C>(d);
      jacobian_to_xyzz<C>(p, d);
    } else {
      xyzz_set_infinity<C>(p);
    }
    return;
  }
  C::fe_sqr(pp, u2);                                         // PP
  C::fe_mul(u1, u1, pp);                                     // Q = U1 PP
  C::fe_mul(t, p.zz, q.zz); C::fe_mul(p.zz, t, pp);          // ZZ3 = ZZ1 ZZ2 PP
  C::fe_mul(pp, pp, u2);                                     // PPP
  C::fe_mul(t, p.zzz, q.zzz); C::fe_mul(p.zzz, t, pp);       // ZZZ3 = ZZZ1 ZZZ2 PPP
  C::fe_sqr(t, s2);
  C::fe_sub(t, t, pp); C::fe_sub(t, t, u1); C::fe_sub(p.x, t, u1);     // X3 = R^2 - PPP - 2Q
  C::fe_mul(s1, s1, pp);                                     // S1 PPP
  C::fe_sub(u1, u1, p.x); C::fe_mul(u1, s2, u1);             // R (Q - X3)
  C::fe_sub(p.y, u1, s1);
}
// (X, Y, Z) -> (X, Y, Z^2, Z^3): the same point, x = X / Z^2, y = Y / Z^3
template <class C>
ECGPU_HD void jacobian_to_xyzz(Xyzz<C>& r, const Jac<C>& p) {
  r.x = p.x; r.y = p.y;
  C::fe_sqr(r.zz, p.z);
  C::fe_mul(r.zzz, r.zz, p.z);
}
template <class C>
ECGPU_HD void xyzz_to_jacobian(Jac<C>& r, const Xyzz<C>& p) {
  C::fe_mul(r.x, p.x, p.zz);
  C::fe_mul(r.y, p.y, p.zzz);
  r.z = p.zz;
}

}  // namespace msm
}  // namespace ecgpu
