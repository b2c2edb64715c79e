// secp256k1 multi-scalar multiplication: sum_i k_i * P_i for one large n (Pippenger bucket method).
//
// The reference's form of this operation is lincomb_ext over a slice (k256/src/arithmetic/mul.rs:325-393),
// a Straus interleaving with two 8-point tables per term - O(n) table memory and 128 shared
// doublings, unusable at 2^20..2^26 terms.  The result is the same group element; the schedule here is
//   0. the GLV split of every scalar (decompose_scalar, mul.rs:260-268: k = k1 + k2 lambda, |k1|, |k2| < 2^128), so a term
//      is two half-terms (k1, P) and (k2, lambda P = (beta x, y)) and there are 8 windows instead of 16: half the buckets
//      to reduce and 128 instead of 256 doublings in the serial tail, for the same number of bucket additions
//   1. signed c-bit digits of every half-scalar, computed once (c = 16: 8 windows + a carry window, 2^15 buckets per window)
//   2. two-level counting sort of the (half-term, window) pairs by bucket (512 coarse bins, then 64 buckets within a
//      bin; LDS-privatised counters at both levels, msm_kernels.hpp)
//   3. one lane per bucket sums its points in XYZZ coordinates (mixed addition 8M + 2S; exceptional cases handled)
//   4. per window sum_j j*B_j by segmented running sums, then Horner over the windows
// Every stage is a kernel over device memory; no host round trips until the final point.
#pragma once
#include "mulfast_k256.hpp"

namespace ecgpu {
namespace msm {

constexpr int C = 16;                         // window bits
constexpr int NHALF = 2;                      // GLV halves per term
constexpr int NWIN = 9;                       // 8 full windows of a 128-bit half + the carry of the signed recoding
constexpr int NDIG = NWIN * NHALF;            // digit columns per term
constexpr int NBUCKET = 1 << (C - 1);         // |digit| in 1..2^15
// bucket reduction tree: 2^15 buckets = NSEG1 x SEG1 x SEG0 per window.  Short runs keep the dependent chains of the
// two segment kernels short (16 and 32 additions); the window kernel finishes with LDS tree sums over NSEG1 lanes.
constexpr int LOG_SEG0 = 3, SEG0 = 1 << LOG_SEG0;     // buckets per level-0 run
constexpr int LOG_SEG1 = 4, SEG1 = 1 << LOG_SEG1;     // level-0 results per level-1 run
constexpr int NSEG0 = NBUCKET / SEG0;                 // level-0 runs per window (4096)
constexpr int LOG_NSEG1 = C - 1 - LOG_SEG0 - LOG_SEG1;
constexpr int NSEG1 = 1 << LOG_NSEG1;                 // level-1 runs per window (256)
constexpr int SUMW_LEN = NSEG0 / NSEG1, NSUMW = NSEG1;  // partial sums of the level-0 weighted parts, one per window-kernel lane

// general Jacobian addition (11M + 5S) with the exceptional cases handled
ECGPU_HD void jac_add(JacK256& r, const JacK256& p, const JacK256& q) {
  using namespace k256;
  if (is_zero(p.z)) { r = q; return; }
  if (is_zero(q.z)) { r = p; return; }
  FeK256 z1z1, z2z2, u1, u2, s1, s2, h, rr, t;
  sqr(z1z1, p.z); sqr(z2z2, q.z);
  mul(u1, p.x, z2z2); mul(u2, q.x, z1z1);
  mul(t, q.z, z2z2); mul(s1, p.y, t);
  mul(t, p.z, z1z1); mul(s2, q.y, t);
  sub(h, u2, u1);
  sub(rr, s2, s1);
  if (is_zero(h)) {
    if (is_zero(rr)) { r = p; jac_double(r); return; }
    set_zero(r.x); set_zero(r.y); set_zero(r.z);
    return;
  }
  FeK256 hh, hhh, v;
  sqr(hh, h); mul(hhh, hh, h); mul(v, u1, hh);
  JacK256 o;
  sqr(t, rr); sub(t, t, hhh); sub(t, t, v); sub(o.x, t, v);
  sub(t, v, o.x); mul(t, rr, t);
  mul(s1, s1, hhh); sub(o.y, t, s1);
  mul(t, p.z, q.z); mul(o.z, t, h);
  r = o;
}


// XYZZ coordinates for the bucket accumulators: x = X / ZZ, y = Y / ZZZ with ZZ^3 = ZZZ^2, infinity <=> ZZ = 0.
// Adding an affine point costs 8M + 2S (madd-2008-s), one squaring less than the Jacobian mixed addition; the 134 M
// bucket additions of a 2^23-term sum are where that squaring counts.  Buckets leave the accumulation as Jacobian
// triples (X ZZ, Y ZZZ, ZZ) - two multiplications - because the reduction tree doubles, and doubling is cheaper there.
struct XyzzK256 {
  FeK256 x, y, zz, zzz;
};
ECGPU_HD void xyzz_set_infinity(XyzzK256& p) { k256::set_zero(p.x); k256::set_zero(p.y); k256::set_zero(p.zz); k256::set_zero(p.zzz); }
// p += (x2, y2), an affine point that is not the identity.  Exceptional cases by control flow: p at infinity, the same
// point (doubling), opposite points (infinity).
ECGPU_HD void xyzz_add_mixed(XyzzK256& p, const FeK256& x2, const FeK256& y2) {
  using namespace k256;
  if (is_zero(p.zz)) {
    p.x = x2; p.y = y2; set_one(p.zz); set_one(p.zzz);
    return;
  }
  FeK256 pp, r, t, q;
  mul(pp, x2, p.zz); sub(pp, pp, p.x);           // P = U2 - X1
  mul(r, y2, p.zzz); sub(r, r, p.y);             // R = S2 - Y1
  if (__builtin_expect(is_zero(pp), 0)) {
    if (is_zero(r)) {                             // same point: 2 (x2, y2), brought from Jacobian (X, Y, Z) to (X, Y, Z^2, Z^3)
      JacK256 d;
      jac_double_affine(d, x2, y2);
      p.x = d.x; p.y = d.y;
      sqr(p.zz, d.z); mul(p.zzz, p.zz, d.z);
    } else {
      xyzz_set_infinity(p);
    }
    return;
  }
  sqr(t, pp);                                    // PP
  mul(q, p.x, t);                                // Q = X1 PP
  mul(p.zz, p.zz, t);                            // ZZ3 = ZZ1 PP
  mul(t, t, pp);                                 // PPP
  mul(p.zzz, p.zzz, t);                          // ZZZ3 = ZZZ1 PPP
  sqr(pp, r);
  sub(pp, pp, t); sub(pp, pp, q); sub(p.x, pp, q);   // X3 = R^2 - PPP - 2Q
  mul(t, p.y, t);                                // Y1 PPP
  sub(q, q, p.x); mul(q, r, q);                  // R (Q - X3)
  sub(p.y, q, t);
}
ECGPU_HD void xyzz_to_jacobian(JacK256& r, const XyzzK256& p) {
  k256::mul(r.x, p.x, p.zz);
  k256::mul(r.y, p.y, p.zzz);
  r.z = p.zz;
}

}  // namespace msm
}  // namespace ecgpu
