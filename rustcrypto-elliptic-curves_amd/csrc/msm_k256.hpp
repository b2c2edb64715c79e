// secp256k1 multi-scalar multiplication: sum_i k_i * P_i for one large n (Pippenger bucket method).
//
// The reference's form of this operation is lincomb_ext over a slice (k256/src/arithmetic/mul.rs:325-393),
// a Straus interleaving with two 8-point tables per term - O(n) table memory and 128 shared
// doublings, unusable at 2^20..2^26 terms.  The result is the same group element; the schedule here is
//   1. signed c-bit digits of every scalar (c = 16: 16 windows + a carry window, 2^15 buckets per window)
//   2. counting sort of the (term, window) pairs by bucket: LDS-privatised histogram, exclusive scan, scatter
//   3. one lane per bucket sums its points with Jacobian mixed additions (exceptional cases handled)
//   4. per window sum_j j*B_j by segmented running sums, then Horner over the windows
// Every stage is a kernel over device memory; no host round trips until the final point.
#pragma once
#include "mulfast_k256.hpp"

namespace ecgpu {
namespace msm {

constexpr int C = 16;                         // window bits
constexpr int NWIN = 17;                      // 16 full windows + the carry of the signed recoding
constexpr int NBUCKET = 1 << (C - 1);         // |digit| in 1..2^15
constexpr int SORT_CHUNKS = 15;               // workgroups per window in the counting sort: 17 x 15 = 255, one per CU
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

}  // namespace msm
}  // namespace ecgpu
