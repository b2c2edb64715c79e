// secp256k1 variable-base scalar multiplication, throughput schedule.
//
// Contract: the group element k*P (affine bytes), not the reference's projective triple - so the
// schedule is free.  Same outer structure as k256/src/arithmetic/mul.rs
// (GLV split, signed radix-16 digits, 4 doublings + 2 table additions per digit) but
//   * Jacobian coordinates: doubling 3M+4S and mixed addition 8M+3S instead of the complete
//     homogeneous formulas (8 and 12 multiplications); the exceptional cases of the incomplete
//     formulas are detected and handled explicitly so every input gives the right answer;
//   * one table [P..8P] brought to a common Z ("effective affine": the whole multiplication runs
//     on the isomorphic curve y^2 = x^3 + 7 Z8^6, the final Z is multiplied by Z8), with beta*x kept
//     beside x so the lambda-half of the GLV split reads the same table;
//   * the conversion to affine is batched per lane over consecutive results (Montgomery's trick,
//     BatchInvert / batch_normalize, k256 projective.rs:325-379): one field inversion per BATCH
//     points instead of one per point.
// This path is not constant-time (the reference's is); it is meant for bulk public data.
#pragma once
#include "curve_k256.hpp"

namespace ecgpu {

struct JacK256 {   // x = X / Z^2, y = Y / Z^3, infinity <=> Z == 0
  FeK256 x, y, z;
};
// Table slot: an affine point on the isomorphic curve, 64 contiguous bytes.  The table holds two slots per
// multiple, [2(j-1)] = (x, y) and [2(j-1)+1] = (beta*x, y), so that either GLV half reads ONE 64-byte block
// of the lane's own table (a lane-divergent read of lane-contiguous memory: no over-fetch; kept in a global
// workspace rather than the dword-interleaved private segment, where a divergent index touches up to eight
// 256-byte rows per dword).
struct alignas(16) TabSlotK256 {
  FeK256 x, y;
};
// ECGPU_K256_NO_BETA_SLOTS (A/B switch): 8 slots of (x, y) only, 512 B per lane; the lambda half multiplies x by beta when it
// reads an entry (one more multiplication on half of the additions) instead of keeping beta*x beside x (1 KB per lane).
// Window width of the throughput schedule in bits (template parameter WB of the table / digit functions below).  4 (rounds 1-3): table
// [P .. 8P], 33 digit positions per GLV half (32 signed nibbles and the carry digit), 4 doublings per position.  5 (round 4): table [P .. 16P],
// 26 positions (5-bit fields of |k_i| + 0x...1084210842 minus 16; |k_i| < 2^128 leaves the top field room for the last carry), 5 doublings per
// position: 11.5 additions less per unit for 8 more table entries.  Measured (profiles/r04_ab_measurements.txt, set four): single-term
// multiplication 134.6-135.0 -> 131.0-131.5 ms per 2^24 (-2.7 %): K256_WB = 5 is its default; the two-term kernel, which rescales both of its
// tables entry by entry, loses 6 % with 16 entries and stays on 4 bits.
#ifndef K256_WB
#define K256_WB 5
#endif
#ifdef ECGPU_K256_NO_BETA_SLOTS
constexpr int K256_SLOT_STRIDE = 1;
#else
constexpr int K256_SLOT_STRIDE = 2;
#endif
template <int WB>
struct K256Win {
  static_assert(WB == 4 || WB == 5, "window width");
  static constexpr int NE = 1 << (WB - 1);                        // table entries
  static constexpr int NPOS = (WB == 4) ? 33 : 26;                // digit positions per half
  static constexpr int SLOTS = K256_SLOT_STRIDE * NE;             // 64-byte table slots per term
};
constexpr int K256_DW = 5;                                        // recoded words per half (WB = 4: four words and the carry digit)

namespace k256 {

// In-place doubling, a = 0 (3M + 4S; with the fused Y3 below 4M + 3S of which two multiplications share one reduction), ordered for a
// short live set (at most five field elements): A = X^2, B = Y^2, Z3 = 2 Y Z, D = 4 X B, C = B^2, E = 3 A, X3 = E^2 - 2D, Y3 = E (D - X3) - 8C.
// Infinity (Z = 0) stays infinity; secp256k1 has no point with Y = 0.
ECGPU_HD void jac_double(JacK256& p) {
  FeK256 a, b, t;
  sqr(a, p.x);
  sqr(b, p.y);
  mul(p.z, p.y, p.z); shl<1>(p.z, p.z);      // Z3
  mul(p.y, p.x, b); shl<2>(p.y, p.y);        // D (in p.y)
#ifndef ECGPU_K256_NO_FUSED_DBL                  // Y3 = E (D - X3) + B (-8B) as one fused sum of two products: the squaring C = B^2 rides on the columns of the
                                                 // multiplication and shares its reduction (+0.5 % on the headline, +1.2 % on the constant-time kernel; A/B switch, tools/ab_round3h.sh)
  shl<1>(t, a); add(a, t, a);                // E (in a)
  sqr(t, a);
  sub(t, t, p.y); sub(p.x, t, p.y);          // X3 = E^2 - 2D
  sub(p.y, p.y, p.x);                        // D - X3
  shl<3>(t, b); neg(t, t);                   // -8B
  mul_add2(p.y, a, p.y, b, t);               // E (D - X3) - 8 B^2
#else
  sqr(b, b);                                 // C
  shl<1>(t, a); add(a, t, a);                // E (in a)
  sqr(t, a);
  sub(t, t, p.y); sub(p.x, t, p.y);          // X3 = E^2 - 2D
  sub(p.y, p.y, p.x); mul(p.y, a, p.y);      // E (D - X3)
  shl<3>(b, b);                              // 8C
  sub(p.y, p.y, b);
#endif
}
ECGPU_HD void jac_double(JacK256& r, const JacK256& p) { r = p; jac_double(r); }

// doubling of an affine point (Z = 1)
ECGPU_HD void jac_double_affine(JacK256& r, const FeK256& x, const FeK256& y) {
  r.x = x; r.y = y; set_one(r.z);
  jac_double(r);
}

// In-place p += (x2, y2) for an affine, non-identity (x2, y2).  8M + 3S.  `zr` (optional) receives
// the ratio Z3 / Z1 = H, which the table construction needs.  Real control flow (not selects) for
// the special cases so that no second copy of the accumulator has to stay live:
//   p at infinity            -> (x2, y2, 1)
//   same x, same y (H=R=0)   -> doubling of (x2, y2)
//   same x, opposite y       -> Z3 = Z1 * 0 = 0: infinity falls out of the formula
ECGPU_HD void jac_add_mixed(JacK256& p, const FeK256& x2, const FeK256& y2, FeK256* zr) {
  if (is_zero_fast(p.z)) {
    p.x = x2; p.y = y2; set_one(p.z);
    if (zr) set_one(*zr);
    return;
  }
  FeK256 h, r, t, u;
  sqr(t, p.z);                               // Z1Z1
  mul(h, x2, t);                             // U2
  mul(t, p.z, t); mul(r, t, y2);             // S2
  sub(h, h, p.x);                            // H
  sub(r, r, p.y);                            // R
  if (__builtin_expect(is_zero_fast(h) && is_zero(r), 0)) {   // never taken for honest GLV digits; kept exact
    if (zr) dbl(*zr, y2);
    jac_double_affine(p, x2, y2);
    return;
  }
  if (zr) *zr = h;
  mul(p.z, p.z, h);                          // Z3
  sqr(t, h);                                 // HH
  mul(h, t, h);                              // HHH
  mul(t, p.x, t);                            // V
  sqr(u, r);
  sub(u, u, h); sub(u, u, t); sub(p.x, u, t);    // X3 = R^2 - HHH - 2V
#ifdef ECGPU_K256_NO_FUSED_Y3                    // A/B switch (tools/ab_round3f.sh): two multiplications, two reductions and a subtraction
  sub(t, t, p.x); mul(t, r, t);              // R (V - X3)
  mul(h, p.y, h);                            // Y1 HHH
  sub(p.y, t, h);
#else
  sub(t, t, p.x);                            // V - X3
  neg(u, p.y);
  mul_add2(p.y, r, t, u, h);                 // Y3 = R (V - X3) + (-Y1) HHH: both products on one set of columns, ONE reduction
#endif
}
ECGPU_HD void jac_add_mixed(JacK256& r, const JacK256& p, const FeK256& x2, const FeK256& y2, FeK256* zr) {
  r = p;
  jac_add_mixed(r, x2, y2, zr);
}

// Doubling of an affine point with update (co-Z, a = 0; 2M + 4S): d = 2P in Jacobian coordinates with Z = 2y, and
// (qx, qy) = P rewritten to that same denominator, (4 x y^2, 8 y^4).
ECGPU_HD void coz_double_affine(JacK256& d, FeK256& qx, FeK256& qy, const FeK256& x, const FeK256& y) {
  FeK256 b, e, m, t;
  sqr(b, x);                                 // B = x^2
  sqr(e, y);                                 // E = y^2
  mul(qx, x, e); shl<2>(qx, qx);             // S = 4 x y^2  (= x Z^2)
  sqr(qy, e); shl<3>(qy, qy);                // 8 L = 8 y^4  (= y Z^3)
  shl<1>(m, b); add(m, m, b);                // M = 3 B
  sqr(t, m);
  sub(t, t, qx); sub(d.x, t, qx);            // X = M^2 - 2S
  sub(t, qx, d.x); mul(t, m, t);
  sub(d.y, t, qy);                           // Y = M (S - X) - 8L
  shl<1>(d.z, y);                            // Z = 2 y
}

// Co-Z addition with update (Meloni; 4M + 2S without the new Z): (qx, qy) and (r.x, r.y) are two points with the SAME
// denominator Z.  r <- q + r with denominator Z h, (qx, qy) <- q rewritten to that denominator, h = qx - r.x (before).
// Exceptional iff q = +-r, which the table chain (q = P, r = jP, 2 <= j <= 7, P of prime order) never meets.
ECGPU_HD void coz_add_update(FeK256& rx, FeK256& ry, FeK256& qx, FeK256& qy, FeK256& h) {
  FeK256 c, w2, d, t;
  sub(h, qx, rx);
  sqr(c, h);                                 // C = (X1 - X2)^2
  mul(qx, qx, c);                            // W1
  mul(w2, rx, c);                            // W2
  sub(d, qy, ry);                            // Y1 - Y2
  sub(t, qx, w2); mul(qy, qy, t);            // A1 = Y1 (W1 - W2)
  sqr(t, d);
  sub(t, t, qx); sub(rx, t, w2);             // X3 = D - W1 - W2
  sub(t, qx, rx); mul(t, d, t);
  sub(ry, t, qy);                            // Y3 = (Y1 - Y2)(W1 - X3) - A1
}

// [P, 2P, .., 8P] with a common denominator.  On return tab[j-1] = (x', beta x', y') are the affine
// coordinates of jP (slots 2(j-1) and, with beta*x, 2(j-1)+1) on the curve isomorphic by u = `zglobal`
// (x' = x u^2, y' = y u^3), i.e. Jacobian
// coordinates (x', y', zglobal) of jP on secp256k1.  P must not be the identity.
// The chain 2P, 3P = 2P + P, .. runs in co-Z form (round 3: 6 + 6 x 6 multiplications instead of 7 + 6 x 11): every step
// rewrites P to the denominator of the new multiple, so the additions are co-Z additions, no Z is ever multiplied out (only
// the ratios h_j = Z_(j+1) / Z_j are kept) and the last rewritten P IS entry 0 at the common denominator.
template <int WB>
ECGPU_HD void table_build_globalz(TabSlotK256* tab, FeK256& zglobal, const FeK256& px, const FeK256& py) {
  constexpr int NE = K256Win<WB>::NE;
  FeK256 mx[NE], my[NE];   // (mx[j], my[j]) = (j+1) P over the denominator Z_j;  Z_1 = 2 y, Z_j = Z_(j-1) zr[j]
  FeK256 zr[NE];
  JacK256 d;
  FeK256 qx, qy;
  coz_double_affine(d, qx, qy, px, py);
  mx[1] = d.x; my[1] = d.y;
  FeK256 rx = d.x, ry = d.y;
#pragma unroll 1
  for (int j = 2; j < NE; j++) {
    coz_add_update(rx, ry, qx, qy, zr[j]);
    mx[j] = rx; my[j] = ry;
  }
  FeK256 beta_; beta(beta_);
  // scale (j+1)P to the denominator of NE P: s_j = Z_(NE-1) / Z_j = prod_{i > j} zr[i]
  FeK256 s; set_one(s);
  constexpr int SS = K256_SLOT_STRIDE;
  tab[(NE - 1) * SS].x = mx[NE - 1]; tab[(NE - 1) * SS].y = my[NE - 1];
  if constexpr (SS == 2) { tab[2 * NE - 1].y = my[NE - 1]; mul(tab[2 * NE - 1].x, mx[NE - 1], beta_); }
#pragma unroll 1
  for (int j = NE - 2; j >= 1; j--) {
    mul(s, s, zr[j + 1]);                    // s = Z_(NE-1) / Z_j
    FeK256 s2, s3;
    sqr(s2, s);
    mul(s3, s2, s);
    FeK256 tx, ty;
    mul(tx, mx[j], s2);
    mul(ty, my[j], s3);
    tab[SS * j].x = tx; tab[SS * j].y = ty;
    if constexpr (SS == 2) { tab[2 * j + 1].y = ty; mul(tab[2 * j + 1].x, tx, beta_); }
  }
  mul(zglobal, s, d.z);                      // Z_(NE-1) = (Z_(NE-1) / Z_1) 2y
  tab[0].x = qx; tab[0].y = qy;              // P over Z_(NE-1): the last rewrite of the chain
  if constexpr (SS == 2) { tab[1].y = qy; mul(tab[1].x, qx, beta_); }
}

// Recoded digits of one GLV half (|k| < 2^128, four words).  w[0 .. 4]: WB = 4: k + 0x8888.. and the carry digit; WB = 5: the five
// words of k + sum_j 16 * 32^j (130 bits).  half_digit(w, j) = the signed digit at position j, j < K256_NPOS.
template <int WB>
ECGPU_HD void recode_half(u32* w, const u32* k) {
  u32 c = 0;
  if constexpr (WB == 4) {
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = addc(k[i], 0x88888888u, c);
    w[4] = c;
  } else {
#pragma unroll
    for (int i = 0; i < 5; i++) {
      u32 cw = 0;                              // bit b of the constant is set iff b = 4 (mod 5), b < 5 NPOS
#pragma unroll
      for (int b = 0; b < 32; b++) cw |= (((32 * i + b) % 5 == 4) && (32 * i + b < 5 * K256Win<WB>::NPOS)) ? (1u << b) : 0u;
      w[i] = addc(i < 4 ? k[i] : 0u, cw, c);
    }
  }
}
template <int WB>
ECGPU_HD int half_digit(const u32* w, int j) {       // j is wave-uniform: the selects below are not divergent
  if constexpr (WB == 4) {
    if (j == 32) return (int)w[4];
    u32 word = w[0];
#pragma unroll
    for (int q = 1; q < 4; q++) word = (j >> 3) == q ? w[q] : word;
    return (int)((word >> (4 * (j & 7))) & 15u) - 8;
  } else {
    const int bit = 5 * j, wi = bit >> 5, sh = bit & 31;
    u32 lo = w[0], hi = w[1];
#pragma unroll
    for (int q = 1; q < 4; q++) { lo = wi == q ? w[q] : lo; hi = wi == q ? w[q + 1] : hi; }
    if (wi == 4) { lo = w[4]; hi = 0; }
    return (int)((u32)((((u64)hi << 32) | lo) >> sh) & 31u) - 16;
  }
}

// Adds digit d of one GLV half: d in [-8, 8], `lam` selects beta*x, `neg` is the sign of that half.
ECGPU_HD void add_digit(JacK256& acc, const TabSlotK256* tab, int d, bool lam, bool neg) {
  const int ad = d < 0 ? -d : d;
  if (ad != 0) {
#ifdef ECGPU_K256_NO_BETA_SLOTS
    const TabSlotK256* e = tab + (ad - 1);
    ECGPU_TABLE_TOUCH(ad - 1);
    FeK256 x = e->x;
    FeK256 y = e->y;
    if (lam) { FeK256 b; beta(b); mul(x, x, b); }
#else
    const TabSlotK256* e = tab + (2 * (ad - 1) + (lam ? 1 : 0));
    ECGPU_TABLE_TOUCH(2 * (ad - 1) + (lam ? 1 : 0));
    FeK256 x = e->x;
    FeK256 y = e->y;
#endif
    if (neg != (d < 0)) k256::neg(y, y);
    jac_add_mixed(acc, x, y, nullptr);
  }
}

// k * P for an affine, non-identity P; result in Jacobian coordinates on secp256k1.
template <int WB>
ECGPU_HD void mul_fast_jac(JacK256& acc, const FeK256& px, const FeK256& py, const u32* k, TabSlotK256* tab) {
  constexpr int NPOS = K256Win<WB>::NPOS;
  GlvSplit s;
  glv_split(s, k);
  FeK256 zg;
  table_build_globalz<WB>(tab, zg, px, py);
  u32 w1[K256_DW], w2[K256_DW];
  recode_half<WB>(w1, s.k1);
  recode_half<WB>(w2, s.k2);
  set_zero(acc.x); set_zero(acc.y); set_zero(acc.z);      // infinity
#pragma unroll 1
  for (int i = NPOS - 1; i >= 0; i--) {
    if (i != NPOS - 1) {
#pragma unroll 1
      for (int j = 0; j < WB; j++) jac_double(acc);
    }
#pragma unroll 1
    for (int h = 0; h < 2; h++) add_digit(acc, tab, half_digit<WB>(h ? w2 : w1, i), h != 0, h ? s.neg2 : s.neg1);
  }
  mul(acc.z, acc.z, zg);     // back from the isomorphic curve
}

// Montgomery's trick over `cnt` Jacobian points held by this lane: one inversion for all of them.
// zs is scratch for cnt prefix products.  Writes canonical affine x, y limbs (0, 0 for infinity).
template <int MAXB>
ECGPU_HD void jac_batch_to_affine(FeK256* ax, FeK256* ay, u32* inf, const JacK256* pts, int cnt, FeK256* pre) {
  FeK256 acc; set_one(acc);
#pragma unroll 1
  for (int i = 0; i < cnt; i++) {
    pre[i] = acc;                                     // product of the non-zero Z before i
    FeK256 z = pts[i].z;
    const bool zr = is_zero(z);
    FeK256 one; set_one(one);
    select(z, zr, one, z);
    mul(acc, acc, z);
  }
  FeK256 ai;
  inv(ai, acc);
#pragma unroll 1
  for (int i = cnt - 1; i >= 0; i--) {
    FeK256 z = pts[i].z;
    const bool zr = is_zero(z);
    FeK256 one; set_one(one);
    select(z, zr, one, z);
    FeK256 zi, zi2, zi3;
    mul(zi, ai, pre[i]);                              // 1 / z_i
    mul(ai, ai, z);                                   // inverse of the product before i
    sqr(zi2, zi);
    mul(zi3, zi2, zi);
    FeK256 x, y, zero; set_zero(zero);
    mul(x, pts[i].x, zi2);
    mul(y, pts[i].y, zi3);
    select(ax[i], zr, zero, x);
    select(ay[i], zr, zero, y);
    inf[i] = zr ? 1u : 0u;
  }
}

}  // namespace k256
}  // namespace ecgpu
