// secp256k1 scalar multiplication, reference-faithful form ("exact XYZ" mode).
//
// These follow k256/src/arithmetic/mul.rs step for step - GLV split, signed radix-16 digits,
// [P..8P] tables built by repeated complete addition, 4 doublings + 2 table additions per
// digit - so that the projective (X, Y, Z) they return is the very triple the reference
// returns.  The throughput kernels (mulfast_k256.hpp) are free to use a cheaper schedule
// because their contract is the affine result.
#pragma once
#include "curve_k256.hpp"

namespace ecgpu {
namespace k256 {

// LookupTable::from (mul.rs:65-73)
ECGPU_HD void table_build(PtK256* t, const PtK256& p) {
  t[0] = p;
#pragma unroll 1
  for (int j = 0; j < 7; j++) pt_add(t[j + 1], p, t[j]);
}
// LookupTable::select (mul.rs:92-127), the reference's constant-time scan: all eight entries are read whatever the
// digit is - the addresses touched and the instructions executed do not depend on it - and merged under an
// arithmetic mask (v_bfi_b32), then the conditional negation under the sign mask.  (Lanes are independent, so no
// ballot or shuffle is involved: the scan is per lane, the digit never leaves its registers.)
ECGPU_HD void table_select(PtK256& r, const PtK256* t, int d) {
  const u32 sgn = (u32)d >> 31;                              // xmask (mul.rs:95)
  const u32 ad = ((u32)d ^ (0u - sgn)) + sgn;                // xabs = (x + xmask) ^ xmask, 0..8
  PtK256 e; pt_identity(e);
#pragma unroll 1
  for (u32 j = 1; j <= 8; j++) {
    const u32 diff = ad ^ j;
    const u32 m = 0u - ((diff - 1u) >> 31);                  // all ones iff xabs == j (diff < 2^31)
    const PtK256& c = t[j - 1];
    ECGPU_TABLE_TOUCH(j - 1);
#pragma unroll
    for (int w = 0; w < 8; w++) {
      e.x.v[w] = (e.x.v[w] & ~m) | (c.x.v[w] & m);
      e.y.v[w] = (e.y.v[w] & ~m) | (c.y.v[w] & m);
      e.z.v[w] = (e.z.v[w] & ~m) | (c.z.v[w] & m);
    }
  }
  FeK256 ny; neg(ny, e.y);
  const u32 nm = 0u - sgn;                                   // neg_mask (mul.rs:122-124)
#pragma unroll
  for (int w = 0; w < 8; w++) e.y.v[w] = (e.y.v[w] & ~nm) | (ny.v[w] & nm);
  r = e;
}

// lincomb (mul.rs:342-393) for NT terms; `tab` is scratch for 2*NT tables of 8 points.
template <int NT>
ECGPU_HD void lincomb_ref(PtK256& acc, const PtK256* pts, const u32 (*ks)[8], PtK256* tab) {
  Radix16<4> d1[NT], d2[NT];
#pragma unroll 1
  for (int t = 0; t < NT; t++) {
    GlvSplit s;
    glv_split(s, ks[t]);
    PtK256 p = pts[t], pb;
    pt_endomorphism(pb, p);
    FeK256 ny;
    neg(ny, p.y);  select(p.y, s.neg1, ny, p.y);     // mul.rs:357-362: conditional_select(x, -x, r1_sign)
    neg(ny, pb.y); select(pb.y, s.neg2, ny, pb.y);
    table_build(tab + 16 * t, p);
    table_build(tab + 16 * t + 8, pb);
    radix16_recode<4>(d1[t], s.k1);
    radix16_recode<4>(d2[t], s.k2);
  }
  pt_identity(acc);
  PtK256 e;
#pragma unroll 1
  for (int t = 0; t < NT; t++) {          // mul.rs:370-377: the 33rd digits
    table_select(e, tab + 16 * t, (int)d1[t].top);     pt_add(acc, acc, e);
    table_select(e, tab + 16 * t + 8, (int)d2[t].top); pt_add(acc, acc, e);
  }
#pragma unroll 1
  for (int i = 31; i >= 0; i--) {         // mul.rs:379-391
#pragma unroll 1
    for (int j = 0; j < 4; j++) pt_double(acc, acc);
#pragma unroll 1
    for (int h = 0; h < 2 * NT; h++) {
      const int t = h >> 1;
      const u32* y = (h & 1) ? d2[t].y : d1[t].y;
      // word i>>3, nibble i&7 (uniform across lanes)
      u32 w = y[0];
      w = (i >> 3) == 1 ? y[1] : w;
      w = (i >> 3) == 2 ? y[2] : w;
      w = (i >> 3) == 3 ? y[3] : w;
      table_select(e, tab + 8 * h, radix16_digit(w, i & 7));
      pt_add(acc, acc, e);
    }
  }
}

// lincomb (mul.rs:342-393) for a RUN-TIME number of terms (LinearCombinationExt over a slice, mul.rs:325-340): the same steps in the
// same order, so the (X, Y, Z) that comes out is the reference's; the tables (16 points per term) and the recoded digits (10 words
// per term) live in scratch the caller provides (global memory: a term's tables are 1.5 KB).
ECGPU_HD void lincomb_ref_term(const PtK256& p0, const u32* k, PtK256* tab16, u32* dig10) {
  GlvSplit s;
  glv_split(s, k);
  PtK256 p = p0, pb;
  pt_endomorphism(pb, p);
  FeK256 ny;
  neg(ny, p.y);  select(p.y, s.neg1, ny, p.y);       // mul.rs:357-362: conditional_select(x, -x, r1_sign)
  neg(ny, pb.y); select(pb.y, s.neg2, ny, pb.y);
  table_build(tab16, p);
  table_build(tab16 + 8, pb);
  Radix16<4> d1, d2;
  radix16_recode<4>(d1, s.k1);
  radix16_recode<4>(d2, s.k2);
#pragma unroll
  for (int q = 0; q < 4; q++) { dig10[q] = d1.y[q]; dig10[4 + q] = d2.y[q]; }
  dig10[8] = d1.top; dig10[9] = d2.top;
}
ECGPU_HD void lincomb_ref_run(PtK256& acc, int terms, const PtK256* tab, const u32* dig) {
  pt_identity(acc);
  PtK256 e;
#pragma unroll 1
  for (int t = 0; t < terms; t++) {       // mul.rs:370-377: the 33rd digits
    table_select(e, tab + 16 * t, (int)dig[10 * t + 8]);     pt_add(acc, acc, e);
    table_select(e, tab + 16 * t + 8, (int)dig[10 * t + 9]); pt_add(acc, acc, e);
  }
#pragma unroll 1
  for (int i = 31; i >= 0; i--) {         // mul.rs:379-391
#pragma unroll 1
    for (int j = 0; j < 4; j++) pt_double(acc, acc);
#pragma unroll 1
    for (int h = 0; h < 2 * terms; h++) {
      const u32 w = dig[10 * (h >> 1) + 4 * (h & 1) + (i >> 3)];
      table_select(e, tab + 8 * h, radix16_digit(w, i & 7));
      pt_add(acc, acc, e);
    }
  }
}

// `&P * &k` (mul.rs:442-445)
ECGPU_HD void mul_ref(PtK256& r, const PtK256& p, const u32* k, PtK256* tab) {
  u32 ks[1][8];
#pragma unroll
  for (int i = 0; i < 8; i++) ks[0][i] = k[i];
  lincomb_ref<1>(r, &p, ks, tab);
}

// precompute_gen_lookup_table (mul.rs:399-413): 33 tables of [1..8] * 2^(8i) * G.
// `gen` must hold the affine generator as a projective point; tab has 33*8 entries.
ECGPU_HD void gen_table_build(PtK256* tab, const PtK256& g0) {
  PtK256 g = g0;
#pragma unroll 1
  for (int i = 0; i < 33; i++) {
    table_build(tab + 8 * i, g);
#pragma unroll 1
    for (int j = 0; j < 8; j++) pt_double(g, g);
  }
}

// mul_by_generator with precomputed tables (mul.rs:424-439)
ECGPU_HD void mul_gen_ref(PtK256& r, const u32* k, const PtK256* tab) {
  Radix16<8> d;
  radix16_recode<8>(d, k);
  PtK256 acc, acc2, e;
  table_select(acc, tab + 8 * 32, (int)d.top);
  pt_identity(acc2);
#pragma unroll 1
  for (int i = 31; i >= 0; i--) {
    // digits 2i+1 and 2i live in word i>>2, nibbles 2*(i&3)+1 and 2*(i&3)
    u32 w = d.y[0];
#pragma unroll
    for (int j = 1; j < 8; j++) w = (i >> 2) == j ? d.y[j] : w;
    table_select(e, tab + 8 * i, radix16_digit(w, 2 * (i & 3) + 1)); pt_add(acc2, acc2, e);
    table_select(e, tab + 8 * i, radix16_digit(w, 2 * (i & 3)));     pt_add(acc, acc, e);
  }
#pragma unroll 1
  for (int j = 0; j < 4; j++) pt_double(acc2, acc2);
  pt_add(r, acc, acc2);
}

// generator (affine.rs:63-75)
ECGPU_HD void generator(PtK256& g) {
  const u32 gx[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu, 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
  const u32 gy[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u, 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
#pragma unroll
  for (int i = 0; i < 8; i++) { g.x.v[i] = gx[i]; g.y.v[i] = gy[i]; }
  set_one(g.z);
}

}  // namespace k256
}  // namespace ecgpu
