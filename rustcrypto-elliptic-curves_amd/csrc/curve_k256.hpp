// secp256k1 group law and scalar recoding, one point per lane.
//
// Mirrors k256::ProjectivePoint / AffinePoint (k256/src/arithmetic/projective.rs:38-42,
// affine.rs:35-47): homogeneous projective (X:Y:Z), identity (0:1:0), the complete
// Renes-Costello-Batina formulas for a = 0, b = 7.  Evaluating the same formulas with exact
// field arithmetic reproduces the reference's (X, Y, Z) triples bit for bit after
// normalisation, which is what tests/ compare.
#pragma once
#include "fe_k256.hpp"

namespace ecgpu {

struct PtK256 {   // projective
  FeK256 x, y, z;
};
struct AfK256 {   // affine + infinity flag (AffinePoint::IDENTITY = (0, 0, infinity = 1), affine.rs:51-55)
  FeK256 x, y;
  u32 inf;
};

namespace k256 {

static constexpr u32 B3 = 21;   // 3 * CURVE_EQUATION_B (k256/src/arithmetic.rs:26-34)

ECGPU_HD void pt_identity(PtK256& r) { set_zero(r.x); set_one(r.y); set_zero(r.z); }
ECGPU_HD void pt_from_affine(PtK256& r, const AfK256& a) {   // projective.rs:314-323
  PtK256 id; pt_identity(id);
  const bool inf = a.inf != 0;
  FeK256 one; set_one(one);
  select(r.x, inf, id.x, a.x);
  select(r.y, inf, id.y, a.y);
  select(r.z, inf, id.z, one);
}
ECGPU_HD void pt_select(PtK256& r, bool c, const PtK256& a, const PtK256& b) {
  select(r.x, c, a.x, b.x); select(r.y, c, a.y, b.y); select(r.z, c, a.z, b.z);
}
ECGPU_HD void pt_neg(PtK256& r, const PtK256& a) { r.x = a.x; neg(r.y, a.y); r.z = a.z; }   // projective.rs:87-93
ECGPU_HD bool pt_is_identity(const PtK256& a) { return is_zero(a.z); }                      // projective.rs:483-485

// projective.rs:96-161  (RCB 2015 Algorithm 7): 12 M + small-constant multiplications
ECGPU_HD void pt_add(PtK256& r, const PtK256& p, const PtK256& q) {
  FeK256 xx, yy, zz, t0, t1, xy_pairs, yz_pairs, xz_pairs;
  mul(xx, p.x, q.x);
  mul(yy, p.y, q.y);
  mul(zz, p.z, q.z);
  add(t0, p.x, p.y); add(t1, q.x, q.y); mul(xy_pairs, t0, t1);
  add(t0, xx, yy); sub(xy_pairs, xy_pairs, t0);
  add(t0, p.y, p.z); add(t1, q.y, q.z); mul(yz_pairs, t0, t1);
  add(t0, yy, zz); sub(yz_pairs, yz_pairs, t0);
  add(t0, p.x, p.z); add(t1, q.x, q.z); mul(xz_pairs, t0, t1);
  add(t0, xx, zz); sub(xz_pairs, xz_pairs, t0);

  FeK256 bzz3, yy_m_bzz3, yy_p_bzz3, byz3, xx3, bxx9;
  mul_small(bzz3, zz, B3);
  sub(yy_m_bzz3, yy, bzz3);
  add(yy_p_bzz3, yy, bzz3);
  mul_small(byz3, yz_pairs, B3);
  mul_small(xx3, xx, 3);
  mul_small(bxx9, xx3, B3);

  mul(t0, xy_pairs, yy_m_bzz3); mul(t1, byz3, xz_pairs); sub(r.x, t0, t1);
  mul(t0, yy_p_bzz3, yy_m_bzz3); mul(t1, bxx9, xz_pairs); add(r.y, t0, t1);
  mul(t0, yz_pairs, yy_p_bzz3); mul(t1, xx3, xy_pairs); add(r.z, t0, t1);
}

// projective.rs:164-221  (RCB Algorithm 8): 11 M; returns `p` when the affine operand is the identity
ECGPU_HD void pt_add_mixed(PtK256& r, const PtK256& p, const AfK256& q) {
  FeK256 xx, yy, t0, t1, xy_pairs, yz_pairs, xz_pairs;
  mul(xx, p.x, q.x);
  mul(yy, p.y, q.y);
  add(t0, p.x, p.y); add(t1, q.x, q.y); mul(xy_pairs, t0, t1);
  add(t0, xx, yy); sub(xy_pairs, xy_pairs, t0);
  mul(t0, q.y, p.z); add(yz_pairs, t0, p.y);
  mul(t0, q.x, p.z); add(xz_pairs, t0, p.x);

  FeK256 bzz3, yy_m_bzz3, yy_p_bzz3, byz3, xx3, bxx9;
  mul_small(bzz3, p.z, B3);
  sub(yy_m_bzz3, yy, bzz3);
  add(yy_p_bzz3, yy, bzz3);
  mul_small(byz3, yz_pairs, B3);
  mul_small(xx3, xx, 3);
  mul_small(bxx9, xx3, B3);

  PtK256 s;
  mul(t0, xy_pairs, yy_m_bzz3); mul(t1, byz3, xz_pairs); sub(s.x, t0, t1);
  mul(t0, yy_p_bzz3, yy_m_bzz3); mul(t1, bxx9, xz_pairs); add(s.y, t0, t1);
  mul(t0, yz_pairs, yy_p_bzz3); mul(t1, xx3, xy_pairs); add(s.z, t0, t1);
  pt_select(r, q.inf != 0, p, s);
}

// projective.rs:225-274  (RCB Algorithm 9): 6 M + 2 S
ECGPU_HD void pt_double(PtK256& r, const PtK256& p) {
  FeK256 yy, zz, xy2, bzz3, bzz9, yy_m_bzz9, yy_p_bzz3, yy_zz, t, t0;
  sqr(yy, p.y);
  sqr(zz, p.z);
  mul(xy2, p.x, p.y); dbl(xy2, xy2);
  mul_small(bzz3, zz, B3);
  mul_small(bzz9, bzz3, 3);
  sub(yy_m_bzz9, yy, bzz9);
  add(yy_p_bzz3, yy, bzz3);
  mul(yy_zz, yy, zz);
  mul_small(t, yy_zz, 24 * 7);
  mul(t0, yy, p.y); mul(t0, t0, p.z);
  mul(r.x, xy2, yy_m_bzz9);
  mul(r.y, yy_m_bzz9, yy_p_bzz3); add(r.y, r.y, t);
  mul_small(r.z, t0, 8);
}

// ENDOMORPHISM_BETA, projective.rs:29-34 (little-endian 32-bit limbs)
ECGPU_HD void beta(FeK256& b) {
  b.v[0] = 0x719501EEu; b.v[1] = 0xC1396C28u; b.v[2] = 0x12F58995u; b.v[3] = 0x9CF04975u;
  b.v[4] = 0xAC3434E9u; b.v[5] = 0x6E64479Eu; b.v[6] = 0x657C0710u; b.v[7] = 0x7AE96A2Bu;
}
// projective.rs:287-293
ECGPU_HD void pt_endomorphism(PtK256& r, const PtK256& p) {
  FeK256 b; beta(b);
  mul(r.x, p.x, b); r.y = p.y; r.z = p.z;
}

// projective.rs:73-84 (one inversion per point; the batched form lives in the normalise kernel)
ECGPU_HD void pt_to_affine(AfK256& r, const PtK256& p) {
  FeK256 zi;
  const bool inf = is_zero(p.z);
  inv(zi, p.z);
  mul(r.x, p.x, zi); normalize(r.x, r.x);
  mul(r.y, p.y, zi); normalize(r.y, r.y);
  FeK256 z; set_zero(z);
  select(r.x, inf, z, r.x);
  select(r.y, inf, z, r.y);
  r.inf = inf ? 1u : 0u;
}

// y^2 == x^3 + 7 ?  (affine.rs:247-269)
ECGPU_HD bool af_on_curve(const AfK256& a) {
  FeK256 l, r, b;
  sqr(l, a.y);
  sqr(r, a.x); mul(r, r, a.x);
  set_u32(b, 7); add(r, r, b);
  return equal(l, r);
}

// ---------------------------------------------------------------------------------------------
// scalars: 8 little-endian 32-bit words, canonical (< n)
// ---------------------------------------------------------------------------------------------
// group order n, k256/src/lib.rs:76-79
ECGPU_HD void order(u32* n) {
  n[0] = 0xD0364141u; n[1] = 0xBFD25E8Cu; n[2] = 0xAF48A03Bu; n[3] = 0xBAAEDCE6u;
  n[4] = 0xFFFFFFFEu; n[5] = 0xFFFFFFFFu; n[6] = 0xFFFFFFFFu; n[7] = 0xFFFFFFFFu;
}
// Reduce<U256>::reduce: one conditional subtraction (scalar.rs:700-713); returns whether k was >= n
ECGPU_HD bool scalar_reduce_once(u32* k) {
  u32 n[8], t[8];
  order(n);
  const u32 bw = mp_sub<8>(t, k, n);
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = bw ? k[i] : t[i];
  return bw == 0;
}

// GLV split, k256/src/arithmetic/mul.rs:260-268 with the constants of :129-152.
// The reference computes r1, r2 modulo n and then takes (is_high ? -r : r); since |k1|, |k2| < 2^128
// (proof at mul.rs:154-257) the same magnitudes and signs come out of plain integer arithmetic:
//   c1 = round(k*g1 / 2^384), c2 = round(k*g2 / 2^384),
//   k2 = c1*(-b1) - c2*b2,   k1 = k - c1*a1 - c2*a2        (a1 = b2)
struct GlvSplit {
  u32 k1[4], k2[4];   // magnitudes, < 2^128
  bool neg1, neg2;
};

template <int NA, int NB>
ECGPU_HD void mp_mul_rect(u32* r, const u32* a, const u32* b) {   // r[0..NA+NB) = a * b
  Acc96 c{0, 0};
#pragma unroll
  for (int k = 0; k < NA + NB - 1; k++) {
#pragma unroll
    for (int i = 0; i < NA; i++) {
      const int j = k - i;
      if (j >= 0 && j < NB) mac(c, a[i], b[j]);
    }
    r[k] = acc_pop(c);
  }
  r[NA + NB - 1] = (u32)c.lo;
}

ECGPU_HD void mul_shift_384_round(u32* c, const u32* k, const u32* g) {   // wide64.rs:64-119, shift = 384
  u32 w[16];
  mp_mul_wide<8>(w, k, g);
  const u32 rnd = w[11] >> 31;
  u32 cy = rnd;
#pragma unroll
  for (int i = 0; i < 4; i++) c[i] = addc(w[12 + i], 0u, cy);
}

ECGPU_HD void glv_split(GlvSplit& s, const u32* k) {
  const u32 G1[8] = {0x45DBB031u, 0xE893209Au, 0x71E8CA7Fu, 0x3DAA8A14u, 0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u};
  const u32 G2[8] = {0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu, 0x0ABFE4C4u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u};
  const u32 MB1[4] = {0x0ABFE4C3u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u};       // -b1
  const u32 A1[4] = {0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u};        // a1 = b2
  const u32 A2[5] = {0x9D44CFD8u, 0x57C1108Du, 0xA8E2F3F6u, 0x14CA50F7u, 0x1u};   // a2 (129 bits)
  u32 c1[4], c2[4];
  mul_shift_384_round(c1, k, G1);
  mul_shift_384_round(c2, k, G2);
  // k2 = c1*(-b1) - c2*b2
  u32 p[8], q[8], d[8];
  mp_mul_rect<4, 4>(p, c1, MB1);
  mp_mul_rect<4, 4>(q, c2, A1);
  u32 bw = mp_sub<8>(d, p, q);
  s.neg2 = bw != 0;
  {
    u32 z[8], nd[8];
    mp_zero<8>(z);
    mp_sub<8>(nd, z, d);
#pragma unroll
    for (int i = 0; i < 4; i++) s.k2[i] = bw ? nd[i] : d[i];
  }
  // k1 = k - c1*a1 - c2*a2  (mod 2^256, then read as a signed number)
  u32 e[9], f[9];
  mp_mul_rect<4, 4>(e, c1, A1);
  e[8] = 0;
  mp_mul_rect<4, 5>(f, c2, A2);
  u32 t[8];
  mp_sub<8>(t, k, e);
  mp_sub<8>(t, t, f);
  const bool ng = (t[7] >> 31) != 0;
  s.neg1 = ng;
  {
    u32 z[8], nt[8];
    mp_zero<8>(z);
    mp_sub<8>(nt, z, t);
#pragma unroll
    for (int i = 0; i < 4; i++) s.k1[i] = ng ? nt[i] : t[i];
  }
}

// Radix16Decomposition (mul.rs:274-305): digit_i = nibble_i(x + 0x88..8) - 8, the carry out of the
// top nibble is the extra (always >= 0) digit.  NW = number of 32-bit words of x.
template <int NW>
struct Radix16 {
  u32 y[NW];
  u32 top;
};
template <int NW>
ECGPU_HD void radix16_recode(Radix16<NW>& r, const u32* x) {
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) r.y[i] = addc(x[i], 0x88888888u, c);
  r.top = c;
}
ECGPU_HD int radix16_digit(u32 word, int nib) { return (int)((word >> (4 * nib)) & 15u) - 8; }

}  // namespace k256
}  // namespace ecgpu
