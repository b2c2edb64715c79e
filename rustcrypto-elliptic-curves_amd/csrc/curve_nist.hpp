// NIST P-256 / P-384 group law and scalar multiplication (the primeorder crate), one point per lane.
//
// Mirrors primeorder::ProjectivePoint<C> / AffinePoint<C> (primeorder/src/projective.rs:37-41,
// affine.rs:29-41) with PointArithmetic = EquationAIsMinusThree
// (primeorder/src/point_arithmetic.rs:199-317: Renes-Costello-Batina Algorithms 4, 5, 6) and the
// unsigned 4-bit fixed-window `mul` (primeorder/src/projective.rs:106-150).  Same formulas, exact
// field arithmetic => the same (X, Y, Z) triples as the reference after leaving Montgomery form.
#pragma once
#include "fe_mont.hpp"

namespace ecgpu {

struct P256Params {
  using Mod = P256Mod;
  static constexpr int ID = 1;
  // Montgomery forms of EQUATION_B and GENERATOR (p256/src/arithmetic.rs:37-59), ORDER (p256/src/lib.rs:74-108)
  static constexpr u32 B[8] = {0x29C4BDDFu, 0xD89CDF62u, 0x78843090u, 0xACF005CDu, 0xF7212ED6u, 0xE5A220ABu, 0x04874834u, 0xDC30061Du};
  static constexpr u32 GX[8] = {0x18A9143Cu, 0x79E730D4u, 0x5FEDB601u, 0x75BA95FCu, 0x77622510u, 0x79FB732Bu, 0xA53755C6u, 0x18905F76u};
  static constexpr u32 GY[8] = {0xCE95560Au, 0xDDF25357u, 0xBA19E45Cu, 0x8B4AB8E4u, 0xDD21F325u, 0xD2E88688u, 0x25885D85u, 0x8571FF18u};
  static constexpr u32 ORDER[8] = {0xFC632551u, 0xF3B9CAC2u, 0xA7179E84u, 0xBCE6FAADu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0xFFFFFFFFu};
};
struct P384Params {
  using Mod = P384Mod;
  static constexpr int ID = 2;
  // p384/src/arithmetic.rs:36-61, p384/src/lib.rs:50-64
  static constexpr u32 B[12] = {0x9D412DCCu, 0x08118871u, 0x7A4C32ECu, 0xF729ADD8u, 0x1920022Eu, 0x77F2209Bu,
                                0x94938AE2u, 0xE3374BEEu, 0x1F022094u, 0xB62B21F4u, 0x604FBFF9u, 0xCD08114Bu};
  static constexpr u32 GX[12] = {0x49C0B528u, 0x3DD07566u, 0xA0D6CE38u, 0x20E378E2u, 0x541B4D6Eu, 0x879C3AFCu,
                                 0x59A30EFFu, 0x64548684u, 0x614EDE2Bu, 0x812FF723u, 0x299E1513u, 0x4D3AADC2u};
  static constexpr u32 GY[12] = {0x4B03A4FEu, 0x23043DADu, 0x7BB4A9ACu, 0xA1BFA8BFu, 0x2E83B050u, 0x8BADE756u,
                                 0x68F4FFD9u, 0xC6C35219u, 0x3969A840u, 0xDD800226u, 0x5A15C5E9u, 0x2B78ABC2u};
  static constexpr u32 ORDER[12] = {0xCCC52973u, 0xECEC196Au, 0x48B0A77Au, 0x581A0DB2u, 0xF4372DDFu, 0xC7634D81u,
                                    0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
};

template <class P>
struct PtNist {   // homogeneous projective, identity (0 : 1 : 0)
  FeMont<typename P::Mod> x, y, z;
};
template <class P>
struct AfNist {   // AffinePoint::IDENTITY = (0, 0, infinity = 1), primeorder/src/affine.rs:48-52
  FeMont<typename P::Mod> x, y;
  u32 inf;
};

namespace nist {

template <class P> using Fe = FeMont<typename P::Mod>;

template <class P> ECGPU_HD void curve_b(Fe<P>& b) {
#pragma unroll
  for (int i = 0; i < P::Mod::N; i++) b.v[i] = P::B[i];
}
template <class P> ECGPU_HD void pt_identity(PtNist<P>& r) { mont::set_zero(r.x); mont::set_one(r.y); mont::set_zero(r.z); }
template <class P> ECGPU_HD void pt_generator(PtNist<P>& g) {
#pragma unroll
  for (int i = 0; i < P::Mod::N; i++) { g.x.v[i] = P::GX[i]; g.y.v[i] = P::GY[i]; }
  mont::set_one(g.z);
}
template <class P> ECGPU_HD void pt_select(PtNist<P>& r, bool c, const PtNist<P>& a, const PtNist<P>& b) {
  mont::select(r.x, c, a.x, b.x); mont::select(r.y, c, a.y, b.y); mont::select(r.z, c, a.z, b.z);
}

// primeorder/src/point_arithmetic.rs:209-238  (RCB Algorithm 4, a = -3): 12 M + 2 multiplications by b
template <class P>
ECGPU_HD void pt_add(PtNist<P>& r, const PtNist<P>& p, const PtNist<P>& q) {
  using namespace mont;
  Fe<P> B; curve_b<P>(B);
  Fe<P> xx, yy, zz, t0, t1, xy_pairs, yz_pairs, xz_pairs;
  mul(xx, p.x, q.x); mul(yy, p.y, q.y); mul(zz, p.z, q.z);
  add(t0, p.x, p.y); add(t1, q.x, q.y); mul(xy_pairs, t0, t1); add(t0, xx, yy); sub(xy_pairs, xy_pairs, t0);
  add(t0, p.y, p.z); add(t1, q.y, q.z); mul(yz_pairs, t0, t1); add(t0, yy, zz); sub(yz_pairs, yz_pairs, t0);
  add(t0, p.x, p.z); add(t1, q.x, q.z); mul(xz_pairs, t0, t1); add(t0, xx, zz); sub(xz_pairs, xz_pairs, t0);
  Fe<P> bzz_part, bzz3_part, yy_m_bzz3, yy_p_bzz3, zz3, bxz_part, bxz3_part, xx3_m_zz3;
  mul(t0, B, zz); sub(bzz_part, xz_pairs, t0);
  dbl(t0, bzz_part); add(bzz3_part, t0, bzz_part);
  sub(yy_m_bzz3, yy, bzz3_part); add(yy_p_bzz3, yy, bzz3_part);
  dbl(t0, zz); add(zz3, t0, zz);
  mul(t0, B, xz_pairs); add(t1, zz3, xx); sub(bxz_part, t0, t1);
  dbl(t0, bxz_part); add(bxz3_part, t0, bxz_part);
  dbl(t0, xx); add(t0, t0, xx); sub(xx3_m_zz3, t0, zz3);
  mul(t0, yy_p_bzz3, xy_pairs); mul(t1, yz_pairs, bxz3_part); sub(r.x, t0, t1);
  mul(t0, yy_p_bzz3, yy_m_bzz3); mul(t1, xx3_m_zz3, bxz3_part); add(r.y, t0, t1);
  mul(t0, yy_m_bzz3, yz_pairs); mul(t1, xy_pairs, xx3_m_zz3); add(r.z, t0, t1);
}

// primeorder/src/point_arithmetic.rs:247-277  (RCB Algorithm 5)
template <class P>
ECGPU_HD void pt_add_mixed(PtNist<P>& r, const PtNist<P>& p, const AfNist<P>& q) {
  using namespace mont;
  Fe<P> B; curve_b<P>(B);
  Fe<P> xx, yy, t0, t1, xy_pairs, yz_pairs, xz_pairs;
  mul(xx, p.x, q.x); mul(yy, p.y, q.y);
  add(t0, p.x, p.y); add(t1, q.x, q.y); mul(xy_pairs, t0, t1); add(t0, xx, yy); sub(xy_pairs, xy_pairs, t0);
  mul(t0, q.y, p.z); add(yz_pairs, t0, p.y);
  mul(t0, q.x, p.z); add(xz_pairs, t0, p.x);
  Fe<P> bz_part, bz3_part, yy_m_bzz3, yy_p_bzz3, z3, bxz_part, bxz3_part, xx3_m_zz3;
  mul(t0, B, p.z); sub(bz_part, xz_pairs, t0);
  dbl(t0, bz_part); add(bz3_part, t0, bz_part);
  sub(yy_m_bzz3, yy, bz3_part); add(yy_p_bzz3, yy, bz3_part);
  dbl(t0, p.z); add(z3, t0, p.z);
  mul(t0, B, xz_pairs); add(t1, z3, xx); sub(bxz_part, t0, t1);
  dbl(t0, bxz_part); add(bxz3_part, t0, bxz_part);
  dbl(t0, xx); add(t0, t0, xx); sub(xx3_m_zz3, t0, z3);
  PtNist<P> s;
  mul(t0, yy_p_bzz3, xy_pairs); mul(t1, yz_pairs, bxz3_part); sub(s.x, t0, t1);
  mul(t0, yy_p_bzz3, yy_m_bzz3); mul(t1, xx3_m_zz3, bxz3_part); add(s.y, t0, t1);
  mul(t0, yy_m_bzz3, yz_pairs); mul(t1, xy_pairs, xx3_m_zz3); add(s.z, t0, t1);
  pt_select(r, q.inf != 0, p, s);   // :275 conditional_assign(lhs, rhs.is_identity())
}

// primeorder/src/point_arithmetic.rs:286-317  (RCB Algorithm 6)
template <class P>
ECGPU_HD void pt_double(PtNist<P>& r, const PtNist<P>& p) {
  using namespace mont;
  Fe<P> B; curve_b<P>(B);
  Fe<P> xx, yy, zz, xy2, xz2, t0, t1;
  sqr(xx, p.x); sqr(yy, p.y); sqr(zz, p.z);
  mul(t0, p.x, p.y); dbl(xy2, t0);
  mul(t0, p.x, p.z); dbl(xz2, t0);
  Fe<P> bzz_part, bzz3_part, yy_m_bzz3, yy_p_bzz3, y_frag, x_frag, zz3, bxz2_part, bxz6_part, xx3_m_zz3, yz2;
  mul(t0, B, zz); sub(bzz_part, t0, xz2);
  dbl(t0, bzz_part); add(bzz3_part, t0, bzz_part);
  sub(yy_m_bzz3, yy, bzz3_part); add(yy_p_bzz3, yy, bzz3_part);
  mul(y_frag, yy_p_bzz3, yy_m_bzz3); mul(x_frag, yy_m_bzz3, xy2);
  dbl(t0, zz); add(zz3, t0, zz);
  mul(t0, B, xz2); add(t1, zz3, xx); sub(bxz2_part, t0, t1);
  dbl(t0, bxz2_part); add(bxz6_part, t0, bxz2_part);
  dbl(t0, xx); add(t0, t0, xx); sub(xx3_m_zz3, t0, zz3);
  PtNist<P> s;
  mul(t0, xx3_m_zz3, bxz6_part); add(s.y, y_frag, t0);
  mul(t0, p.y, p.z); dbl(yz2, t0);
  mul(t0, bxz6_part, yz2); sub(s.x, x_frag, t0);
  mul(t0, yz2, yy); dbl(t0, t0); dbl(s.z, t0);
  r = s;
}

// x^3 - 3x + b
template <class P>
ECGPU_HD void curve_rhs(Fe<P>& r, const Fe<P>& x) {
  using namespace mont;
  Fe<P> B, t; curve_b<P>(B);
  sqr(r, x); mul(r, r, x);
  dbl(t, x); add(t, t, x);
  sub(r, r, t); add(r, r, B);
}

// ProjectivePoint::mul (primeorder/src/projective.rs:106-150).  k: canonical scalar, little-endian
// 32-bit limbs; pc: scratch for the 16-entry table.  The window's entry is picked by the reference's constant-time
// scan (:132-137): entries 1..15 are all read and merged under an arithmetic mask, so neither the addresses touched
// nor the instructions executed depend on the scalar.
template <class P>
ECGPU_HD void mul_ref(PtNist<P>& q, const PtNist<P>& p, const u32* k, PtNist<P>* pc) {
  constexpr int N = P::Mod::N;
  pt_identity<P>(pc[0]);
  pc[1] = p;
#pragma unroll 1
  for (int i = 2; i < 16; i++) {
    if ((i & 1) == 0) pt_double<P>(pc[i], pc[i >> 1]);
    else pt_add<P>(pc[i], pc[i - 1], p);
  }
  pt_identity<P>(q);
#pragma unroll 1
  for (int pos = 32 * N - 4; pos >= 0; pos -= 4) {
    u32 w = k[0];
#pragma unroll
    for (int j = 1; j < N; j++) w = (pos >> 5) == j ? k[j] : w;
    const u32 slot = (w >> (pos & 31)) & 0xFu;
    PtNist<P> t;
    pt_identity<P>(t);
#pragma unroll 1
    for (u32 i = 1; i < 16; i++) {
      const u32 m = 0u - (((slot ^ i) - 1u) >> 31);          // all ones iff slot == i
      const PtNist<P>& c = pc[i];
      ECGPU_TABLE_TOUCH(i);
#pragma unroll
      for (int w = 0; w < N; w++) {
        t.x.v[w] = (t.x.v[w] & ~m) | (c.x.v[w] & m);
        t.y.v[w] = (t.y.v[w] & ~m) | (c.y.v[w] & m);
        t.z.v[w] = (t.z.v[w] & ~m) | (c.z.v[w] & m);
      }
    }
    pt_add<P>(q, q, t);
    if (pos != 0) {
#pragma unroll 1
      for (int j = 0; j < 4; j++) pt_double<P>(q, q);
    }
  }
}

}  // namespace nist
}  // namespace ecgpu
