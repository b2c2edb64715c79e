// Curve traits: one struct per curve exposing the same static interface, so that the batch
// kernels (kernels.hip) are written once.  The names follow the reference's trait surface
// (FieldElement ops, ProjectivePoint::{add, add_mixed, double}, Mul<Scalar>, MulByGenerator).
#pragma once
#include "mul_k256.hpp"
#include "curve_nist.hpp"

namespace ecgpu {

template <int NW>
ECGPU_HD void words_load_be(u32* limbs, const u32* be) {   // big-endian byte string -> LE limbs
#pragma unroll
  for (int i = 0; i < NW; i++) limbs[i] = bswap32(be[NW - 1 - i]);
}
template <int NW>
ECGPU_HD void words_store_be(u32* be, const u32* limbs) {
#pragma unroll
  for (int i = 0; i < NW; i++) be[NW - 1 - i] = bswap32(limbs[i]);
}

struct CurveK256 {
  static constexpr int ID = 0;
  static constexpr int NW = 8;           // 32-bit words per field element / scalar
  static constexpr int NB = 32;          // bytes
  static constexpr int REF_TABLE_PTS = 16;   // per-term scratch of the reference-faithful mul
  static constexpr int GEN_TABLE_PTS = 33 * 8;
  static constexpr bool A_IS_ZERO = true;       // y^2 = x^3 + 7
  using Fe = FeK256;
  using Pt = PtK256;
  using Af = AfK256;

  static ECGPU_HD void fe_load(Fe& r, const u32* be) { words_load_be<8>(r.v, be); }
  static ECGPU_HD void fe_store(u32* be, const Fe& a) { Fe n; k256::normalize(n, a); words_store_be<8>(be, n.v); }
  static ECGPU_HD bool fe_is_canonical(const Fe& a) { return k256::is_canonical(a); }
  static ECGPU_HD void fe_mul(Fe& r, const Fe& a, const Fe& b) { k256::mul(r, a, b); }
  static ECGPU_HD void fe_sqr(Fe& r, const Fe& a) { k256::sqr(r, a); }
  // r = a b - e f: both products on one set of column accumulators, one reduction (fe_k256.hpp::mul_add2); r may alias any operand
  static ECGPU_HD void fe_mul_sub2(Fe& r, const Fe& a, const Fe& b, const Fe& e, const Fe& f) { Fe n; k256::neg(n, e); k256::mul_add2(r, a, b, n, f); }
  static ECGPU_HD void fe_add(Fe& r, const Fe& a, const Fe& b) { k256::add(r, a, b); }
  static ECGPU_HD void fe_sub(Fe& r, const Fe& a, const Fe& b) { k256::sub(r, a, b); }
  static ECGPU_HD void fe_neg(Fe& r, const Fe& a) { k256::neg(r, a); }
  static ECGPU_HD void fe_inv(Fe& r, const Fe& a) { k256::inv(r, a); }
  static ECGPU_HD bool fe_sqrt(Fe& r, const Fe& a) { return k256::sqrt(r, a); }
  static ECGPU_HD bool fe_is_zero(const Fe& a) { return k256::is_zero(a); }
  static ECGPU_HD bool fe_is_zero_fast(const Fe& a) { return k256::is_zero_fast(a); }
  static ECGPU_HD bool fe_is_odd(const Fe& a) { return k256::is_odd(a); }
  static ECGPU_HD void fe_zero(Fe& r) { k256::set_zero(r); }
  static ECGPU_HD void fe_one(Fe& r) { k256::set_one(r); }
  static ECGPU_HD void fe_select(Fe& r, bool c, const Fe& a, const Fe& b) { k256::select(r, c, a, b); }
  // x^3 + a x + b
  static ECGPU_HD void curve_rhs(Fe& r, const Fe& x) {
    Fe b; k256::set_u32(b, 7);
    k256::sqr(r, x); k256::mul(r, r, x); k256::add(r, r, b);
  }

  static ECGPU_HD void pt_identity(Pt& r) { k256::pt_identity(r); }
  static ECGPU_HD void pt_add(Pt& r, const Pt& p, const Pt& q) { k256::pt_add(r, p, q); }
  static ECGPU_HD void pt_add_mixed(Pt& r, const Pt& p, const Af& q) { k256::pt_add_mixed(r, p, q); }
  static ECGPU_HD void pt_double(Pt& r, const Pt& p) { k256::pt_double(r, p); }
  static ECGPU_HD void pt_generator(Pt& g) { k256::generator(g); }

  static ECGPU_HD void scalar_load(u32* k, const u32* be) { words_load_be<8>(k, be); }
  static ECGPU_HD void order(u32* n) { k256::order(n); }
  static ECGPU_HD void modulus(u32* p) {
    p[0] = 0xFFFFFC2Fu; p[1] = 0xFFFFFFFEu;
#pragma unroll
    for (int i = 2; i < 8; i++) p[i] = 0xFFFFFFFFu;
  }

  // `&P * &k` exactly as the reference computes it (exact X, Y, Z)
  static ECGPU_HD void mul_ref(Pt& r, const Pt& p, const u32* k, Pt* tab) { k256::mul_ref(r, p, k, tab); }
  template <int NT>
  static ECGPU_HD void lincomb_ref(Pt& r, const Pt* p, const u32 (*k)[8], Pt* tab) { k256::lincomb_ref<NT>(r, p, k, tab); }
  static ECGPU_HD void gen_table_build(Pt* tab) { Pt g; k256::generator(g); k256::gen_table_build(tab, g); }
  static ECGPU_HD void mul_gen_ref(Pt& r, const u32* k, const Pt* gen_tab, Pt*) { k256::mul_gen_ref(r, k, gen_tab); }
};


// NIST curves through the primeorder layer: Montgomery field, RCB a = -3, 4-bit window.
template <class P>
struct CurveNist {
  static constexpr int ID = P::ID;
  static constexpr int NW = P::Mod::N;
  static constexpr int NB = 4 * NW;
  static constexpr int REF_TABLE_PTS = 16;
  static constexpr int GEN_TABLE_PTS = 1;       // mul_by_generator is G * k (primeorder/src/projective.rs:422-431)
  static constexpr bool A_IS_ZERO = false;      // a = -3
  using Mod = typename P::Mod;
  using Fe = FeMont<Mod>;
  using Pt = PtNist<P>;
  using Af = AfNist<P>;

  static ECGPU_HD void fe_load(Fe& r, const u32* be) { u32 c[NW]; words_load_be<NW>(c, be); mont::to_mont<Mod>(r, c); }
  static ECGPU_HD void fe_store(u32* be, const Fe& a) { u32 c[NW]; mont::from_mont<Mod>(c, a); words_store_be<NW>(be, c); }
  static ECGPU_HD void fe_mul(Fe& r, const Fe& a, const Fe& b) { mont::mul(r, a, b); }
  static ECGPU_HD void fe_sqr(Fe& r, const Fe& a) { mont::sqr(r, a); }
  // r = a b - e f (two Montgomery multiplications: a fused form would need a second final subtraction and saves too little here); r may alias any operand
  static ECGPU_HD void fe_mul_sub2(Fe& r, const Fe& a, const Fe& b, const Fe& e, const Fe& f) { Fe t, u; mont::mul(t, a, b); mont::mul(u, e, f); mont::sub(r, t, u); }
  static ECGPU_HD void fe_add(Fe& r, const Fe& a, const Fe& b) { mont::add(r, a, b); }
  static ECGPU_HD void fe_sub(Fe& r, const Fe& a, const Fe& b) { mont::sub(r, a, b); }
  static ECGPU_HD void fe_neg(Fe& r, const Fe& a) { mont::neg(r, a); }
  static ECGPU_HD void fe_half(Fe& r, const Fe& a) { mont::half(r, a); }
  static ECGPU_HD void fe_inv(Fe& r, const Fe& a) { mont::inv(r, a); }
  static ECGPU_HD bool fe_sqrt(Fe& r, const Fe& a) { return mont::sqrt(r, a); }
  static ECGPU_HD bool fe_is_zero(const Fe& a) { return mont::is_zero(a); }
  static ECGPU_HD bool fe_is_zero_fast(const Fe& a) { return mont::is_zero_fast(a); }
  static ECGPU_HD bool fe_is_odd(const Fe& a) { u32 c[NW]; mont::from_mont<Mod>(c, a); return c[0] & 1; }   // p256 field.rs:109-112
  static ECGPU_HD void fe_zero(Fe& r) { mont::set_zero(r); }
  static ECGPU_HD void fe_one(Fe& r) { mont::set_one(r); }
  static ECGPU_HD void fe_select(Fe& r, bool c, const Fe& a, const Fe& b) { mont::select(r, c, a, b); }
  static ECGPU_HD void curve_rhs(Fe& r, const Fe& x) { nist::curve_rhs<P>(r, x); }

  static ECGPU_HD void pt_identity(Pt& r) { nist::pt_identity<P>(r); }
  static ECGPU_HD void pt_add(Pt& r, const Pt& p, const Pt& q) { nist::pt_add<P>(r, p, q); }
  static ECGPU_HD void pt_add_mixed(Pt& r, const Pt& p, const Af& q) { nist::pt_add_mixed<P>(r, p, q); }
  static ECGPU_HD void pt_double(Pt& r, const Pt& p) { nist::pt_double<P>(r, p); }
  static ECGPU_HD void pt_generator(Pt& g) { nist::pt_generator<P>(g); }

  static ECGPU_HD void scalar_load(u32* k, const u32* be) { words_load_be<NW>(k, be); }
  static ECGPU_HD void order(u32* n) {
#pragma unroll
    for (int i = 0; i < NW; i++) n[i] = P::ORDER[i];
  }
  static ECGPU_HD void modulus(u32* p) {
#pragma unroll
    for (int i = 0; i < NW; i++) p[i] = Mod::P[i];
  }

  static ECGPU_HD void mul_ref(Pt& r, const Pt& p, const u32* k, Pt* tab) { nist::mul_ref<P>(r, p, k, tab); }
  // LinearCombination default: x*k + y*l (primeorder/src/projective.rs:415-420)
  template <int NT>
  static ECGPU_HD void lincomb_ref(Pt& r, const Pt* p, const u32 (*k)[NW], Pt* tab) {
    nist::mul_ref<P>(r, p[0], k[0], tab);
#pragma unroll 1
    for (int t = 1; t < NT; t++) {
      Pt s;
      nist::mul_ref<P>(s, p[t], k[t], tab);
      nist::pt_add<P>(r, r, s);
    }
  }
  static ECGPU_HD void gen_table_build(Pt* tab) { nist::pt_generator<P>(tab[0]); }
  static ECGPU_HD void mul_gen_ref(Pt& r, const u32* k, const Pt* gen_tab, Pt* tab) { nist::mul_ref<P>(r, gen_tab[0], k, tab); }
};
// point formats of the C ABI (ecgpu_point_format)
enum { FMT_AFFINE = 0, FMT_PROJECTIVE = 1 };

// v -= m if v >= m (Reduce<U256>::reduce, k256 scalar.rs:700-713)
template <int NW>
ECGPU_HD void reduce_once(u32* v, const u32* m) {
  u32 t[NW];
  const u32 bw = mp_sub<NW>(t, v, m);
#pragma unroll
  for (int i = 0; i < NW; i++) v[i] = bw ? v[i] : t[i];
}

using CurveP256 = CurveNist<P256Params>;
using CurveP384 = CurveNist<P384Params>;

}  // namespace ecgpu
