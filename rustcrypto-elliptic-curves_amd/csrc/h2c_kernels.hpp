// Hash to curve, the field-arithmetic half of GroupDigest::hash_from_bytes:
// map_to_curve = simplified SWU for p = 3 (mod 4) in the straight-line form of
// k256/src/arithmetic/hash2curve.rs:100-143 (the external crate's generic osswu gives the same point, see
// tools/gen_h2c_constants.py), followed for secp256k1 by the 3-isogeny E' -> E (hash2curve.rs:186-262), and the sum
// Q0 + Q1 of the two mapped points (cofactor 1).  expand_message_xmd / hash_to_field are host glue
// (ecgpu/hash2curve.py).  Device code only; not a tuned path (three exponentiations per mapped element).
#pragma once
#include "kernels.hpp"

namespace ecgpu {

template <int ID> struct H2cParams;
#include "h2c_constants.inc"

namespace h2c {

// canonical little-endian words -> the curve's field form
template <class C>
__device__ __forceinline__ void konst(typename C::Fe& r, const u32* w) {
  u32 be[C::NW];
  words_store_be<C::NW>(be, w);
  C::fe_load(r, be);
}
// r = a^e, e public (little-endian words), 4-bit fixed window
template <class C>
__device__ __forceinline__ void pow_window(typename C::Fe& r, const typename C::Fe& a, const u32* e) {
  using Fe = typename C::Fe;
  Fe tab[15];
  tab[0] = a;
#pragma unroll 1
  for (int i = 1; i < 15; i++) C::fe_mul(tab[i], tab[i - 1], a);
  Fe acc;
  C::fe_one(acc);
#pragma unroll 1
  for (int j = 8 * C::NW - 1; j >= 0; j--) {
#pragma unroll 1
    for (int s = 0; s < 4; s++) C::fe_sqr(acc, acc);
    u32 w = e[0];
#pragma unroll
    for (int q = 1; q < C::NW; q++) w = (j >> 3) == q ? e[q] : w;
    const u32 d = (w >> (4 * (j & 7))) & 15u;
    if (d) C::fe_mul(acc, acc, tab[d - 1]);
  }
  r = acc;
}
template <class C, int N>
__device__ __forceinline__ void horner(typename C::Fe& r, const u32 (*co)[C::NW], const typename C::Fe& x) {
  typename C::Fe k;
  konst<C>(r, co[N - 1]);
#pragma unroll 1
  for (int i = N - 2; i >= 0; i--) {
    C::fe_mul(r, r, x);
    konst<C>(k, co[i]);
    C::fe_add(r, r, k);
  }
}

template <class C>
__device__ __forceinline__ void map_to_curve(typename C::Fe& x, typename C::Fe& y, const typename C::Fe& u) {
  using Fe = typename C::Fe;
  using P = H2cParams<C::ID>;
  constexpr int NW = C::NW;
  Fe Z, A, B, c2, one, zero;
  konst<C>(Z, P::Z); konst<C>(A, P::A); konst<C>(B, P::B); konst<C>(c2, P::C2);
  C::fe_one(one); C::fe_zero(zero);
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
  u32 e[NW], pm[NW];
  C::modulus(pm);                                 // c1 = (p - 3) / 4
  u32 bw = 0;
  pm[0] = subb(pm[0], 3u, bw);
#pragma unroll
  for (int i = 1; i < NW; i++) pm[i] = subb(pm[i], 0u, bw);
#pragma unroll
  for (int i = 0; i < NW; i++) e[i] = (pm[i] >> 2) | (i + 1 < NW ? pm[i + 1] << 30 : 0u);
  pow_window<C>(y1, tv4, e);
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
  Fe xn, xi;
  C::fe_select(xn, e2, x1n, x2n);
  C::fe_inv(xi, xd);
  C::fe_mul(x, xn, xi);
  C::fe_select(y, e2, y1, y2);
  if (C::fe_is_odd(u) != C::fe_is_odd(y)) C::fe_neg(y, y);      // sgn0(u) == sgn0(y)
  if constexpr (C::ID == 0) {
    // 3-isogeny to secp256k1: x' = xnum(x) / xden(x), y' = y ynum(x) / yden(x); one inversion for both denominators
    Fe xn_, xd_, yn_, yd_, den, di;
    horner<C, P::N_XNUM>(xn_, P::XNUM, x);
    horner<C, P::N_XDEN>(xd_, P::XDEN, x);
    horner<C, P::N_YNUM>(yn_, P::YNUM, x);
    horner<C, P::N_YDEN>(yd_, P::YDEN, x);
    C::fe_mul(den, xd_, yd_);
    C::fe_inv(di, den);
    C::fe_mul(t, di, yd_);                        // 1 / xden
    C::fe_mul(x, xn_, t);
    C::fe_mul(t, di, xd_);                        // 1 / yden
    C::fe_mul(t, t, yn_);
    C::fe_mul(y, y, t);
  }
}

// u: n x count field elements (canonical big-endian, < p; count = 1 or 2).  out = map(u_0) [+ map(u_1)], affine.
template <class C>
__global__ void __launch_bounds__(256) map_kernel(const u32* u, int count, u32* out_xy, uint8_t* out_inf, size_t n) {
  using Fe = typename C::Fe;
  constexpr int NW = C::NW;
  ECGPU_GRID_STRIDE(i, n) {
    Fe uu, x0, y0;
    C::fe_load(uu, u + i * count * NW);
    map_to_curve<C>(x0, y0, uu);
    if (count == 1) {
      C::fe_store(out_xy + i * 2 * NW, x0);
      C::fe_store(out_xy + i * 2 * NW + NW, y0);
      if (out_inf) out_inf[i] = 0;
    } else {
      Fe x1, y1;
      C::fe_load(uu, u + (i * count + 1) * NW);
      map_to_curve<C>(x1, y1, uu);
      typename C::Pt p, q, r;
      p.x = x0; p.y = y0; C::fe_one(p.z);
      q.x = x1; q.y = y1; C::fe_one(q.z);
      C::pt_add(r, p, q);
      store_affine_from_projective<C>(out_xy + i * 2 * NW, out_inf ? out_inf + i : nullptr, r);
    }
  }
}

}  // namespace h2c
}  // namespace ecgpu
