// Jacobian-coordinate group law over the curve traits (x = X/Z^2, y = Y/Z^3, infinity <=> Z = 0),
// used by the throughput schedules of every curve.  The reference works in homogeneous projective
// coordinates with complete formulas (k256 projective.rs:96-274, primeorder point_arithmetic.rs:199-317);
// these are the cheaper incomplete formulas with their exceptional cases handled explicitly, so the
// group element that comes out is the same for every input.
#pragma once
#include "traits.hpp"

namespace ecgpu {

template <class C>
struct Jac {
  typename C::Fe x, y, z;
};
template <class C>
struct AffEntry {   // affine table entry in the curve's internal field form (no infinity: tables never hold it)
  typename C::Fe x, y;
};

// Where a lane keeps the recoded digits of the unit it is working on.  The window loops read ONE word of them per window,
// indexed by the (wave-uniform) loop counter; kept in registers that is NW VGPRs held across every field operation of
// the loop plus an NW-way select chain per read.  On the device the kernels hand in a column of a __shared__ array
// (word q of lane t at base[q * 256 + t]: consecutive lanes hit consecutive banks, conflict-free) - 160 KB of LDS per CU
// over 1 024 resident lanes is 40 dwords per lane, enough for the digits of one unit; the host twin passes a plain array.
struct DigitMem {
  u32* base;
  int stride;
  ECGPU_HD void st(int q, u32 v) const { base[q * stride] = v; }
  ECGPU_HD u32 ld(int q) const { return base[q * stride]; }
};

namespace jac {

template <class C> ECGPU_HD void set_infinity(Jac<C>& p) { C::fe_zero(p.x); C::fe_zero(p.y); C::fe_zero(p.z); }
template <class C> ECGPU_HD void fe_dbl(typename C::Fe& r, const typename C::Fe& a) { C::fe_add(r, a, a); }

// In-place doubling.  a = 0 (k256): 3M + 4S.  a = -3 (NIST): 4M + 4S (dbl-2004-hmv).
template <class C>
ECGPU_HD void dbl(Jac<C>& p) {
  using Fe = typename C::Fe;
  if constexpr (C::A_IS_ZERO) {
    Fe a, b, t;
    C::fe_sqr(a, p.x);
    C::fe_sqr(b, p.y);
    C::fe_mul(p.z, p.y, p.z); fe_dbl<C>(p.z, p.z);
    C::fe_mul(p.y, p.x, b); fe_dbl<C>(p.y, p.y); fe_dbl<C>(p.y, p.y);      // D
    C::fe_sqr(b, b);                                                        // C
    fe_dbl<C>(t, a); C::fe_add(a, t, a);                                    // E
    C::fe_sqr(t, a);
    C::fe_sub(t, t, p.y); C::fe_sub(p.x, t, p.y);
    C::fe_sub(p.y, p.y, p.x); C::fe_mul(p.y, a, p.y);
    fe_dbl<C>(b, b); fe_dbl<C>(b, b); fe_dbl<C>(b, b);
    C::fe_sub(p.y, p.y, b);
  } else {
    // dbl-2004-hmv, 4M + 4S with one halving: the additions of a fully reduced Montgomery field cost ~10 % of a
    // multiplication each, and this form needs 10 of them where dbl-2001-b (3M + 5S) needs 17.
    //   alpha = 3 (X - Z^2)(X + Z^2), Z3 = 2 Y Z, beta4 = 4 X Y^2, X3 = alpha^2 - 2 beta4,
    //   Y3 = alpha (beta4 - X3) - 8 Y^4   with 8 Y^4 = (2Y)^4 / 2
    Fe t1, t2, t3;
    C::fe_sqr(t1, p.z);
    C::fe_sub(t2, p.x, t1);
    C::fe_add(t1, p.x, t1);
    C::fe_mul(t2, t1, t2);
    fe_dbl<C>(t1, t2); C::fe_add(t2, t1, t2);            // alpha
    fe_dbl<C>(p.y, p.y);                                 // 2Y
    C::fe_mul(p.z, p.y, p.z);                            // Z3
    C::fe_sqr(p.y, p.y);                                 // 4 Y^2
    C::fe_mul(t3, p.y, p.x);                             // beta4
    C::fe_sqr(p.y, p.y);                                 // 16 Y^4
    C::fe_half(p.y, p.y);                                // 8 Y^4
    C::fe_sqr(p.x, t2);
    fe_dbl<C>(t1, t3);
    C::fe_sub(p.x, p.x, t1);                             // X3
    C::fe_sub(t1, t3, p.x);
    C::fe_mul(t1, t1, t2);
    C::fe_sub(p.y, t1, p.y);                             // Y3
  }
}

// In-place p += (x2, y2), (x2, y2) affine and not the identity.  8M + 3S, independent of the curve
// coefficients.  Special cases by control flow: p at infinity; same point (doubling); opposite points
// (Z3 = Z1 * 0 = 0 falls out of the formula).
template <class C>
ECGPU_HD void add_mixed(Jac<C>& p, const typename C::Fe& x2, const typename C::Fe& y2) {
  using Fe = typename C::Fe;
  if (C::fe_is_zero_fast(p.z)) {
    p.x = x2; p.y = y2; C::fe_one(p.z);
    return;
  }
  Fe h, r, t, u;
  C::fe_sqr(t, p.z);
  C::fe_mul(h, x2, t);
  C::fe_mul(t, p.z, t); C::fe_mul(r, t, y2);
  C::fe_sub(h, h, p.x);
  C::fe_sub(r, r, p.y);
  if (__builtin_expect(C::fe_is_zero_fast(h), 0)) {
    if (C::fe_is_zero(r)) {               // same point
      p.x = x2; p.y = y2; C::fe_one(p.z);
      dbl<C>(p);
    } else {                              // opposite points
      set_infinity<C>(p);
    }
    return;
  }
  C::fe_mul(p.z, p.z, h);
  C::fe_sqr(t, h);
  C::fe_mul(h, t, h);
  C::fe_mul(t, p.x, t);
  C::fe_sqr(u, r);
  C::fe_sub(u, u, h); C::fe_sub(u, u, t); C::fe_sub(p.x, u, t);
  C::fe_sub(t, t, p.x);
  C::fe_mul_sub2(p.y, r, t, p.y, h);     // Y3 = R (V - X3) - Y1 HHH
}

// In-place p += (x2, y2) for p = (X1, Y1, 1) AFFINE and finite (the caller knows: the first addition after a set), (x2, y2) affine
// and not the identity: 4M + 2S, the mixed addition with Z1 = 1.  Same special cases by control flow.
template <class C>
ECGPU_HD void add_affine(Jac<C>& p, const typename C::Fe& x2, const typename C::Fe& y2) {
  using Fe = typename C::Fe;
  Fe h, r, t, u;
  C::fe_sub(h, x2, p.x);
  C::fe_sub(r, y2, p.y);
  if (__builtin_expect(C::fe_is_zero_fast(h), 0)) {
    if (C::fe_is_zero(r)) dbl<C>(p);      // same point (Z = 1 already)
    else set_infinity<C>(p);              // opposite points
    return;
  }
  p.z = h;
  C::fe_sqr(t, h);
  C::fe_mul(h, t, h);
  C::fe_mul(t, p.x, t);
  C::fe_sqr(u, r);
  C::fe_sub(u, u, h); C::fe_sub(u, u, t); C::fe_sub(p.x, u, t);
  C::fe_sub(t, t, p.x);
  C::fe_mul_sub2(p.y, r, t, p.y, h);     // Y3 = R (V - X3) - Y1 HHH
}

// Co-Z arithmetic for table chains P, 2P, 3P = 2P + P, .. (Meloni's additions with update): two points that share their
// denominator Z add in 4M + 2S without touching Z, and the addend comes back rewritten to the sum's denominator Z h, so
// the chain never multiplies a Z out - only the ratios h are kept.
//
// Doubling with update, a = -3 (4M + 4S; 2M + 4S for an affine input, `z_is_one`): (dx, dy) = 2P over the denominator
// z2 = 2 Y Z, (qx, qy) = P rewritten to that denominator: (4 X Y^2, 8 Y^4).  P is finite and not of order 2.
template <class C>
ECGPU_HD void coz_double_update(typename C::Fe& dx, typename C::Fe& dy, typename C::Fe& z2, typename C::Fe& qx, typename C::Fe& qy, const Jac<C>& p,
                                bool z_is_one) {
  static_assert(!C::A_IS_ZERO, "a = -3 (secp256k1 has its own chain in mulfast_k256.hpp)");
  using Fe = typename C::Fe;
  Fe t1, t2, al, e;
  if (z_is_one) C::fe_one(t1); else C::fe_sqr(t1, p.z);
  C::fe_sub(t2, p.x, t1);
  C::fe_add(t1, p.x, t1);
  C::fe_mul(t2, t1, t2);
  fe_dbl<C>(t1, t2); C::fe_add(al, t1, t2);            // alpha = 3 (X - Z^2)(X + Z^2)
  fe_dbl<C>(z2, p.y);                                  // 2Y
  if (!z_is_one) C::fe_mul(z2, z2, p.z);               // Z2 = 2 Y Z
  C::fe_sqr(e, p.y);                                   // Y^2
  C::fe_mul(qx, p.x, e); fe_dbl<C>(qx, qx); fe_dbl<C>(qx, qx);      // S = 4 X Y^2
  C::fe_sqr(qy, e); fe_dbl<C>(qy, qy); fe_dbl<C>(qy, qy); fe_dbl<C>(qy, qy);   // 8 Y^4
  C::fe_sqr(t1, al);
  C::fe_sub(t1, t1, qx); C::fe_sub(dx, t1, qx);        // X = alpha^2 - 2S
  C::fe_sub(t1, qx, dx); C::fe_mul(t1, al, t1);
  C::fe_sub(dy, t1, qy);                               // Y = alpha (S - X) - 8 Y^4
}
// Co-Z addition with update (4M + 2S): (qx, qy) and (rx, ry) share the denominator Z.  r <- q + r over Z h,
// (qx, qy) <- q over Z h, h = qx - rx (of the inputs).  Exceptional iff q = +-r: the table chains (q = P, r = jP,
// 2 <= j <= 7, P of prime order) never meet it.
template <class C>
ECGPU_HD void coz_add_update(typename C::Fe& rx, typename C::Fe& ry, typename C::Fe& qx, typename C::Fe& qy, typename C::Fe& h) {
  using Fe = typename C::Fe;
  Fe c, w2, d, t;
  C::fe_sub(h, qx, rx);
  C::fe_sqr(c, h);                                     // C = (X1 - X2)^2
  C::fe_mul(qx, qx, c);                                // W1
  C::fe_mul(w2, rx, c);                                // W2
  C::fe_sub(d, qy, ry);                                // Y1 - Y2
  C::fe_sub(t, qx, w2); C::fe_mul(qy, qy, t);          // A1 = Y1 (W1 - W2)
  C::fe_sqr(t, d);
  C::fe_sub(t, t, qx); C::fe_sub(rx, t, w2);           // X3 = D - W1 - W2
  C::fe_sub(t, qx, rx); C::fe_mul(t, d, t);
  C::fe_sub(ry, t, qy);                                // Y3 = (Y1 - Y2)(W1 - X3) - A1
}

// general addition r = p + q (11M + 5S), all special cases handled
template <class C>
ECGPU_HD void add(Jac<C>& r, const Jac<C>& p, const Jac<C>& q) {
  using Fe = typename C::Fe;
  if (C::fe_is_zero_fast(p.z)) { r = q; return; }
  if (C::fe_is_zero_fast(q.z)) { r = p; return; }
  Fe z1z1, z2z2, u1, u2, s1, s2, h, rr, t;
  C::fe_sqr(z1z1, p.z); C::fe_sqr(z2z2, q.z);
  C::fe_mul(u1, p.x, z2z2); C::fe_mul(u2, q.x, z1z1);
  C::fe_mul(t, q.z, z2z2); C::fe_mul(s1, p.y, t);
  C::fe_mul(t, p.z, z1z1); C::fe_mul(s2, q.y, t);
  C::fe_sub(h, u2, u1);
  C::fe_sub(rr, s2, s1);
  if (C::fe_is_zero_fast(h)) {
    if (C::fe_is_zero(rr)) { r = p; dbl<C>(r); return; }
    set_infinity<C>(r);
    return;
  }
  Fe hh, hhh, v;
  C::fe_sqr(hh, h); C::fe_mul(hhh, hh, h); C::fe_mul(v, u1, hh);
  Jac<C> o;
  C::fe_sqr(t, rr); C::fe_sub(t, t, hhh); C::fe_sub(t, t, v); C::fe_sub(o.x, t, v);
  C::fe_sub(t, v, o.x); C::fe_mul(t, rr, t);
  C::fe_mul(s1, s1, hhh); C::fe_sub(o.y, t, s1);
  C::fe_mul(t, p.z, q.z); C::fe_mul(o.z, t, h);
  r = o;
}

// Montgomery's trick over the cnt results of one lane: out x, y in internal form (identity -> zeros),
// inf flags.  `pre` is scratch for cnt field elements.
template <class C>
ECGPU_HD void batch_to_affine(typename C::Fe* ax, typename C::Fe* ay, u32* inf, const Jac<C>* pts, int cnt, typename C::Fe* pre) {
  using Fe = typename C::Fe;
  Fe acc; C::fe_one(acc);
#pragma unroll 1
  for (int i = 0; i < cnt; i++) {
    pre[i] = acc;
    Fe z = pts[i].z;
    if (C::fe_is_zero(z)) C::fe_one(z);
    C::fe_mul(acc, acc, z);
  }
  Fe ai;
  C::fe_inv(ai, acc);
#pragma unroll 1
  for (int i = cnt - 1; i >= 0; i--) {
    Fe z = pts[i].z;
    const bool zr = C::fe_is_zero(z);
    if (zr) C::fe_one(z);
    Fe zi, t;
    C::fe_mul(zi, ai, pre[i]);
    C::fe_mul(ai, ai, z);
    C::fe_sqr(t, zi);
    C::fe_mul(ax[i], pts[i].x, t);
    C::fe_mul(t, t, zi);
    C::fe_mul(ay[i], pts[i].y, t);
    if (zr) { C::fe_zero(ax[i]); C::fe_zero(ay[i]); }
    inf[i] = zr ? 1u : 0u;
  }
}

// The shared epilogue of the throughput kernels (BatchNormalize, k256 projective.rs:325-379 / primeorder
// projective.rs:346-413, applied per lane): the cnt Jacobian results of one lane - element j is global index
// base + j * stride - go to affine with ONE inversion and are written in the caller's format: affine x || y
// (+ infinity byte, zeros for the identity) or the homogeneous representative (x : y : 1), identity (0 : 1 : 0).
// J is any {x, y, z} of C::Fe; `pre` is scratch for cnt field elements.
// `idx` (optional): the global index of element j, for callers whose results do not sit at base + j * stride (the dynamically scheduled kernels)
template <class C, class J>
ECGPU_HD void store_batch_affine(const J* res, typename C::Fe* pre, int cnt, size_t base, size_t stride, u32* out, int out_fmt,
                                 uint8_t* out_inf, const size_t* idx = nullptr) {
  using Fe = typename C::Fe;
  constexpr int NW = C::NW;
  Fe acc; C::fe_one(acc);
#pragma unroll 1
  for (int j = 0; j < cnt; j++) {
    pre[j] = acc;
    Fe z = res[j].z;
    if (C::fe_is_zero(z)) C::fe_one(z);
    C::fe_mul(acc, acc, z);
  }
  Fe ai;
  C::fe_inv(ai, acc);
#pragma unroll 1
  for (int j = cnt - 1; j >= 0; j--) {
    const size_t i = idx ? idx[j] : base + (size_t)j * stride;
    Fe z = res[j].z, one, zero, zi, t, x, y;
    C::fe_one(one); C::fe_zero(zero);
    const bool zr = C::fe_is_zero(z);
    if (zr) z = one;
    C::fe_mul(zi, ai, pre[j]);
    C::fe_mul(ai, ai, z);
    C::fe_sqr(t, zi);
    C::fe_mul(x, res[j].x, t);
    C::fe_mul(t, t, zi);
    C::fe_mul(y, res[j].y, t);
    if (zr) { x = zero; y = zero; }
    if (out_fmt == FMT_PROJECTIVE) {
      if (zr) y = one;
      u32* o = out + i * 3 * NW;
      C::fe_store(o, x); C::fe_store(o + NW, y); C::fe_store(o + 2 * NW, zr ? zero : one);
    } else {
      u32* o = out + i * 2 * NW;
      C::fe_store(o, x); C::fe_store(o + NW, y);
      if (out_inf) out_inf[i] = zr ? 1 : 0;
    }
  }
}

}  // namespace jac
}  // namespace ecgpu
