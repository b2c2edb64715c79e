// Batched ECDSA around the scalar-multiplication kernels.
//
// The reference enters the external `ecdsa` crate's hazmat::{verify_prehashed, sign_prehashed} from
// k256/src/ecdsa.rs:182-209, p256/src/ecdsa.rs:72-75 and p384/src/ecdsa.rs:69-72:
//   verify:  w = s^-1, u1 = z w, u2 = r w (mod n);  R = u1 G + u2 Q;  accept iff x(R) mod n == r
//   sign:    R = k G;  r = x(R) mod n;  s = k^-1 (z + r d) mod n
// Pipeline here, every stage one kernel over device memory:
//   verify:  verify_prep (range checks, public-key check, one scalar inversion per 16 signatures, u1, u2)
//            -> fixed-base kernel (u1 G, fixedbase.hpp) -> variable-base kernel (u2 Q, the headline kernel)
//            -> verify_check (x(A + B) mod n == r, decided without an inversion)
//   sign:    fixed-base kernel (k G) -> sign_finish (r, batched k^-1, s, recovery id)
// Device code only.
#pragma once
#include "kernels.hpp"
#include "scalar_mont.hpp"

namespace ecgpu {

template <int ID> struct OrderById;
template <> struct OrderById<0> { using T = K256Order; };
template <> struct OrderById<1> { using T = P256Order; };
template <> struct OrderById<2> { using T = P384Order; };
template <class C> using OrderOf = typename OrderById<C::ID>::T;

enum { ECDSA_LOW_S = 2 };    // ECGPU_ECDSA_LOW_S

namespace ecdsa {

template <int L>
__device__ __forceinline__ void load_be(u32* limbs, const u32* be) { words_load_be<L>(limbs, be); }
template <int L>
__device__ __forceinline__ void store_be(u32* be, const u32* limbs) { words_store_be<L>(be, limbs); }

// 0 < x < n
template <class O>
__device__ __forceinline__ bool in_range(const u32* x) {
  u32 n[O::L];
  smont::order<O>(n);
  return !mp_is_zero<O::L>(x) && !mp_geq<O::L>(x, n);
}
template <class O>
__device__ __forceinline__ bool is_high(const u32* x) {      // x > (n - 1) / 2   (k256 scalar.rs:519-523)
  u32 h[O::L];
#pragma unroll
  for (int i = 0; i < O::L; i++) h[i] = O::HALF[i];
  return !mp_geq<O::L>(h, x);
}

// on the curve, coordinates canonical, not the identity encoding (VerifyingKey::from_encoded_point ->
// PublicKey::from_affine rejects the identity)
template <class C>
__device__ __forceinline__ bool public_key_ok(const u32* xy) {
  constexpr int NW = C::NW;
  u32 lx[NW], ly[NW], p[NW];
  words_load_be<NW>(lx, xy);
  words_load_be<NW>(ly, xy + NW);
  C::modulus(p);
  if (mp_geq<NW>(lx, p) || mp_geq<NW>(ly, p)) return false;
  if (mp_is_zero<NW>(lx) && mp_is_zero<NW>(ly)) return false;
  typename C::Fe x, y, l, r, d;
  C::fe_load(x, xy);
  C::fe_load(y, xy + NW);
  C::fe_sqr(l, y);
  C::curve_rhs(r, x);
  C::fe_sub(d, l, r);
  return C::fe_is_zero(d);
}

// Batched inversion of BATCH scalars held in Montgomery form (Montgomery's trick), in place.
template <class O, int BATCH>
__device__ __forceinline__ void batch_invert(u32 (*v)[O::L], int cnt) {
  constexpr int L = O::L;
  u32 pre[BATCH][L], acc[L];
#pragma unroll
  for (int i = 0; i < L; i++) acc[i] = O::ONE[i];
#pragma unroll 1
  for (int b = 0; b < cnt; b++) {
    mp_copy<L>(pre[b], acc);
    smont::mul<O>(acc, acc, v[b]);
  }
  u32 ai[L];
  smont::inv<O>(ai, acc);
#pragma unroll 1
  for (int b = cnt - 1; b >= 0; b--) {
    u32 t[L];
    smont::mul<O>(t, ai, pre[b]);
    smont::mul<O>(ai, ai, v[b]);
    mp_copy<L>(v[b], t);
  }
}

// verify, stage 1.  z: prehash after bits2field (NB bytes each); sig: r || s; q: public keys x || y.
// Writes u1, u2 (canonical big-endian scalars) and ok (0 = already rejected).
template <class C, int BATCH>
__global__ void __launch_bounds__(256) verify_prep_kernel(const u32* z, const u32* sig, const u32* q, u32* u1, u32* u2, uint8_t* ok, size_t n,
                                                          unsigned flags) {
  using O = OrderOf<C>;
  constexpr int L = O::L;
  u32 w[BATCH][L];
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * BATCH) {
    int cnt = 0;
    u32 good = 0;
#pragma unroll 1
    for (int b = 0; b < BATCH; b++) {
      const size_t i = base + (size_t)b * T;
      if (i >= n) break;
      u32 r[L], s[L];
      load_be<L>(r, sig + i * 2 * L);
      load_be<L>(s, sig + i * 2 * L + L);
      bool g = in_range<O>(r) && in_range<O>(s);
      if ((flags & ECDSA_LOW_S) && is_high<O>(s)) g = false;
      if (!public_key_ok<C>(q + i * 2 * L)) g = false;
      if (!g) { mp_zero<L>(s); s[0] = 1; }
      smont::to_mont<O>(w[b], s);
      good |= (g ? 1u : 0u) << b;
      cnt = b + 1;
    }
    batch_invert<O, BATCH>(w, cnt);
#pragma unroll 1
    for (int b = 0; b < cnt; b++) {
      const size_t i = base + (size_t)b * T;
      u32 e[L], r[L], a[L], c[L];
      load_be<L>(e, z + i * L);
      smont::reduce_once<O>(e);                 // Reduce::reduce_bytes: the prehash is < 2^(32 L) < 2n
      load_be<L>(r, sig + i * 2 * L);
      const bool g = (good >> b) & 1u;
      if (!g) { mp_zero<L>(e); mp_zero<L>(r); }
      smont::mul<O>(a, e, w[b]);                // plain * Montgomery -> plain
      smont::mul<O>(c, r, w[b]);
      store_be<L>(u1 + i * L, a);
      store_be<L>(u2 + i * L, c);
      ok[i] = g ? 1 : 0;
    }
  }
}

// verify, last stage.  A = u1 G and B = u2 Q arrive affine (zeros + flag for the identity); the signature is
// valid iff x(A + B) mod n == r.  With lambda = N / D the chord or tangent slope, x3 = lambda^2 - xA - xB, so
// x3 == c (mod p)  <=>  N^2 == (c + xA + xB) D^2, tested for c = r and, when r + n < p, c = r + n.
template <class C>
__global__ void __launch_bounds__(256) verify_check_kernel(const u32* a_xy, const uint8_t* a_inf, const u32* b_xy, const uint8_t* b_inf,
                                                           const u32* sig, uint8_t* ok, size_t n) {
  using O = OrderOf<C>;
  using Fe = typename C::Fe;
  constexpr int L = C::NW;
  ECGPU_GRID_STRIDE(i, n) {
    if (!ok[i]) continue;
    u32 r[L], ord[L], p[L], r2[L];
    ecdsa::load_be<L>(r, sig + i * 2 * L);
    smont::order<O>(ord);
    C::modulus(p);
    const u32 cy = mp_add<L>(r2, r, ord);
    const bool second = (cy == 0) && !mp_geq<L>(r2, p);       // r + n is also a possible x coordinate
    const bool ai = a_inf[i] != 0, bi = b_inf[i] != 0;
    const u32* pa = a_xy + i * 2 * L;
    const u32* pb = b_xy + i * 2 * L;
    bool valid = false;
    if (ai && bi) {
      valid = false;
    } else if (ai || bi) {
      u32 x[L];
      ecdsa::load_be<L>(x, (ai ? pb : pa));
      valid = mp_eq<L>(x, r) || (second && mp_eq<L>(x, r2));
    } else {
      Fe xa, ya, xb, yb, N, D, t, lhs, s;
      C::fe_load(xa, pa); C::fe_load(ya, pa + L);
      C::fe_load(xb, pb); C::fe_load(yb, pb + L);
      bool same_x = true, same_y = true;
#pragma unroll
      for (int j = 0; j < L; j++) { same_x &= (pa[j] == pb[j]); same_y &= (pa[L + j] == pb[L + j]); }
      bool defined = true;
      if (same_x) {
        if (same_y) {                 // tangent: N = 3 x^2 + a, D = 2 y  (y != 0 on a curve of odd order)
          C::fe_sqr(t, xa);
          if (!C::A_IS_ZERO) { Fe one; C::fe_one(one); C::fe_sub(t, t, one); }
          C::fe_add(N, t, t); C::fe_add(N, N, t);
          C::fe_add(D, ya, ya);
        } else {
          defined = false;            // A = -B: the sum is the identity
        }
      } else {
        C::fe_sub(N, yb, ya);
        C::fe_sub(D, xb, xa);
      }
      if (defined) {
        C::fe_sqr(lhs, N);
        C::fe_sqr(D, D);
        C::fe_add(s, xa, xb);
        u32 be[L];
        Fe c, rhs, d;
        ecdsa::store_be<L>(be, r);
        C::fe_load(c, be);
        C::fe_add(c, c, s);
        C::fe_mul(rhs, c, D);
        C::fe_sub(d, lhs, rhs);
        valid = C::fe_is_zero(d);
        if (!valid && second) {
          ecdsa::store_be<L>(be, r2);
          C::fe_load(c, be);
          C::fe_add(c, c, s);
          C::fe_mul(rhs, c, D);
          C::fe_sub(d, lhs, rhs);
          valid = C::fe_is_zero(d);
        }
      }
    }
    ok[i] = valid ? 1 : 0;
  }
}

// sign, last stage.  R = k G arrives affine.  Writes r || s, the recovery id (y_is_odd | x_reduced << 1) and ok
// (0: d or k out of range, or r = 0, or s = 0 - the reference returns Err for those).
template <class C, int BATCH>
__global__ void __launch_bounds__(256) sign_finish_kernel(const u32* d, const u32* k, const u32* z, const u32* r_xy, const uint8_t* r_inf,
                                                          u32* sig, uint8_t* recid, uint8_t* ok, size_t n, unsigned flags) {
  using O = OrderOf<C>;
  constexpr int L = O::L;
  u32 w[BATCH][L];
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * BATCH) {
    int cnt = 0;
    u32 good = 0;
#pragma unroll 1
    for (int b = 0; b < BATCH; b++) {
      const size_t i = base + (size_t)b * T;
      if (i >= n) break;
      u32 kk[L], dd[L];
      load_be<L>(kk, k + i * L);
      load_be<L>(dd, d + i * L);
      const bool g = in_range<O>(kk) && in_range<O>(dd) && !r_inf[i];
      if (!g) { mp_zero<L>(kk); kk[0] = 1; }
      smont::to_mont<O>(w[b], kk);
      good |= (g ? 1u : 0u) << b;
      cnt = b + 1;
    }
    batch_invert<O, BATCH>(w, cnt);
#pragma unroll 1
    for (int b = 0; b < cnt; b++) {
      const size_t i = base + (size_t)b * T;
      u32 x[L], ord[L], t[L], e[L], dm[L], rd[L], s[L];
      load_be<L>(x, r_xy + i * 2 * L);
      smont::order<O>(ord);
      const u32 bw = mp_sub<L>(t, x, ord);
      const bool x_reduced = (bw == 0);                 // x >= n: r = x - n  (p < 2n on these curves)
      mp_select<L>(x, x_reduced, t, x);
      load_be<L>(e, z + i * L);
      smont::reduce_once<O>(e);
      load_be<L>(dm, d + i * L);
      smont::to_mont<O>(dm, dm);
      smont::mul<O>(rd, x, dm);                         // r d
      smont::add<O>(e, e, rd);
      smont::mul<O>(s, e, w[b]);                        // k^-1 (z + r d)
      bool g = ((good >> b) & 1u) && !mp_is_zero<L>(x) && !mp_is_zero<L>(s);
      u32 y_odd = bswap32(r_xy[i * 2 * L + 2 * L - 1]) & 1u;
      if ((flags & ECDSA_LOW_S) && is_high<O>(s)) {     // normalize_s and flip the recovery parity (k256 ecdsa.rs:190-194)
        mp_sub<L>(s, ord, s);
        y_odd ^= 1u;
      }
      if (!g) { mp_zero<L>(x); mp_zero<L>(s); }
      store_be<L>(sig + i * 2 * L, x);
      store_be<L>(sig + i * 2 * L + L, s);
      if (recid) recid[i] = g ? (uint8_t)(y_odd | (x_reduced ? 2u : 0u)) : 0;
      ok[i] = g ? 1 : 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Public-key recovery (VerifyingKey::recover_from_prehash of the external ecdsa crate, exercised by
// k256/src/ecdsa.rs:259-336): R = decompress(r [+ n], y_is_odd), Q = -(z r^-1) G + (s r^-1) R.
//   recover_prep -> fixed-base kernel (u1 G) -> variable-base kernel (u2 R) -> recover_finish (affine sum)
// ---------------------------------------------------------------------------------------------
template <class C, int BATCH>
__global__ void __launch_bounds__(256) recover_prep_kernel(const u32* z, const u32* sig, const uint8_t* recid, u32* r_xy, u32* u1, u32* u2,
                                                           uint8_t* ok, size_t n, unsigned flags) {
  using O = OrderOf<C>;
  using Fe = typename C::Fe;
  constexpr int L = O::L;
  u32 w[BATCH][L];
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * BATCH) {
    int cnt = 0;
    u32 good = 0;
#pragma unroll 1
    for (int b = 0; b < BATCH; b++) {
      const size_t i = base + (size_t)b * T;
      if (i >= n) break;
      u32 r[L], s[L], x[L], ord[L], pm[L];
      load_be<L>(r, sig + i * 2 * L);
      load_be<L>(s, sig + i * 2 * L + L);
      bool g = in_range<O>(r) && in_range<O>(s) && recid[i] < 4;
      if ((flags & ECDSA_LOW_S) && is_high<O>(s)) g = false;     // the final verify_prehash of the reference rejects it
      // x coordinate of R: r, or r + n when the recovery id says x was reduced; must stay below p
      smont::order<O>(ord);
      C::modulus(pm);
      mp_copy<L>(x, r);
      if (recid[i] & 2) {
        const u32 cy = mp_add<L>(x, r, ord);
        if (cy || mp_geq<L>(x, pm)) g = false;
      }
      u32 xb[L];
      store_be<L>(xb, x);
      Fe fx, rhs, y, ny;
      C::fe_load(fx, xb);
      C::curve_rhs(rhs, fx);
      const bool has = C::fe_sqrt(y, rhs);
      C::fe_neg(ny, y);
      const bool odd = C::fe_is_odd(y);
      C::fe_select(y, odd == ((recid[i] & 1) != 0), y, ny);
      g = g && has;
      if (!g) { C::fe_zero(fx); C::fe_zero(y); mp_zero<L>(r); r[0] = 1; }
      C::fe_store(r_xy + i * 2 * L, fx);
      C::fe_store(r_xy + i * 2 * L + L, y);
      smont::to_mont<O>(w[b], r);
      good |= (g ? 1u : 0u) << b;
      cnt = b + 1;
    }
    batch_invert<O, BATCH>(w, cnt);                  // r^-1, Montgomery form
#pragma unroll 1
    for (int b = 0; b < cnt; b++) {
      const size_t i = base + (size_t)b * T;
      u32 e[L], s[L], a[L], c[L], ord[L];
      load_be<L>(e, z + i * L);
      smont::reduce_once<O>(e);
      load_be<L>(s, sig + i * 2 * L + L);
      const bool g = (good >> b) & 1u;
      if (!g) { mp_zero<L>(e); mp_zero<L>(s); }
      smont::mul<O>(a, e, w[b]);                      // z r^-1
      smont::order<O>(ord);
      if (!mp_is_zero<L>(a)) mp_sub<L>(a, ord, a);    // u1 = -(z r^-1)
      smont::mul<O>(c, s, w[b]);                      // u2 = s r^-1
      store_be<L>(u1 + i * L, a);
      store_be<L>(u2 + i * L, c);
      ok[i] = g ? 1 : 0;
    }
  }
}

// out = A + B in affine coordinates, one field inversion per BATCH elements; ok[i] = 0 (and zeros) for the identity
template <class C, int BATCH>
__global__ void __launch_bounds__(256) recover_finish_kernel(const u32* a_xy, const uint8_t* a_inf, const u32* b_xy, const uint8_t* b_inf, u32* out_xy,
                                                             uint8_t* ok, size_t n) {
  using Fe = typename C::Fe;
  constexpr int L = C::NW;
  Fe pre[BATCH];
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * BATCH) {
    u32 kinds = 0;          // 0 = identity / rejected, 1 = one operand is the identity, 2 = chord or tangent
    int cnt = 0;
    Fe acc; C::fe_one(acc);
#pragma unroll 1
    for (int b = 0; b < BATCH; b++) {
      const size_t i = base + (size_t)b * T;
      if (i >= n) break;
      cnt = b + 1;
      pre[b] = acc;
      if (!ok[i]) continue;
      const bool ai = a_inf[i] != 0, bi = b_inf[i] != 0;
      if (ai && bi) continue;
      if (ai || bi) { kinds |= 1u << (2 * b); continue; }
      const u32* pa = a_xy + i * 2 * L;
      const u32* pb = b_xy + i * 2 * L;
      bool same_x = true, same_y = true;
#pragma unroll
      for (int j = 0; j < L; j++) { same_x &= (pa[j] == pb[j]); same_y &= (pa[L + j] == pb[L + j]); }
      if (same_x && !same_y) continue;
      Fe d, t;
      if (same_x) { C::fe_load(t, pa + L); C::fe_add(d, t, t); }
      else { Fe xa, xb; C::fe_load(xa, pa); C::fe_load(xb, pb); C::fe_sub(d, xb, xa); }
      kinds |= 2u << (2 * b);
      C::fe_mul(acc, acc, d);
    }
    Fe inv_all;
    C::fe_inv(inv_all, acc);
#pragma unroll 1
    for (int b = cnt - 1; b >= 0; b--) {
      const size_t i = base + (size_t)b * T;
      const u32 kind = (kinds >> (2 * b)) & 3u;
      const u32* pa = a_xy + i * 2 * L;
      const u32* pb = b_xy + i * 2 * L;
      Fe x3, y3;
      C::fe_zero(x3); C::fe_zero(y3);
      if (kind == 1) {
        const u32* src = a_inf[i] ? pb : pa;
        C::fe_load(x3, src);
        C::fe_load(y3, src + L);
      } else if (kind == 2) {
        Fe xa, ya, xb, yb, N, D, di, lam, t;
        C::fe_load(xa, pa); C::fe_load(ya, pa + L);
        C::fe_load(xb, pb); C::fe_load(yb, pb + L);
        bool same_x = true;
#pragma unroll
        for (int j = 0; j < L; j++) same_x &= (pa[j] == pb[j]);
        if (same_x) {
          C::fe_sqr(t, xa);
          if (!C::A_IS_ZERO) { Fe one; C::fe_one(one); C::fe_sub(t, t, one); }
          C::fe_add(N, t, t); C::fe_add(N, N, t);
          C::fe_add(D, ya, ya);
        } else {
          C::fe_sub(N, yb, ya);
          C::fe_sub(D, xb, xa);
        }
        C::fe_mul(di, inv_all, pre[b]);
        C::fe_mul(inv_all, inv_all, D);
        C::fe_mul(lam, N, di);
        C::fe_sqr(x3, lam);
        C::fe_sub(x3, x3, xa); C::fe_sub(x3, x3, xb);
        C::fe_sub(t, xa, x3);
        C::fe_mul(y3, lam, t);
        C::fe_sub(y3, y3, ya);
      }
      C::fe_store(out_xy + i * 2 * L, x3);
      C::fe_store(out_xy + i * 2 * L + L, y3);
      ok[i] = kind ? 1 : 0;
    }
  }
}

}  // namespace ecdsa
}  // namespace ecgpu
