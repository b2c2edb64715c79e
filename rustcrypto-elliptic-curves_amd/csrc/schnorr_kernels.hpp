// BIP340 Schnorr verification for batches: the elliptic-curve part of
// `VerifyingKey::verify_prehash` (k256/src/schnorr/verifying.rs:62-93).
//   e = tagged_hash("BIP0340/challenge", r || P.x || m) mod n is computed by the caller (SHA-256 is host-side glue);
//   R = s G + (-e) P must be finite, have even y and x(R) = r.
// Pipeline: verify_prep (decode r, s, lift_x of the key, -e) -> fixed-base kernel (s G) -> variable-base kernel
// ((-e) P) -> verify_check (affine sum with one inversion per BATCH signatures, parity and x tests).
// secp256k1 only.  Device code only.
#pragma once
#include "kernels.hpp"

namespace ecgpu {
namespace schnorr {

// decode: r in [1, p) (Signature::try_from, k256/src/schnorr.rs:142-160), s in [1, n), key x < p with a square root
// (VerifyingKey::from_bytes -> decompact, verifying.rs:39-45).  Writes P = (x, even y), u1 = s, u2 = -e mod n.
template <int UNUSED>      // a template only so that the header can be included by every curve's translation unit
__global__ void __launch_bounds__(256) verify_prep_kernel(const u32* px, const u32* sig, const u32* e, u32* p_xy, u32* u1, u32* u2, uint8_t* ok,
                                                          size_t n) {
  ECGPU_GRID_STRIDE(i, n) {
    u32 r[8], s[8], x[8], ee[8], ord[8], pm[8];
    words_load_be<8>(r, sig + i * 16);
    words_load_be<8>(s, sig + i * 16 + 8);
    words_load_be<8>(x, px + i * 8);
    words_load_be<8>(ee, e + i * 8);
    k256::order(ord);
    CurveK256::modulus(pm);
    bool g = !mp_is_zero<8>(r) && !mp_geq<8>(r, pm) && !mp_is_zero<8>(s) && !mp_geq<8>(s, ord) && !mp_geq<8>(x, pm);
    FeK256 fx, rhs, y, ny;
    CurveK256::fe_load(fx, px + i * 8);
    CurveK256::curve_rhs(rhs, fx);
    const bool has = k256::sqrt(y, rhs);
    k256::neg(ny, y);
    if (k256::is_odd(y)) y = ny;
    g = g && has;
    k256::scalar_reduce_once(ee);            // Reduce::reduce_bytes of the challenge hash
    u32 me[8];
    mp_sub<8>(me, ord, ee);                  // -e  (0 stays 0)
    if (mp_is_zero<8>(ee)) mp_zero<8>(me);
    if (!g) { mp_zero<8>(s); mp_zero<8>(me); k256::set_zero(fx); k256::set_zero(y); }
    CurveK256::fe_store(p_xy + i * 16, fx);
    CurveK256::fe_store(p_xy + i * 16 + 8, y);
    words_store_be<8>(u1 + i * 8, s);
    words_store_be<8>(u2 + i * 8, me);
    ok[i] = g ? 1 : 0;
  }
}

// R = A + B in affine coordinates; valid iff R is finite, y(R) is even and x(R) = r.
template <int BATCH>
__global__ void __launch_bounds__(256) verify_check_kernel(const u32* a_xy, const uint8_t* a_inf, const u32* b_xy, const uint8_t* b_inf,
                                                           const u32* sig, uint8_t* ok, size_t n) {
  FeK256 pre[BATCH];
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * BATCH) {
    // kind: 0 = rejected / identity, 1 = R is one of the operands (other is the identity), 2 = chord or tangent
    u32 kinds = 0;
    int cnt = 0;
    FeK256 acc; k256::set_one(acc);
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
      const u32* pa = a_xy + i * 16;
      const u32* pb = b_xy + i * 16;
      bool same_x = true, same_y = true;
#pragma unroll
      for (int j = 0; j < 8; j++) { same_x &= (pa[j] == pb[j]); same_y &= (pa[8 + j] == pb[8 + j]); }
      if (same_x && !same_y) continue;                   // A = -B
      FeK256 d, t;
      if (same_x) { CurveK256::fe_load(t, pa + 8); k256::add(d, t, t); }                     // 2 y
      else { FeK256 xa, xb; CurveK256::fe_load(xa, pa); CurveK256::fe_load(xb, pb); k256::sub(d, xb, xa); }
      kinds |= 2u << (2 * b);
      k256::mul(acc, acc, d);
    }
    FeK256 inv_all;
    k256::inv(inv_all, acc);
#pragma unroll 1
    for (int b = cnt - 1; b >= 0; b--) {
      const size_t i = base + (size_t)b * T;
      const u32 kind = (kinds >> (2 * b)) & 3u;
      bool valid = false;
      const u32* pa = a_xy + i * 16;
      const u32* pb = b_xy + i * 16;
      FeK256 x3, y3;
      if (kind == 1) {
        const u32* src = a_inf[i] ? pb : pa;
        CurveK256::fe_load(x3, src);
        CurveK256::fe_load(y3, src + 8);
        valid = true;
      } else if (kind == 2) {
        FeK256 xa, ya, xb, yb, N, D, di, lam, t;
        CurveK256::fe_load(xa, pa); CurveK256::fe_load(ya, pa + 8);
        CurveK256::fe_load(xb, pb); CurveK256::fe_load(yb, pb + 8);
        bool same_x = true;
#pragma unroll
        for (int j = 0; j < 8; j++) same_x &= (pa[j] == pb[j]);
        if (same_x) { k256::sqr(t, xa); k256::add(N, t, t); k256::add(N, N, t); k256::add(D, ya, ya); }
        else { k256::sub(N, yb, ya); k256::sub(D, xb, xa); }
        k256::mul(di, inv_all, pre[b]);                   // 1 / D
        k256::mul(inv_all, inv_all, D);
        k256::mul(lam, N, di);
        k256::sqr(x3, lam);
        k256::sub(x3, x3, xa); k256::sub(x3, x3, xb);
        k256::sub(t, xa, x3);
        k256::mul(y3, lam, t);
        k256::sub(y3, y3, ya);
        valid = true;
      }
      if (valid) {
        FeK256 r, d;
        CurveK256::fe_load(r, sig + i * 16);
        k256::sub(d, x3, r);
        valid = k256::is_zero(d) && !k256::is_odd(y3);
      }
      ok[i] = valid ? 1 : 0;
    }
  }
}

}  // namespace schnorr
}  // namespace ecgpu
