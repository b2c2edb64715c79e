// Constant-time variable-base scalar multiplication for SECRET scalars on secp256k1 (ECDH), per-lane body (host + device).
//
// The reference's `Mul` (k256/src/arithmetic/mul.rs:342-393) is already GLV + complete formulas + masked table scans; what it
// pays per multiplication is two tables of eight PROJECTIVE points built by complete additions (14 x 12M), projective table
// entries in the 66 additions of the window loop (12M each) and its own inversion for the affine result.  The fold that makes
// Jacobian formulas exception-free on P-256 / P-384 (varbase_ct.hpp) does not carry over to the endomorphism split - with
// k = k1 + k2 lambda the accumulator is 16 (s1 + lambda s2) P and the GLV lattice has vectors short enough for
// 16 s1 -+ d + 16 s2 lambda = 0 (mod n) - so this schedule keeps the reference's COMPLETE formulas (total on every input, no
// argument needed) and takes its savings elsewhere:
//   * one table [P .. 8P] per unit, built by complete additions jP + P (7 x 12M) and brought to AFFINE form with one
//     inversion per pass (x = X / Z, y = Y / Z: 5M per entry); lambda P's table is the same entries with x multiplied by beta
//     when the lambda half reads them (one multiplication per window instead of a second table);
//   * per window four complete doublings (6M + 2S), ONE masked scan over the eight entries that picks BOTH halves' digits
//     (two AND / OR accumulators per entry read; lane-interleaved workspace: a wave reads each entry as one 1 KB row), masked
//     negations for the digit and half signs, and two complete MIXED additions (11M; a zero digit is the addend's infinity
//     flag, which the formula resolves by a select);
//   * per-lane batched conversion of the results to affine (one inversion per pass).
// 128 x 8 + 66 x 11 + 33 + ~150 = ~1 930 field multiplications instead of the reference schedule's 1 984 + 272.  Nothing but
// data depends on the scalar: digits by the reference's branch-free recoding (Radix16Decomposition), no digit-indexed address,
// no digit-dependent branch; the caveat of every schedule here applies (the field additions' rare carry path is a branch).
#pragma once
#include "varbase_ct.hpp"

namespace ecgpu {
namespace vbct {

ECGPU_HD u32 k256_zero_mask(const FeK256& a) { return 0u - (u32)k256::is_zero(a); }
ECGPU_HD void k256_mask_select(FeK256& r, u32 mk, const FeK256& a, const FeK256& b) {
#pragma unroll
  for (int w = 0; w < 8; w++) r.v[w] = (a.v[w] & mk) | (b.v[w] & ~mk);
}

// One pass of one lane: units base, base + T, .., base + (BATCH - 1) T (those below n).
template <int BATCH>
ECGPU_HD void lane_pass_k256(const u32* scalars, const u32* points, int pt_fmt, u32* out, int out_fmt, uint8_t* out_inf, size_t n, size_t base,
                             size_t T, const LaneMem& ws, const DigitMem& dm) {
  using C = CurveK256;
  constexpr int NW = 8, CW = 2;
  static_assert(BATCH <= 32, "table slots per pass");
  const int pw = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * NW;
  int cnt = 0;
  u32 infs = 0;                                // identity inputs, one bit per slot
  FeK256 one, zero;
  k256::set_one(one); k256::set_zero(zero);
  // ---- phase A: [P .. 8P] in homogeneous projective coordinates by complete mixed additions
#pragma unroll 1
  for (int s = 0; s < BATCH; s++) {
    const size_t i = base + (size_t)s * T;
    if (i >= n) break;                         // public: the batch size
    cnt = s + 1;
    const u32* src = points + i * pw;
    PtK256 p;
    C::fe_load(p.x, src);
    C::fe_load(p.y, src + NW);
    u32 inf_mask;
    if (pt_fmt == FMT_PROJECTIVE) {            // public: the wire format
      C::fe_load(p.z, src + 2 * NW);
      inf_mask = k256_zero_mask(p.z);
    } else {
      u32 z = 0;
#pragma unroll
      for (int w = 0; w < 2 * NW; w++) z |= src[w];
      inf_mask = 0u - (((z | (0u - z)) >> 31) ^ 1u);
      p.z = one;
    }
    {                                          // an identity input: the table is built for G, the result forced to the identity
      PtK256 g;
      k256::generator(g);
      k256_mask_select(p.x, inf_mask, g.x, p.x);
      k256_mask_select(p.y, inf_mask, g.y, p.y);
      k256_mask_select(p.z, inf_mask, one, p.z);
    }
    infs |= (inf_mask & 1u) << s;
    // j P = (j - 1) P + P with the complete addition (projective.rs:96-161: total, so no case analysis for a projective input)
    PtK256 t = p;
    fe_st<C>(ws, entry_chunk<C>(s, 0), t.x); fe_st<C>(ws, entry_chunk<C>(s, 0) + CW, t.y); fe_st<C>(ws, entry_chunk<C>(s, 0) + 2 * CW, t.z);
#pragma unroll 1
    for (int j = 1; j < 8; j++) {
      PtK256 u;
      k256::pt_add(u, t, p);
      t = u;
      fe_st<C>(ws, entry_chunk<C>(s, j), t.x); fe_st<C>(ws, entry_chunk<C>(s, j) + CW, t.y); fe_st<C>(ws, entry_chunk<C>(s, j) + 2 * CW, t.z);
    }
  }
  // ---- phase B: all cnt * 8 entries to affine (x = X / Z, y = Y / Z) with one inversion
  {
    FeK256 acc = one;
#pragma unroll 1
    for (int e = 0; e < cnt * 8; e++) {
      fe_st<C>(ws, pre_chunk<C, BATCH>(e), acc);
      FeK256 z;
      fe_ld<C>(z, ws, e * 3 * CW + 2 * CW);
      k256_mask_select(z, k256_zero_mask(z), one, z);      // only for input that is not on the curve: keep the batch clean
      k256::mul(acc, acc, z);
    }
    FeK256 ai;
    k256::inv(ai, acc);
#pragma unroll 1
    for (int e = cnt * 8 - 1; e >= 0; e--) {
      FeK256 z, zi, pre, x, y;
      fe_ld<C>(z, ws, e * 3 * CW + 2 * CW);
      k256_mask_select(z, k256_zero_mask(z), one, z);
      fe_ld<C>(pre, ws, pre_chunk<C, BATCH>(e));
      k256::mul(zi, ai, pre);
      k256::mul(ai, ai, z);
      fe_ld<C>(x, ws, e * 3 * CW);
      k256::mul(x, x, zi);
      fe_st<C>(ws, e * 3 * CW, x);
      fe_ld<C>(y, ws, e * 3 * CW + CW);
      k256::mul(y, y, zi);
      fe_st<C>(ws, e * 3 * CW + CW, y);
    }
  }
  // ---- phase C: the window loop (mul.rs:365-391), one unit after the other; the result is parked in the unit's entry 0
  FeK256 beta_;
  k256::beta(beta_);
#pragma unroll 1
  for (int b = 0; b < cnt; b++) {
    const size_t i = base + (size_t)b * T;
    u32 k[NW];
    words_load_be<NW>(k, scalars + i * NW);
    k256::scalar_reduce_once(k);
    k256::GlvSplit sp;
    k256::glv_split(sp, k);
    k256::Radix16<4> d1, d2;
    k256::radix16_recode<4>(d1, sp.k1);
    k256::radix16_recode<4>(d2, sp.k2);
#pragma unroll
    for (int w = 0; w < 4; w++) { dm.st(w, d1.y[w]); dm.st(4 + w, d2.y[w]); }
    const u32 neg1 = 0u - (u32)sp.neg1, neg2 = 0u - (u32)sp.neg2;
    PtK256 acc;
    k256::pt_identity(acc);
#pragma unroll 1
    for (int j = 32; j >= 0; j--) {              // digit 32 is the pair of carry digits (0 or 1)
      if (j != 32) {
#pragma unroll 1
        for (int d = 0; d < 4; d++) { PtK256 u; k256::pt_double(u, acc); acc = u; }
      }
      const u32 w1 = dm.ld(j == 32 ? 0 : (j >> 3)), w2 = dm.ld(4 + (j == 32 ? 0 : (j >> 3)));
      const int s1 = (j == 32) ? (int)d1.top : k256::radix16_digit(w1, j & 7);
      const int s2 = (j == 32) ? (int)d2.top : k256::radix16_digit(w2, j & 7);
      const u32 sg1 = (u32)(s1 >> 31), sg2 = (u32)(s2 >> 31);
      const u32 mag1 = ((u32)s1 ^ sg1) - sg1, mag2 = ((u32)s2 ^ sg2) - sg2;           // 0 .. 8
      AfK256 q1, q2;
      q1.x = zero; q1.y = zero; q2.x = zero; q2.y = zero;
#pragma unroll 2
      for (int e = 0; e < 8; e++) {
        ECGPU_TABLE_TOUCH(b * 8 + e);
        const u32 m1 = 0u - (((mag1 ^ (u32)(e + 1)) - 1u) >> 31);                    // all ones iff mag1 == e + 1
        const u32 m2 = 0u - (((mag2 ^ (u32)(e + 1)) - 1u) >> 31);
        FeK256 tx, ty;
        fe_ld<C>(tx, ws, (b * 8 + e) * 3 * CW);
        fe_ld<C>(ty, ws, (b * 8 + e) * 3 * CW + CW);
#pragma unroll
        for (int w = 0; w < NW; w++) {
          q1.x.v[w] |= tx.v[w] & m1; q1.y.v[w] |= ty.v[w] & m1;
          q2.x.v[w] |= tx.v[w] & m2; q2.y.v[w] |= ty.v[w] & m2;
        }
      }
      k256::mul(q2.x, q2.x, beta_);                                                  // lambda (x, y) = (beta x, y)
      FeK256 ny;
      k256::neg(ny, q1.y);
      k256_mask_select(q1.y, sg1 ^ neg1, ny, q1.y);                                  // digit < 0 xor half negative
      k256::neg(ny, q2.y);
      k256_mask_select(q2.y, sg2 ^ neg2, ny, q2.y);
      q1.inf = (0u - ((mag1 - 1u) >> 31)) & 1u;                                      // zero digit: AffinePoint::IDENTITY
      q2.inf = (0u - ((mag2 - 1u) >> 31)) & 1u;
      PtK256 u;
      k256::pt_add_mixed(u, acc, q1);
      k256::pt_add_mixed(acc, u, q2);
    }
    const u32 inf = 0u - ((infs >> b) & 1u);
    k256_mask_select(acc.z, inf, zero, acc.z);            // an identity input: the result is the identity (Z = 0)
    fe_st<C>(ws, entry_chunk<C>(b, 0), acc.x); fe_st<C>(ws, entry_chunk<C>(b, 0) + CW, acc.y); fe_st<C>(ws, entry_chunk<C>(b, 0) + 2 * CW, acc.z);
  }
  // ---- phase D: x = X / Z, y = Y / Z with one inversion for the cnt results of this lane (Z = 0: the identity)
  {
    FeK256 run = one;
    u32 res_inf = 0;
#pragma unroll 1
    for (int b = 0; b < cnt; b++) {
      FeK256 z;
      fe_st<C>(ws, pre_chunk<C, BATCH>(b), run);
      fe_ld<C>(z, ws, entry_chunk<C>(b, 0) + 2 * CW);
      const u32 zm = k256_zero_mask(z);
      res_inf |= (zm & 1u) << b;
      k256_mask_select(z, zm, one, z);
      k256::mul(run, run, z);
    }
    FeK256 inv;
    k256::inv(inv, run);
#pragma unroll 1
    for (int b = cnt - 1; b >= 0; b--) {
      const size_t i = base + (size_t)b * T;
      const u32 inf = 0u - ((res_inf >> b) & 1u);
      FeK256 z, pre, zi, x, yv;
      fe_ld<C>(z, ws, entry_chunk<C>(b, 0) + 2 * CW);
      k256_mask_select(z, inf, one, z);
      fe_ld<C>(pre, ws, pre_chunk<C, BATCH>(b));
      k256::mul(zi, inv, pre);
      k256::mul(inv, inv, z);
      fe_ld<C>(x, ws, entry_chunk<C>(b, 0));
      k256::mul(x, x, zi);
      fe_ld<C>(yv, ws, entry_chunk<C>(b, 0) + CW);
      k256::mul(yv, yv, zi);
      k256_mask_select(x, inf, zero, x);
      if (out_fmt == FMT_PROJECTIVE) {          // public: the wire format.  (x : y : 1), identity (0 : 1 : 0)
        FeK256 zo;
        k256_mask_select(yv, inf, one, yv);
        k256_mask_select(zo, inf, zero, one);
        u32* o = out + i * 3 * NW;
        C::fe_store(o, x); C::fe_store(o + NW, yv); C::fe_store(o + 2 * NW, zo);
      } else {
        k256_mask_select(yv, inf, zero, yv);
        u32* o = out + i * 2 * NW;
        C::fe_store(o, x); C::fe_store(o + NW, yv);
        if (out_inf) out_inf[i] = (uint8_t)(inf & 1u);
      }
    }
  }
}

}  // namespace vbct
}  // namespace ecgpu
