// Constant-time variable-base scalar multiplication for SECRET scalars on secp256k1 (ECDH), per-lane body (host + device).
//
// The reference's `Mul` (k256/src/arithmetic/mul.rs:342-393) is GLV + complete formulas + masked table scans; what it pays
// per multiplication is two tables of eight PROJECTIVE points built by complete additions (14 x 12M), complete doublings
// (6M + 2S and a dozen field additions), projective table entries in the 66 additions of the window loop (12M each) and its
// own inversion for the affine result.  The group element k P is what is specified, so this schedule keeps the reference's
// discipline - nothing but data depends on k: no branch, no address - on the cheaper JACOBIAN formulas of the public-data
// kernel (mulfast_k256.hpp), which are EXCEPTION-FREE on the operands this loop meets (argument below):
//   * one table [P .. 8P] per unit, brought to a COMMON Z without any inversion ("effective affine": the loop runs on the
//     curve isomorphic by u = Z8, the result's Z is multiplied by Z8 at the end), kept in a lane-interleaved global workspace
//     of 512 bytes per lane: all lanes of a wave read the same entry number, so every load of a scan is one 1 KB row per wave;
//     lambda P's table is the same entries with x multiplied by beta when the lambda half reads them;
//   * per window four Jacobian doublings (a = 0: 3M + 4S, total on every input including Z = 0), ONE masked scan over ALL
//     eight entries that picks BOTH halves' digits (arithmetic AND / OR masks), masked negations for the digit and half
//     signs, and two mixed additions (8M + 3S) executed for EVERY digit; a zero digit or an empty accumulator is resolved by
//     masks afterwards (the accumulator's "empty" state is itself a mask that the digits update, not a test of Z);
//   * per-lane batched conversion of the results to affine (one inversion per pass).
// 128 x 7 + 66 x 11 + 33 + ~80 = ~1 735 field multiplications (and a third of the complete formulas' field additions)
// instead of the reference schedule's 1 984 + 272.  Digits by the reference's branch-free recoding (Radix16Decomposition).
//
// Why the Jacobian mixed addition never meets an exceptional case.  Let L = {(x, y) : x + y lambda = 0 mod n}, the lattice
// of the GLV split, basis v1 = (a1, b1), v2 = (a2, b2) of mul.rs:129-152.  Its shortest non-zero vector in the maximum norm is
// v1 with |b1| = 0xe4437ed6010e88286f547fa90abfe4c3 = 2^127.835 =: mu (enumeration over the reduced basis; checked in
// tests/test_oracle_golden.py::test_k256_glv_bounds).  decompose_scalar returns k = +-k1 +- k2 lambda with
// 0 <= k1 < (a1 + a2 + 1) / 2 = 2^127.346 and 0 <= k2 < (b2 - b1) / 2 + 1 = 2^127.113 (the bounds libsecp256k1 proves for
// this rounding; the same test checks them on the corners of the fundamental cell).  With P1 = +-P and P2 = +-lambda P (the
// halves' signs folded into the points) the accumulator is always x P1 + y P2 with integers 0 <= x <= k1 + 8, 0 <= y <= k2 + 8:
// after the four doublings of window j it is (16 s, 16 t) with s, t the digit prefixes of k1, k2 (s = sum_{i > j} d_i 16^(i-j-1)
// >= 0 as in varbase_ct.hpp), then (16 s + d, 16 t) after the first addition.  An addition acc + Q with Q = e P1 or e P2,
// 1 <= e <= 8, is exceptional iff acc = +-Q or acc = O, i.e. iff (x -+ e, y), (x, y -+ e) or (x, y) lies in L.  All these vectors
// have maximum norm below 2^127.346 + 16 < mu, so they lie in L only if they are ZERO:
//   acc = O                      iff x = y = 0: the accumulator is still empty (tracked as a mask; the result is then Q itself);
//   (16 s -+ e, 16 t) = 0        needs 16 | e: impossible for 1 <= e <= 8;
//   (16 s + d, 16 t -+ e) = 0    likewise.
// So once a non-zero digit has been added the accumulator is never O again and never +-Q; the only special operands are the
// empty accumulator and a zero digit, both masks.  The table build adds P to jP, j = 2 .. 7: exceptional only if (j -+ 1) P = O,
// impossible for a point of prime order n.  An identity INPUT is replaced by G under a mask and the result forced to the
// identity.  Inputs that are not on the curve give unspecified output (they violate the reference's type invariants;
// ecgpu_ecdh_batch rejects them before this kernel); the instruction stream does not depend on them.
// Round 3 first shipped this kernel on the complete formulas (the argument above was thought not to carry over from the fold
// of varbase_ct.hpp; it does, with the bounds made explicit): 8.8 x 10^7 /s then, 1.06 x 10^8 /s now.
// The kernel is instantiated in ops_k256_ct.hip, where the field additions' rare carry paths run unconditionally (fe_k256.hpp ECGPU_K256_RARE): no branch depends on data.
#pragma once
#include "varbase_ct.hpp"
#include "mulfast_k256.hpp"

namespace ecgpu {
namespace vbct {

// all ones iff the weakly reduced field element is zero (raw 0 or raw p), arithmetic
ECGPU_HD u32 k256_zero_mask(const FeK256& a) {
  u32 z = 0, o = 0xFFFFFFFFu;
#pragma unroll
  for (int w = 0; w < 8; w++) z |= a.v[w];
#pragma unroll
  for (int w = 2; w < 8; w++) o &= a.v[w];
  const u32 np = ~o | (a.v[1] ^ 0xFFFFFFFEu) | (a.v[0] ^ (0u - k256::C_LO));     // zero iff a is the raw modulus
  const u32 is0 = ((z | (0u - z)) >> 31) ^ 1u, isp = ((np | (0u - np)) >> 31) ^ 1u;
  return 0u - (is0 | isp);
}
ECGPU_HD void k256_mask_select(FeK256& r, u32 mk, const FeK256& a, const FeK256& b) {
#pragma unroll
  for (int w = 0; w < 8; w++) r.v[w] = (a.v[w] & mk) | (b.v[w] & ~mk);
}

// p += (x2, y2) by the plain Jacobian mixed addition (8M + 3S), NO exceptional-case handling: valid iff p is finite and
// p != +-(x2, y2); anything else gives garbage that the caller masks away.  `zr` (optional) receives Z3 / Z1 = H.
ECGPU_HD void k256_add_mixed_raw(JacK256& p, const FeK256& x2, const FeK256& y2, FeK256* zr) {
  using namespace k256;
  FeK256 h, r, t, u;
  sqr(t, p.z);
  mul(h, x2, t);
  mul(t, p.z, t); mul(r, t, y2);
  sub(h, h, p.x);
  sub(r, r, p.y);
  if (zr) *zr = h;
  mul(p.z, p.z, h);
  sqr(t, h);
  mul(h, t, h);
  mul(t, p.x, t);
  sqr(u, r);
  sub(u, u, h); sub(u, u, t); sub(p.x, u, t);
  sub(t, t, p.x);
  neg(u, p.y);
  mul_add2(p.y, r, t, u, h);                 // Y3 = R (V - X3) + (-Y1) HHH, one reduction (fe_k256.hpp)
}

// acc += (qx, qy) for every digit; `zd` is all ones iff the digit is zero (acc is kept), `empty` all ones while acc is still O
// (the result is then the entry itself).  Both are masks; the addition itself always executes.
ECGPU_HD void k256_ct_accumulate(JacK256& acc, u32& empty, const FeK256& qx, const FeK256& qy, u32 zd, const FeK256& one) {
  JacK256 t = acc;
  k256_add_mixed_raw(t, qx, qy, nullptr);                                            // garbage for an empty accumulator or a zero digit
  const u32 take_q = empty & ~zd;                                                    // first non-zero digit
  k256_mask_select(t.x, take_q, qx, t.x);
  k256_mask_select(t.y, take_q, qy, t.y);
  k256_mask_select(t.z, take_q, one, t.z);
  k256_mask_select(acc.x, zd, acc.x, t.x);
  k256_mask_select(acc.y, zd, acc.y, t.y);
  k256_mask_select(acc.z, zd, acc.z, t.z);
  empty &= zd;
}

// chunk layout of one lane: the unit's table (8 entries x (x, y)), the H ratios of its construction (zr[2 .. 7]), then per slot
// of the pass the parked result (x, y, z) and the prefix product of the output inversion
constexpr int K256_CT_TAB = 0, K256_CT_ZR = 32, K256_CT_RES = 48;
template <int BATCH> constexpr int k256_lane_chunks() { return K256_CT_RES + BATCH * 8; }
template <int BATCH> constexpr int k256_res_chunk(int b) { return K256_CT_RES + b * 6; }
template <int BATCH> constexpr int k256_pre_chunk(int b) { return K256_CT_RES + BATCH * 6 + b * 2; }

// One pass of one lane: units base, base + T, .., base + (BATCH - 1) T (those below n).
template <int BATCH>
ECGPU_HD void lane_pass_k256(const u32* scalars, const u32* points, int pt_fmt, u32* out, int out_fmt, uint8_t* out_inf, size_t n, size_t base,
                             size_t T, const LaneMem& ws, const DigitMem& dm) {
  using C = CurveK256;
  constexpr int NW = 8, CW = 2;
  static_assert(BATCH <= 32, "result slots per pass");
  const int pw = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * NW;
  int cnt = 0;
  u32 res_inf = 0;                             // identity results, one bit per slot
  FeK256 one, zero, beta_;
  k256::set_one(one); k256::set_zero(zero); k256::beta(beta_);
#pragma unroll 1
  for (int b = 0; b < BATCH; b++) {
    const size_t i = base + (size_t)b * T;
    if (i >= n) break;                         // public: the batch size
    cnt = b + 1;
    // ---- the point (public data; identity inputs are replaced by G under a mask and the result forced to the identity)
    const u32* src = points + i * pw;
    FeK256 px, py, pz;
    C::fe_load(px, src);
    C::fe_load(py, src + NW);
    u32 inf_mask;
    if (pt_fmt == FMT_PROJECTIVE) {            // public: the wire format.  Homogeneous (X : Y : Z) = Jacobian (XZ, YZ^2, Z): run on the curve isomorphic by u = Z
      C::fe_load(pz, src + 2 * NW);
      inf_mask = k256_zero_mask(pz);
      FeK256 zz;
      k256::mul(px, px, pz);
      k256::sqr(zz, pz);
      k256::mul(py, py, zz);
    } else {
      u32 z = 0;
#pragma unroll
      for (int w = 0; w < 2 * NW; w++) z |= src[w];
      inf_mask = 0u - (((z | (0u - z)) >> 31) ^ 1u);
      pz = one;
    }
    {
      PtK256 g;
      k256::generator(g);
      k256_mask_select(px, inf_mask, g.x, px);
      k256_mask_select(py, inf_mask, g.y, py);
      k256_mask_select(pz, inf_mask, one, pz);
    }
    // ---- the unit's table [P .. 8P] with the common denominator Z8 (mulfast_k256.hpp::table_build_globalz, branch-free and
    //      through the workspace): the co-Z chain 2P, 3P = 2P + P, .. rewrites P to the denominator of every new multiple, so
    //      each step is a co-Z addition (4M + 2S, no Z multiplied out; exception-free: (j -+ 1) P = O is impossible), the ratios
    //      zr_j = Z_j / Z_(j-1) are kept, entry j-1 <- (X_j s^2, Y_j s^3) with s = Z8 / Z_j, and the last rewrite of P is entry 0
    FeK256 zfix;
    {
      JacK256 m;
      FeK256 qx, qy;
      k256::coz_double_affine(m, qx, qy, px, py);               // 2P (P is finite and has no 2-torsion), P over Z = 2y
      fe_st<C>(ws, K256_CT_TAB + 4, m.x); fe_st<C>(ws, K256_CT_TAB + 4 + CW, m.y);
      const FeK256 z2 = m.z;
#pragma unroll 1
      for (int j = 2; j < 8; j++) {
        FeK256 h;
        k256::coz_add_update(m.x, m.y, qx, qy, h);               // (j + 1) P = jP + P
        fe_st<C>(ws, K256_CT_TAB + 4 * j, m.x); fe_st<C>(ws, K256_CT_TAB + 4 * j + CW, m.y);
        fe_st<C>(ws, K256_CT_ZR + CW * j, h);
      }
      fe_st<C>(ws, K256_CT_TAB, qx); fe_st<C>(ws, K256_CT_TAB + CW, qy);
      FeK256 s = one;
#pragma unroll 1
      for (int j = 6; j >= 1; j--) {
        FeK256 f;
        fe_ld<C>(f, ws, K256_CT_ZR + CW * (j + 1));
        k256::mul(s, s, f);
        FeK256 s2, s3, tx, ty;
        k256::sqr(s2, s);
        k256::mul(s3, s2, s);
        fe_ld<C>(tx, ws, K256_CT_TAB + 4 * j); fe_ld<C>(ty, ws, K256_CT_TAB + 4 * j + CW);
        k256::mul(tx, tx, s2);
        k256::mul(ty, ty, s3);
        fe_st<C>(ws, K256_CT_TAB + 4 * j, tx); fe_st<C>(ws, K256_CT_TAB + 4 * j + CW, ty);
      }
      k256::mul(s, s, z2);                                       // Z8
      k256::mul(zfix, s, pz);                                    // back from both isomorphisms at the end
    }
    // ---- the scalar: GLV split and digits (mul.rs:260-305), all branch-free
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
    // ---- the window loop (mul.rs:365-391)
    JacK256 acc;
    acc.x = zero; acc.y = zero; acc.z = zero;
    u32 empty = 0xFFFFFFFFu;                     // all ones while no non-zero digit has been added (the accumulator is O)
#pragma unroll 1
    for (int j = 32; j >= 0; j--) {              // digit 32 is the pair of carry digits (0 or 1)
      if (j != 32) {
#pragma unroll 1
        for (int d = 0; d < 4; d++) k256::jac_double(acc);
      }
      const u32 w1 = dm.ld(j == 32 ? 0 : (j >> 3)), w2 = dm.ld(4 + (j == 32 ? 0 : (j >> 3)));
      const int s1 = (j == 32) ? (int)d1.top : k256::radix16_digit(w1, j & 7);
      const int s2 = (j == 32) ? (int)d2.top : k256::radix16_digit(w2, j & 7);
      const u32 sg1 = (u32)(s1 >> 31), sg2 = (u32)(s2 >> 31);
      const u32 mag1 = ((u32)s1 ^ sg1) - sg1, mag2 = ((u32)s2 ^ sg2) - sg2;           // 0 .. 8
      FeK256 q1x = zero, q1y = zero, q2x = zero, q2y = zero;
#pragma unroll 2
      for (int e = 0; e < 8; e++) {
        ECGPU_TABLE_TOUCH(b * 8 + e);
        const u32 m1 = 0u - (((mag1 ^ (u32)(e + 1)) - 1u) >> 31);                    // all ones iff mag1 == e + 1
        const u32 m2 = 0u - (((mag2 ^ (u32)(e + 1)) - 1u) >> 31);
        FeK256 tx, ty;
        fe_ld<C>(tx, ws, K256_CT_TAB + 4 * e);
        fe_ld<C>(ty, ws, K256_CT_TAB + 4 * e + CW);
#pragma unroll
        for (int w = 0; w < NW; w++) {
          q1x.v[w] |= tx.v[w] & m1; q1y.v[w] |= ty.v[w] & m1;
          q2x.v[w] |= tx.v[w] & m2; q2y.v[w] |= ty.v[w] & m2;
        }
      }
      k256::mul(q2x, q2x, beta_);                                                    // lambda (x, y) = (beta x, y)
      FeK256 ny;
      k256::neg(ny, q1y);
      k256_mask_select(q1y, sg1 ^ neg1, ny, q1y);                                    // digit < 0 xor half negative
      k256::neg(ny, q2y);
      k256_mask_select(q2y, sg2 ^ neg2, ny, q2y);
      k256_ct_accumulate(acc, empty, q1x, q1y, 0u - ((mag1 - 1u) >> 31), one);
      k256_ct_accumulate(acc, empty, q2x, q2y, 0u - ((mag2 - 1u) >> 31), one);
    }
    k256::mul(acc.z, acc.z, zfix);
    const u32 inf = inf_mask | empty;                     // an identity input or k = 0 mod n: the identity
    res_inf |= (inf & 1u) << b;
    fe_st<C>(ws, k256_res_chunk<BATCH>(b), acc.x); fe_st<C>(ws, k256_res_chunk<BATCH>(b) + CW, acc.y); fe_st<C>(ws, k256_res_chunk<BATCH>(b) + 2 * CW, acc.z);
  }
#pragma unroll
  for (int w = 0; w < 8; w++) dm.st(w, 0u);             // the last unit's recoded halves do not stay in LDS
  // ---- x = X / Z^2, y = Y / Z^3 with one inversion for the cnt results of this lane
  {
    FeK256 run = one;
#pragma unroll 1
    for (int b = 0; b < cnt; b++) {
      FeK256 z;
      fe_st<C>(ws, k256_pre_chunk<BATCH>(b), run);
      fe_ld<C>(z, ws, k256_res_chunk<BATCH>(b) + 2 * CW);
      const u32 zm = k256_zero_mask(z) | (0u - ((res_inf >> b) & 1u));     // Z = 0 besides the identity: only for input that is not on the curve
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
      FeK256 z, pre, zi, zi2, zi3, x, yv;
      fe_ld<C>(z, ws, k256_res_chunk<BATCH>(b) + 2 * CW);
      k256_mask_select(z, inf, one, z);
      fe_ld<C>(pre, ws, k256_pre_chunk<BATCH>(b));
      k256::mul(zi, inv, pre);
      k256::mul(inv, inv, z);
      k256::sqr(zi2, zi);
      k256::mul(zi3, zi2, zi);
      fe_ld<C>(x, ws, k256_res_chunk<BATCH>(b));
      k256::mul(x, x, zi2);
      fe_ld<C>(yv, ws, k256_res_chunk<BATCH>(b) + CW);
      k256::mul(yv, yv, zi3);
      // the parked result and its prefix product are secrets of the same rank as the output: they do not stay in the workspace
      fe_st<C>(ws, k256_res_chunk<BATCH>(b), zero); fe_st<C>(ws, k256_res_chunk<BATCH>(b) + CW, zero); fe_st<C>(ws, k256_res_chunk<BATCH>(b) + 2 * CW, zero);
      fe_st<C>(ws, k256_pre_chunk<BATCH>(b), zero);
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
