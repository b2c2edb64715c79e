// Constant-time variable-base scalar multiplication k*P for SECRET scalars (ECDH: elliptic_curve::ecdh::diffie_hellman,
// `&P * &k` with a secret k) on the prime-order NIST curves, per-lane body (host + device: tests/hosttwin walks it on the
// CPU and records the table entries it reads).
//
// The reference is constant-time for every `Mul` (primeorder/src/projective.rs:106-150: complete formulas, a 15-way
// masked scan per window, 256 / 384 complete doublings of 8M + 3S + 2 multiplications by b).  Its doublings are what it
// pays for completeness, and the group element k*P is what is specified, so this schedule keeps the reference's
// discipline - nothing but data depends on k: no branch, no address - on cheaper formulas that are EXCEPTION-FREE on
// the inputs it meets (argument below), not complete on all of E x E:
//   * k' = min(k, n - k) and +-P by masks, signed 4-bit digits d_j in [-8, 7] by the branch-free recoding
//     nibble_j(k' + 0x88..8) - 8 plus the carry digit d_top in {0, 1};
//   * per-lane table [P .. 8P] as a co-Z chain (one doubling with update, six co-Z additions of 4M + 2S: jacobian.hpp;
//     4 doublings and 3 general additions until late in round 3), all tables of a pass brought to affine form with ONE
//     inversion over one denominator per table (the chain's ratios give the other seven), kept in a lane-interleaved
//     global workspace: all lanes of a wave read the same entry number, so every load of the scan is one contiguous 1 KB
//     row per wave;
//   * per window 4 Jacobian doublings (a = -3: 4M + 4S, total on every input including Z = 0) and ONE mixed addition
//     (8M + 3S) of the entry that a masked scan over ALL eight entries picked (arithmetic AND / OR masks, the sign by a
//     masked negation), executed for every digit; a zero digit or an empty accumulator is resolved by masks afterwards;
//   * per-lane batched conversion to affine, identity results by masks.
// 256 x 8 + 65 x 11 + ~150 field multiplications per P-256 result instead of the reference schedule's 4 361 + 269.
//
// Why the Jacobian addition never meets an exceptional case (P of prime order n, the only kind these curves have
// besides the identity; 0 <= k' <= (n - 1) / 2):  let s_j = sum_{i >= j} d_i 16^(i - j), so the accumulator is
// 16 s_(j+1) P when digit d_j is added.  From k' = s_j 16^j + sum_{i < j} d_i 16^i and -8 <= d_i <= 7 follows
// -7/15 < s_j <= k' / 16^j + 8/15, so s_j >= 0, and 16 s_(j+1) +- d_j lies in [-8, (n - 1) / 2 + 17), strictly inside
// (-n, n).  The addition is exceptional iff 16 s_(j+1) = 0 or = +-d_j modulo n, hence as integers; 16 | d_j forces
// d_j = 0.  So the only special operands are an empty accumulator (s_(j+1) = 0: tracked in a mask, the result is then
// the table entry itself) and a zero digit (the accumulator is kept); once s_j > 0 it stays >= 8.  The table build adds
// P to jP, j = 2 .. 7: exceptional only if (j -+ 1) P = O, impossible for prime n > 8.  An identity INPUT is
// replaced by G under a mask and the result forced to the identity.  Inputs that are not on the curve give unspecified
// output, as they would violate the reference's type invariants; the instruction stream does not depend on them either.
// (The Montgomery field of these curves, fe_mont.hpp, has no carry branch: additions and subtractions end in a masked correction.)
#pragma once
#include "jacobian.hpp"

namespace ecgpu {
namespace vbct {

struct alignas(16) Chunk { u32 w[4]; };

// Lane-interleaved private memory: chunk i of this lane is base[i * stride] (base already points at the lane's column;
// stride = lanes per group, 256 on the device).  A wave reading chunk i reads 64 consecutive 16-byte words.
struct LaneMem {
  Chunk* base;
  size_t stride;
  ECGPU_HD Chunk ld(int i) const { return base[(size_t)i * stride]; }
  ECGPU_HD void st(int i, const Chunk& c) const { base[(size_t)i * stride] = c; }
};

template <class C> constexpr int cw() { return C::NW / 4; }                      // chunks per field element
// chunk layout of one lane: BATCH * 8 entries (x, y and a z slot for the chain's ratios / a parked result's Z), then the BATCH
// prefix products of the shared inversions
template <class C, int BATCH> constexpr int lane_chunks() { return BATCH * (8 * 3 + 1) * cw<C>(); }
template <class C> constexpr int entry_chunk(int slot, int e) { return (slot * 8 + e) * 3 * cw<C>(); }
template <class C, int BATCH> constexpr int pre_chunk(int i) { return BATCH * 8 * 3 * cw<C>() + i * cw<C>(); }

template <class C>
ECGPU_HD void fe_ld(typename C::Fe& r, const LaneMem& m, int chunk) {
#pragma unroll
  for (int c = 0; c < cw<C>(); c++) {
    const Chunk k = m.ld(chunk + c);
#pragma unroll
    for (int w = 0; w < 4; w++) r.v[4 * c + w] = k.w[w];
  }
}
template <class C>
ECGPU_HD void fe_st(const LaneMem& m, int chunk, const typename C::Fe& a) {
#pragma unroll
  for (int c = 0; c < cw<C>(); c++) {
    Chunk k;
#pragma unroll
    for (int w = 0; w < 4; w++) k.w[w] = a.v[4 * c + w];
    m.st(chunk + c, k);
  }
}
template <class C>
ECGPU_HD void jac_st(const LaneMem& m, int chunk, const Jac<C>& p) {
  fe_st<C>(m, chunk, p.x); fe_st<C>(m, chunk + cw<C>(), p.y); fe_st<C>(m, chunk + 2 * cw<C>(), p.z);
}

// r = mk ? a : b word by word, mk all ones or zero.  Arithmetic masks on purpose (see fixedbase_ct.hpp).
template <class C>
ECGPU_HD void fe_mask_select(typename C::Fe& r, u32 mk, const typename C::Fe& a, const typename C::Fe& b) {
#pragma unroll
  for (int w = 0; w < C::NW; w++) r.v[w] = (a.v[w] & mk) | (b.v[w] & ~mk);
}
// all ones iff the field element is zero (fully reduced representations: the NIST fields)
template <class C>
ECGPU_HD u32 fe_zero_mask(const typename C::Fe& a) {
  u32 t = 0;
#pragma unroll
  for (int w = 0; w < C::NW; w++) t |= a.v[w];
  return 0u - (((t | (0u - t)) >> 31) ^ 1u);
}

// p += (x2, y2) by the plain Jacobian mixed addition, 8M + 3S, NO exceptional-case handling: valid iff p is finite and
// p != +-(x2, y2); anything else gives garbage that the caller masks away.
template <class C>
ECGPU_HD void add_mixed_raw(Jac<C>& p, const typename C::Fe& x2, const typename C::Fe& y2) {
  using Fe = typename C::Fe;
  Fe h, r, t, u;
  C::fe_sqr(t, p.z);
  C::fe_mul(h, x2, t);
  C::fe_mul(t, p.z, t); C::fe_mul(r, t, y2);
  C::fe_sub(h, h, p.x);
  C::fe_sub(r, r, p.y);
  C::fe_mul(p.z, p.z, h);
  C::fe_sqr(t, h);
  C::fe_mul(h, t, h);
  C::fe_mul(t, p.x, t);
  C::fe_sqr(u, r);
  C::fe_sub(u, u, h); C::fe_sub(u, u, t); C::fe_sub(p.x, u, t);
  C::fe_sub(t, t, p.x); C::fe_mul(t, r, t);
  C::fe_mul(h, p.y, h);
  C::fe_sub(p.y, t, h);
}
// k' = min(k mod n, n - k mod n) and the mask of the flip, branch-free
template <class C>
ECGPU_HD u32 scalar_fold(u32* k, const u32* be) {
  constexpr int NW = C::NW;
  u32 ord[NW], t[NW], bw = 0;
  C::scalar_load(k, be);
  C::order(ord);
#pragma unroll
  for (int w = 0; w < NW; w++) t[w] = subb(k[w], ord[w], bw);
  const u32 keep = 0u - bw;                                   // borrow: k < n already
#pragma unroll
  for (int w = 0; w < NW; w++) k[w] = (k[w] & keep) | (t[w] & ~keep);
  bw = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) t[w] = subb(ord[w], k[w], bw);       // n - k
  u32 b2 = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) (void)subb(t[w], k[w], b2);          // borrow iff n - k < k
  const u32 flip = 0u - b2;
#pragma unroll
  for (int w = 0; w < NW; w++) k[w] = (t[w] & flip) | (k[w] & ~flip);
  return flip;
}

// One pass of one lane: units base, base + T, .., base + (BATCH - 1) T (those below n).
template <class C, int BATCH>
ECGPU_HD void lane_pass(const u32* scalars, const u32* points, int pt_fmt, u32* out, int out_fmt, uint8_t* out_inf, size_t n, size_t base,
                        size_t T, const LaneMem& ws, const DigitMem& dm) {
  static_assert(BATCH <= 32 && C::NW % 4 == 0, "table slots per pass; 16-byte chunks");
  constexpr int NW = C::NW, CW = cw<C>();
  using Fe = typename C::Fe;
  const int pw = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * NW;
  int cnt = 0;
  u32 infs = 0;                                // identity inputs, one bit per slot (public: a property of the point)
  Fe one, zero;
  C::fe_one(one); C::fe_zero(zero);
  // ---- phase A: Jacobian tables [P .. 8P] (the sign of the scalar's fold is applied to the scanned entry later)
#pragma unroll 1
  for (int s = 0; s < BATCH; s++) {
    const size_t i = base + (size_t)s * T;
    if (i >= n) break;                         // public: the batch size
    cnt = s + 1;
    const u32* src = points + i * pw;
    Jac<C> p;
    C::fe_load(p.x, src);
    C::fe_load(p.y, src + NW);
    u32 inf_mask;
    if (pt_fmt == FMT_PROJECTIVE) {            // public: the wire format
      C::fe_load(p.z, src + 2 * NW);
      inf_mask = fe_zero_mask<C>(p.z);
      Fe zz;
      C::fe_mul(p.x, p.x, p.z);                // homogeneous (X : Y : Z) is Jacobian (X Z : Y Z^2 : Z)
      C::fe_sqr(zz, p.z);
      C::fe_mul(p.y, p.y, zz);
    } else {
      u32 z = 0;
#pragma unroll
      for (int w = 0; w < 2 * NW; w++) z |= src[w];
      inf_mask = 0u - (((z | (0u - z)) >> 31) ^ 1u);
      p.z = one;
    }
    {                                          // an identity input: keep the arithmetic on a valid point (G), force the result later
      typename C::Pt g;
      C::pt_generator(g);
      fe_mask_select<C>(p.x, inf_mask, g.x, p.x);
      fe_mask_select<C>(p.y, inf_mask, g.y, p.y);
      fe_mask_select<C>(p.z, inf_mask, one, p.z);
    }
    infs |= (inf_mask & 1u) << s;
    // co-Z chain (jacobian.hpp): 2P with P rewritten to its denominator, then (e + 1) P = e P + P, every step rewriting P again.
    // Entry e holds (x, y) of (e + 1) P over the denominator D_e; its z slot the ratio D_e / D_(e-1) (entries 2 .. 7);
    // entry 0 ends as P over D_7 with D_7 itself in its z slot.  Exception-free: (e -+ 1) P = O is impossible for a point of order n.
    {
      Fe rx, ry, qx, qy, zacc, h;
      jac::coz_double_update<C>(rx, ry, zacc, qx, qy, p, pt_fmt != FMT_PROJECTIVE);       // public: the wire format
      fe_st<C>(ws, entry_chunk<C>(s, 1), rx); fe_st<C>(ws, entry_chunk<C>(s, 1) + CW, ry);
#pragma unroll 1
      for (int e = 2; e < 8; e++) {
        jac::coz_add_update<C>(rx, ry, qx, qy, h);
        fe_st<C>(ws, entry_chunk<C>(s, e), rx); fe_st<C>(ws, entry_chunk<C>(s, e) + CW, ry); fe_st<C>(ws, entry_chunk<C>(s, e) + 2 * CW, h);
        C::fe_mul(zacc, zacc, h);
      }
      fe_st<C>(ws, entry_chunk<C>(s, 0), qx); fe_st<C>(ws, entry_chunk<C>(s, 0) + CW, qy); fe_st<C>(ws, entry_chunk<C>(s, 0) + 2 * CW, zacc);
    }
  }
  // ---- phase B: one inversion for the D_7 of all cnt tables (a zero denominator - input not on the curve - is replaced by
  //      one under a mask so that it cannot poison its neighbours); 1 / D_e = (1 / D_7) prod_{i > e} h_i
  {
    Fe acc = one;
#pragma unroll 1
    for (int s = 0; s < cnt; s++) {
      fe_st<C>(ws, pre_chunk<C, BATCH>(s), acc);
      Fe z;
      fe_ld<C>(z, ws, entry_chunk<C>(s, 0) + 2 * CW);
      fe_mask_select<C>(z, fe_zero_mask<C>(z), one, z);
      C::fe_mul(acc, acc, z);
    }
    Fe ai;
    C::fe_inv(ai, acc);
#pragma unroll 1
    for (int s = cnt - 1; s >= 0; s--) {
      Fe z, zi7, zi, t, t3, pre, x, y, sfx, hh;
      fe_ld<C>(z, ws, entry_chunk<C>(s, 0) + 2 * CW);
      fe_mask_select<C>(z, fe_zero_mask<C>(z), one, z);
      fe_ld<C>(pre, ws, pre_chunk<C, BATCH>(s));
      C::fe_mul(zi7, ai, pre);
      C::fe_mul(ai, ai, z);
      C::fe_sqr(t, zi7);
      C::fe_mul(t3, t, zi7);
#pragma unroll 1
      for (int q = 0; q < 2; q++) {              // entries 7 (8P) and 0 (P rewritten): both over D_7
        const int ch = entry_chunk<C>(s, q ? 0 : 7);
        fe_ld<C>(x, ws, ch); C::fe_mul(x, x, t); fe_st<C>(ws, ch, x);
        fe_ld<C>(y, ws, ch + CW); C::fe_mul(y, y, t3); fe_st<C>(ws, ch + CW, y);
      }
      fe_ld<C>(sfx, ws, entry_chunk<C>(s, 7) + 2 * CW);
#pragma unroll 1
      for (int e = 6; e >= 1; e--) {
        if (e != 6) { fe_ld<C>(hh, ws, entry_chunk<C>(s, e + 1) + 2 * CW); C::fe_mul(sfx, sfx, hh); }      // D_7 / D_e (public: the loop counter)
        const int ch = entry_chunk<C>(s, e);
        C::fe_mul(zi, zi7, sfx);
        C::fe_sqr(t, zi);
        fe_ld<C>(x, ws, ch); C::fe_mul(x, x, t); fe_st<C>(ws, ch, x);
        C::fe_mul(t, t, zi);
        fe_ld<C>(y, ws, ch + CW); C::fe_mul(y, y, t); fe_st<C>(ws, ch + CW, y);
      }
    }
  }
  // ---- phase C: the window loop, one unit after the other; results stay in the lane's table slot (entry 0 is free
  //      once its unit is done: x, y, z of the result overwrite it) until the shared output inversion
  u32 res_inf = 0;
#pragma unroll 1
  for (int b = 0; b < cnt; b++) {
    const size_t i = base + (size_t)b * T;
    u32 k[NW];
    const u32 flip = scalar_fold<C>(k, scalars + i * NW);
    u32 c = 0;
#ifdef ECGPU_DIGITS_IN_REGISTERS                 // A/B switch: NW VGPRs and a select chain per read
    u32 y[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) y[w] = addc(k[w], 0x88888888u, c);
#else
#pragma unroll
    for (int w = 0; w < NW; w++) dm.st(w, addc(k[w], 0x88888888u, c));       // the recoded digits leave the registers (DigitMem)
#endif
    Jac<C> acc;
    jac::set_infinity<C>(acc);
    u32 acc_inf = 0xFFFFFFFFu;                 // mask: the accumulator is still empty
#pragma unroll 1
    for (int j = 8 * NW; j >= 0; j--) {         // position 8 NW holds the carry digit (0 or 1)
      if (j != 8 * NW) {
#pragma unroll 1
        for (int d = 0; d < 4; d++) jac::dbl<C>(acc);
      }
#ifdef ECGPU_DIGITS_IN_REGISTERS
      u32 word = 0;
#pragma unroll
      for (int q = 0; q < NW; q++) word = ((j >> 3) == q) ? y[q] : word;
#else
      const u32 word = dm.ld(j == 8 * NW ? 0 : (j >> 3));                       // the address is public (the loop counter)
#endif
      const u32 nib = (word >> (4 * (j & 7))) & 15u;
      const int sd = (j == 8 * NW) ? (int)c : (int)nib - 8;
      const u32 sgn = (u32)(sd >> 31), mag = ((u32)sd ^ sgn) - sgn;               // |digit| in 0..8
      const u32 zero_digit = 0u - ((mag - 1u) >> 31);
      Fe qx = zero, qy = zero;
#pragma unroll 2
      for (int e = 0; e < 8; e++) {
        ECGPU_TABLE_TOUCH(b * 8 + e);
        const u32 mk = 0u - (((mag ^ (u32)(e + 1)) - 1u) >> 31);                  // all ones iff mag == e + 1
        Fe tx, ty;
        fe_ld<C>(tx, ws, entry_chunk<C>(0, 0) + (b * 8 + e) * 3 * CW);
        fe_ld<C>(ty, ws, entry_chunk<C>(0, 0) + (b * 8 + e) * 3 * CW + CW);
#pragma unroll
        for (int w = 0; w < NW; w++) { qx.v[w] |= tx.v[w] & mk; qy.v[w] |= ty.v[w] & mk; }
      }
      Fe ny;
      C::fe_neg(ny, qy);
      fe_mask_select<C>(qy, sgn ^ flip, ny, qy);       // digit < 0 xor scalar folded: subtract the entry
      Jac<C> s = acc;
      add_mixed_raw<C>(s, qx, qy);
      // empty accumulator: the sum is the entry itself; zero digit: the accumulator stays
      fe_mask_select<C>(s.x, acc_inf, qx, s.x);
      fe_mask_select<C>(s.y, acc_inf, qy, s.y);
      fe_mask_select<C>(s.z, acc_inf, one, s.z);
      fe_mask_select<C>(acc.x, zero_digit, acc.x, s.x);
      fe_mask_select<C>(acc.y, zero_digit, acc.y, s.y);
      fe_mask_select<C>(acc.z, zero_digit, acc.z, s.z);
      acc_inf &= zero_digit;
    }
    const u32 inf = acc_inf | (0u - ((infs >> b) & 1u));
    res_inf |= (inf & 1u) << b;
    fe_mask_select<C>(acc.z, inf, one, acc.z);          // keep the shared inversion clean
    jac_st<C>(ws, entry_chunk<C>(b, 0), acc);
  }
#ifndef ECGPU_DIGITS_IN_REGISTERS
#pragma unroll
  for (int w = 0; w < NW; w++) dm.st(w, 0u);            // the last unit's recoded scalar does not stay in LDS
#endif
  // ---- phase D: x = X / Z^2, y = Y / Z^3 with one inversion for the cnt results of this lane
  {
    Fe run = one;
#pragma unroll 1
    for (int b = 0; b < cnt; b++) {
      Fe z;
      fe_st<C>(ws, pre_chunk<C, BATCH>(b), run);
      fe_ld<C>(z, ws, entry_chunk<C>(b, 0) + 2 * CW);
      // Z = 0 besides the tracked identities: only for input that is not on the curve (a small-order point of another cubic makes the raw
      // addition meet acc = +-Q).  It must not reach the shared product: fe_inv(0) = 0 would wipe every result of this lane's pass.
      const u32 zm = fe_zero_mask<C>(z) | (0u - ((res_inf >> b) & 1u));
      res_inf |= (zm & 1u) << b;
      fe_mask_select<C>(z, zm, one, z);
      C::fe_mul(run, run, z);
    }
    Fe inv;
    C::fe_inv(inv, run);
#pragma unroll 1
    for (int b = cnt - 1; b >= 0; b--) {
      const size_t i = base + (size_t)b * T;
      const u32 inf = 0u - ((res_inf >> b) & 1u);
      Fe z, pre, zi, t, x, yv;
      fe_ld<C>(z, ws, entry_chunk<C>(b, 0) + 2 * CW);
      fe_mask_select<C>(z, inf, one, z);
      fe_ld<C>(pre, ws, pre_chunk<C, BATCH>(b));
      C::fe_mul(zi, inv, pre);
      C::fe_mul(inv, inv, z);
      C::fe_sqr(t, zi);
      fe_ld<C>(x, ws, entry_chunk<C>(b, 0));
      C::fe_mul(x, x, t);
      C::fe_mul(t, t, zi);
      fe_ld<C>(yv, ws, entry_chunk<C>(b, 0) + CW);
      C::fe_mul(yv, yv, t);
      // the parked result and its prefix product are secrets of the same rank as the output: they do not stay in the workspace
      fe_st<C>(ws, entry_chunk<C>(b, 0), zero); fe_st<C>(ws, entry_chunk<C>(b, 0) + CW, zero); fe_st<C>(ws, entry_chunk<C>(b, 0) + 2 * CW, zero);
      fe_st<C>(ws, pre_chunk<C, BATCH>(b), zero);
      fe_mask_select<C>(x, inf, zero, x);
      if (out_fmt == FMT_PROJECTIVE) {          // public: the wire format.  (x : y : 1), identity (0 : 1 : 0)
        Fe zo;
        fe_mask_select<C>(yv, inf, one, yv);
        fe_mask_select<C>(zo, inf, zero, one);
        u32* o = out + i * 3 * NW;
        C::fe_store(o, x); C::fe_store(o + NW, yv); C::fe_store(o + 2 * NW, zo);
      } else {
        fe_mask_select<C>(yv, inf, zero, yv);
        u32* o = out + i * 2 * NW;
        C::fe_store(o, x); C::fe_store(o + NW, yv);
        if (out_inf) out_inf[i] = (uint8_t)(inf & 1u);
      }
    }
  }
}

}  // namespace vbct
}  // namespace ecgpu
