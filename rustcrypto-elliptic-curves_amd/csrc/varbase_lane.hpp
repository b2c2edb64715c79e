// Variable-base scalar multiplication k*P, throughput schedule for the curves without an efficient
// endomorphism (P-256, P-384).  The reference computes this with complete homogeneous formulas, an
// unsigned 4-bit window and a 16-entry table (primeorder/src/projective.rs:106-150: 4361 / 6473
// field multiplications); the group element is what is specified, so here:
//   * Jacobian coordinates (doubling 4M+4S with a halving for a = -3, general addition 11M+5S);
//   * signed 4-bit digits (k > n/2 is replaced by n - k and -P): table [P .. 8P] per lane in a lane-contiguous global
//     workspace, built as a co-Z chain (one doubling with update, six co-Z additions: jacobian.hpp) and brought to affine
//     form through ONE inverted denominator per table (the chain's ratios give the other seven);
//   * per-lane batched conversion to affine (one inversion per BATCH results).
// (The common-Z "effective affine" table used for k256 needs a = 0: on the isomorphic curve the
// a = -3 doubling shortcut no longer holds.)
// This file is the per-lane body (host + device: tests/hosttwin walks it on the CPU with a small lane count so that
// every slot count 1..BATCH and several passes are checked without a GPU); varbase.hpp wraps it in the kernel.
#pragma once
#include "jacobian.hpp"

namespace ecgpu {
namespace vb {

template <class C> constexpr int nwin() { return 2 * C::NB + 1; }   // nibbles + the carry digit

// workspace of one lane: BATCH tables of 8 points - entry e is (e + 1) P as (x, y) over the denominator D_e of the co-Z chain,
// overwritten by the affine x, y; its z slot holds the chain's ratio D_e / D_(e-1) (entries 2 .. 7), D_1 (entry 1) and D_7
// (entry 0, whose x, y are P rewritten to D_7) - and the BATCH prefix products of the shared inversion
template <class C, int BATCH, int WB = 4>
struct LaneWs {
  Jac<C> tab[BATCH][1 << (WB - 1)];
  typename C::Fe pre[BATCH];
};
// WB = window width in bits: 4 (8 table entries, 8 NW + 1 digit positions: signed nibbles of k + 0x88..8) or 5 (16 entries, ceil((32 NW + 1) / 5)
// positions: 5-bit fields of k + 0x..10842 1084 2108 4210, one word more than the scalar).  Round 4, P-384: a table entry costs ~13 multiplications
// since the co-Z chains, an addition 11: 77 windows and 8 more entries instead of 97 windows save ~75 of 4 200 multiplications.
template <class C, int WB> constexpr int positions() { return WB == 4 ? 8 * C::NW + 1 : (32 * C::NW + 1 + WB - 1) / WB; }
template <class C, int WB> constexpr int digit_words() { return WB == 4 ? C::NW : C::NW + 1; }

// One pass of one lane: units base, base + T, .., base + (BATCH / NT - 1) T (those below n).  A unit is a linear
// combination of NT terms (NT = 1: k * P; NT = 2: LinearCombination::lincomb, k P + l Q, primeorder/src/projective.rs:
// 415-420 - the reference computes the two products separately; here they share the doublings of one window loop);
// term t of unit i is scalar / point number i * NT + t, and the NT tables of a unit take NT of the BATCH table slots.
// Per-lane tables live in a lane-contiguous global workspace (in the private segment a lane-divergent index turns
// every entry read into scattered dword rows: 3.5x the fetch traffic on the k256 kernel).
// The BATCH tables of a pass are built first and brought to affine form with ONE inversion, so that every addition of the
// main loop is a mixed one (8M + 3S instead of 11M + 5S).  Until late in round 3 a table was 4 doublings and 3 general
// additions (80 multiplications) and the inversion ran over all BATCH * 8 denominators (7 multiplications per entry); now
// the chain is co-Z (8 + 6 x 6 + 6 for the last denominator), Montgomery's trick runs over ONE denominator per table and
// the other entries' inverses follow from the chain's ratios (5 multiplications per entry): ~95 instead of ~136 per table.
// Measured against Jacobian tables with general additions: +8.5 % for P-384 (16.6 against 15.3 M/s at 2^21), +3.7 % for
// P-256 (53.2 against 51.3; in round 1, with a square-and-multiply inversion of 384 multiplications instead of the
// 267-multiplication chain, the extra pass over the tables cost P-256 more than the cheaper additions saved).
// `res_ext` / `idx_ext` / `cnt_ext` (optional, all or none): instead of converting its results to affine and storing them, the pass APPENDS them
// (Jacobian) with their global indices to the caller's per-lane buffer - the dynamically scheduled kernel draws small passes (the TABLE inversion
// is shared by the units of one pass) and flushes the buffer with one OUTPUT inversion whenever it is full (varbase.hpp, sched.hpp).
template <class C, int BATCH, int NT = 1, int WB = 4>
ECGPU_HD void lane_pass(const u32* scalars, const u32* points, int pt_fmt, u32* out, int out_fmt, uint8_t* out_inf, size_t n, size_t base,
                        size_t T, LaneWs<C, BATCH, WB>& ws, const DigitMem& dm, Jac<C>* res_ext = nullptr, size_t* idx_ext = nullptr, int* cnt_ext = nullptr) {
  static_assert(BATCH % NT == 0 && BATCH <= 32, "table slots per pass");
  static_assert(WB == 4 || WB == 5, "window width");
  constexpr int NW = C::NW;
  constexpr int NE = 1 << (WB - 1);            // table entries [P .. NE P]
  constexpr int DW = digit_words<C, WB>();     // recoded words per term kept in DigitMem
  constexpr int NPOS = positions<C, WB>();
  constexpr int UB = BATCH / NT;               // units per pass
  using Fe = typename C::Fe;
  Jac<C> res[UB];
  Fe pre[UB];
  const int pw = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * NW;
  int cnt = 0;
  u32 flips = 0, infs = 0;                     // one bit per table slot
  // ---- phase A: Jacobian tables [P .. 8P] of the terms of this pass
#pragma unroll 1
  for (int s = 0; s < BATCH; s++) {
    const int b = s / NT;
    const size_t i = base + (size_t)b * T;
    if (i >= n) break;
    cnt = b + 1;
    const size_t term = i * NT + (size_t)(s % NT);
    u32 k[NW], ord[NW], t[NW];
    C::scalar_load(k, scalars + term * NW);
    C::order(ord);
    reduce_once<NW>(k, ord);
    mp_sub<NW>(t, ord, k);
    const bool flip = !mp_geq<NW>(t, k);          // n - k < k: use n - k and -P
    // input point -> Jacobian (homogeneous X:Y:Z is Jacobian XZ : YZ^2 : Z)
    const u32* src = points + term * pw;
    Jac<C> p;
    C::fe_load(p.x, src);
    C::fe_load(p.y, src + NW);
    bool p_inf;
    if (pt_fmt == FMT_PROJECTIVE) {
      C::fe_load(p.z, src + 2 * NW);
      p_inf = C::fe_is_zero(p.z);
      Fe zz;
      C::fe_mul(p.x, p.x, p.z);
      C::fe_sqr(zz, p.z);
      C::fe_mul(p.y, p.y, zz);
    } else {
      u32 z = 0;
#pragma unroll
      for (int w = 0; w < 2 * NW; w++) z |= src[w];
      p_inf = (z == 0);
      C::fe_one(p.z);
    }
    if (p_inf) {                                // keep the arithmetic on a valid point; the term is left out below
      typename C::Pt g;
      C::pt_generator(g);
      p.x = g.x; p.y = g.y; C::fe_one(p.z);
    }
    if (flip) C::fe_neg(p.y, p.y);
    flips |= (flip ? 1u : 0u) << s;
    infs |= (p_inf ? 1u : 0u) << s;
    // co-Z chain: 2P with P rewritten to its denominator, then (e + 1) P = e P + P, every step rewriting P again
    Jac<C>* tab = ws.tab[s];
    Fe rx, ry, qx, qy, zacc, h;
    jac::coz_double_update<C>(rx, ry, zacc, qx, qy, p, pt_fmt != FMT_PROJECTIVE);
    tab[1].x = rx; tab[1].y = ry;
#pragma unroll 1
    for (int e = 2; e < NE; e++) {
      jac::coz_add_update<C>(rx, ry, qx, qy, h);
      tab[e].x = rx; tab[e].y = ry; tab[e].z = h;
      C::fe_mul(zacc, zacc, h);                // D_e = D_(e-1) h_e
    }
    tab[0].x = qx; tab[0].y = qy; tab[0].z = zacc;       // P over D_(NE-1), and D_(NE-1) itself
  }
  // ---- phase B: one inversion for the D_7 of all cnt * NT tables (a zero denominator - only possible for input that is
  //      not on the curve - is replaced by one so that it cannot poison its neighbours); 1 / D_e = (1 / D_7) prod_{i > e} h_i
  {
    Fe acc; C::fe_one(acc);
#pragma unroll 1
    for (int s = 0; s < cnt * NT; s++) {
      ws.pre[s] = acc;
      Fe z = ws.tab[s][0].z;
      if (C::fe_is_zero(z)) C::fe_one(z);
      C::fe_mul(acc, acc, z);
    }
    Fe ai;
    C::fe_inv(ai, acc);
#pragma unroll 1
    for (int s = cnt * NT - 1; s >= 0; s--) {
      Jac<C>* tab = ws.tab[s];
      Fe z = tab[0].z, zi7, zi, t, sfx;
      if (C::fe_is_zero(z)) C::fe_one(z);
      C::fe_mul(zi7, ai, ws.pre[s]);
      C::fe_mul(ai, ai, z);
      C::fe_sqr(t, zi7);
      C::fe_mul(tab[NE - 1].x, tab[NE - 1].x, t);
      C::fe_mul(tab[0].x, tab[0].x, t);
      C::fe_mul(t, t, zi7);
      C::fe_mul(tab[NE - 1].y, tab[NE - 1].y, t);
      C::fe_mul(tab[0].y, tab[0].y, t);
#pragma unroll 1
      for (int e = NE - 2; e >= 1; e--) {
        if (e == NE - 2) sfx = tab[NE - 1].z; else C::fe_mul(sfx, sfx, tab[e + 1].z);      // D_(NE-1) / D_e
        C::fe_mul(zi, zi7, sfx);
        C::fe_sqr(t, zi);
        C::fe_mul(tab[e].x, tab[e].x, t);
        C::fe_mul(t, t, zi);
        C::fe_mul(tab[e].y, tab[e].y, t);
      }
    }
  }
  // ---- phase C: signed 4-bit windows over the tables; the NT terms of a unit share the doublings
#pragma unroll 1
  for (int b = 0; b < cnt; b++) {
    const size_t i = base + (size_t)b * T;
    // signed nibbles: digit_j = nibble_j(k + 0x88..8) - 8, the carry out of the top nibble is the last digit
    // (the recoded words go to DigitMem - LDS on the device - instead of NT * NW registers held across the whole loop)
    u32 carry[NT];
#ifdef ECGPU_DIGITS_IN_REGISTERS                 // A/B switch: the round-2 form (NT * NW VGPRs and a select chain per read)
    u32 y[NT][NW];
#endif
#pragma unroll
    for (int tt = 0; tt < NT; tt++) {
      const int s = b * NT + tt;
      u32 k[NW], ord[NW], t[NW];
      C::scalar_load(k, scalars + (i * NT + tt) * NW);
      C::order(ord);
      reduce_once<NW>(k, ord);
      if ((flips >> s) & 1u) { mp_sub<NW>(t, ord, k); mp_copy<NW>(k, t); }
      u32 c = 0;
      const bool skip = (infs >> s) & 1u;       // an identity input contributes nothing: all its digits read as zero
      if constexpr (WB == 4) {
#pragma unroll
        for (int w = 0; w < NW; w++) {
          const u32 yw = addc(k[w], 0x88888888u, c);
#ifdef ECGPU_DIGITS_IN_REGISTERS
          y[tt][w] = skip ? 0x88888888u : yw;
#else
          dm.st(tt * NW + w, skip ? 0x88888888u : yw);
#endif
        }
        carry[tt] = skip ? 0u : c;
      } else {
        // 5-bit fields: t = k + sum_j 16 * 32^j, digit_j = field_j(t) - 16 in [-16, 15]; k <= n / 2 < 2^(32 NW - 1) keeps t below 32^NPOS
#pragma unroll
        for (int w = 0; w <= NW; w++) {
          // word w of the constant 0x...4210842108421084 2108...: bit b is set iff b = 4 (mod 5), for b below 5 NPOS
          u32 cw = 0;
#pragma unroll
          for (int b = 0; b < 32; b++) cw |= (((32 * w + b) % 5 == 4) && (32 * w + b < 5 * NPOS)) ? (1u << b) : 0u;
          const u32 yw = addc(w < NW ? k[w] : 0u, cw, c);
          dm.st(tt * DW + w, skip ? cw : yw);
        }
        carry[tt] = 0u;
      }
    }
    Jac<C> acc;
    jac::set_infinity<C>(acc);
#pragma unroll 1
    for (int j = NPOS - 1; j >= 0; j--) {       // WB = 4: position 8 NW holds the carry digits (0 or 1)
      if (j != NPOS - 1) {
#pragma unroll 1
        for (int d = 0; d < WB; d++) jac::dbl<C>(acc);
      }
#pragma unroll 1
      for (int tt = 0; tt < NT; tt++) {
        int sd;
        if constexpr (WB == 4) {
          if (j == 8 * NW) {
            sd = 0;
#pragma unroll
            for (int q = 0; q < NT; q++) sd = (q == tt) ? (int)carry[q] : sd;
          } else {
#ifdef ECGPU_DIGITS_IN_REGISTERS
            u32 word = y[0][0];
#pragma unroll
            for (int r = 0; r < NT; r++)
#pragma unroll
              for (int q = 0; q < NW; q++) word = (r == tt && (j >> 3) == q) ? y[r][q] : word;
#else
            const u32 word = dm.ld(tt * NW + (j >> 3));
#endif
            sd = (int)((word >> (4 * (j & 7))) & 15u) - 8;
          }
        } else {
          // field j of the recoded words: bits 5 j .. 5 j + 4, possibly across a word boundary (the addresses are wave-uniform)
          const int bit = 5 * j, wi = bit >> 5, sh = bit & 31;
          const u32 lo = dm.ld(tt * DW + wi);
          const u32 hi = (sh > 27) ? dm.ld(tt * DW + wi + 1) : 0u;      // wi + 1 <= NW: the top field ends inside word NW
          sd = (int)((u32)((((u64)hi << 32) | lo) >> sh) & 31u) - 16;
        }
        if (sd != 0) {
          const Jac<C>& e = ws.tab[b * NT + tt][(sd < 0 ? -sd : sd) - 1];
          Fe ex = e.x, ey = e.y;
          if (sd < 0) C::fe_neg(ey, ey);
          jac::add_mixed<C>(acc, ex, ey);
        }
      }
    }
    if (res_ext) { res_ext[*cnt_ext] = acc; idx_ext[*cnt_ext] = i; (*cnt_ext)++; }
    else res[b] = acc;
  }
  if (!res_ext) jac::store_batch_affine<C>(res, pre, cnt, base, T, out, out_fmt, out_inf);
}

}  // namespace vb
}  // namespace ecgpu
