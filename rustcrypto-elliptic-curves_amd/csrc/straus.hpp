// Linear combinations of 3 .. 1024 terms, throughput schedule (Straus / Shamir interleaving), per-lane body (host + device:
// tests/hosttwin walks it on the CPU).
//
// The reference interleaves ALL terms of a combination over ONE chain of 128 doublings (k256/src/arithmetic/mul.rs:342-393:
// per term two 8-entry tables, per digit position four doublings and two table additions per term); the primeorder curves have no
// such entry (their LinearCombination is x*k + y*l, primeorder/src/projective.rs:415-420).  Until round 4 this library ran the
// REFERENCE schedule once per term (complete formulas, 1 984 multiplications per secp256k1 term, 4 361 / 6 473 on P-256 / P-384)
// and folded the products (missing 2).  The group element sum_t k_t P_t is what is specified, so here:
//   * a combination is cut into GROUPS of g <= 16 terms (g = its whole length when that is <= 16): the terms of a group share the
//     doublings of one window loop, a combination of more than 16 terms leaves one Jacobian partial sum per group and a second,
//     small kernel adds them up (general additions, all special cases handled) and writes the outputs with one inversion per 16;
//   * per term ONE table [P .. 8P] as a co-Z chain (jacobian.hpp / mulfast_k256.hpp), all tables of a lane's pass - up to 16 -
//     brought to AFFINE form with one shared inversion (the chain's ratios give the other seven denominators of a table), kept in
//     a lane-contiguous global workspace; secp256k1 reads the lambda half's entry as (beta x, y) - one multiplication on half of
//     the additions instead of a second copy of the table;
//   * signed 4-bit digits (secp256k1: of both GLV halves, 33 positions; P-256 / P-384: of min(k, n - k), 8 NB + 1 positions), zero
//     digits skipped, Jacobian doublings and mixed additions with every exceptional case handled (this is the public-data
//     contract: digit-indexed table reads and data-dependent branches; `flags` with secret scalars never get here).
// Per secp256k1 term: ~117 multiplications for the table, 62 additions of 11 (+ 31 by beta) and 896 / g for the shared doublings:
// ~890 at g = 16 against 1 984 + 12 before.  A lane processes SLOTS / g work items (groups) per pass so that small combinations
// still share the table inversion between 16 tables.
#pragma once
#include "jacobian.hpp"
#include "mulfast_k256.hpp"
#ifdef __HIPCC__
#include "sched.hpp"
#endif

namespace ecgpu {
namespace straus {

constexpr int SLOTS = 16;                      // tables per lane and pass

template <class C> constexpr int digit_words() { return C::A_IS_ZERO ? 8 : C::NW; }          // recoded words per term (two GLV halves of 4)
template <class C> constexpr int positions() { return C::A_IS_ZERO ? 33 : 8 * C::NW + 1; }    // digit positions incl. the carry digit

template <class C>
struct LaneWs {
  AffEntry<C> tab[SLOTS][8];                   // entry e of slot s: (e + 1) P, over the chain's denominator D_e until phase B made it affine
  typename C::Fe zr[SLOTS][8];                 // chain ratios h_e = D_e / D_(e-1) (e = 2 .. 7); [0] = D_7 (times the input's own Z)
  typename C::Fe pre[SLOTS];                   // prefix products of the shared inversion
  u32 dig[SLOTS][digit_words<C>() + 1];        // recoded digit words and a flags word
};
// flags word.  secp256k1: bit 0 / 1 the carry digits of the halves, bit 2 / 3 their signs, bit 4 identity input.
//              P-256 / P-384: bit 0 the carry digit, bit 4 identity input (the fold's sign is already in the table: -P).
constexpr u32 FLAG_SKIP = 16u;

// accumulator type and its two operations: secp256k1 uses the fused forms of mulfast_k256.hpp
template <class C> struct AccOf { using type = Jac<C>; };
template <> struct AccOf<CurveK256> { using type = JacK256; };
ECGPU_HD void acc_dbl(JacK256& p) { k256::jac_double(p); }
ECGPU_HD void acc_add(JacK256& p, const FeK256& x, const FeK256& y) { k256::jac_add_mixed(p, x, y, nullptr); }
template <class C> ECGPU_HD void acc_dbl(Jac<C>& p) { jac::dbl<C>(p); }
template <class C> ECGPU_HD void acc_add(Jac<C>& p, const typename C::Fe& x, const typename C::Fe& y) { jac::add_mixed<C>(p, x, y); }

// first step of the table chain: (rx, ry) = 2P over zacc, (qx, qy) = P rewritten to that denominator; p = (x, y[, z]) Jacobian
template <class C>
ECGPU_HD void chain_start(typename C::Fe& rx, typename C::Fe& ry, typename C::Fe& zacc, typename C::Fe& qx, typename C::Fe& qy, const Jac<C>& p, bool z_is_one) {
  if constexpr (C::A_IS_ZERO) {
    // (x, y) is affine on the curve isomorphic by u = Z (x = X Z, y = Y Z^2 for a homogeneous input): the chain runs there and
    // every denominator picks up the factor Z
    JacK256 d;
    k256::coz_double_affine(d, qx, qy, p.x, p.y);
    rx = d.x; ry = d.y; zacc = d.z;
    if (!z_is_one) k256::mul(zacc, zacc, p.z);
  } else {
    jac::coz_double_update<C>(rx, ry, zacc, qx, qy, p, z_is_one);
  }
}

// Work item w of a call = group (w mod gpc) of combination (w div gpc); a group is `g` consecutive terms (the last one the rest).
// One pass of one lane: items base, base + T, .. (at most SLOTS / g of them, those below `items`); each leaves its Jacobian partial
// sum in partial[w] (3 NW words, internal form).
template <class C>
ECGPU_HD void lane_pass(const u32* scalars, const u32* points, int pt_fmt, int terms, int g, int gpc, size_t items, size_t base, size_t T, LaneWs<C>& ws, u32* partial) {
  constexpr int NW = C::NW, DW = digit_words<C>(), NPOS = positions<C>();
  using Fe = typename C::Fe;
  using Acc = typename AccOf<C>::type;
  const int pw = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * NW;
  const int upp = SLOTS / g;                   // work items per pass
  int cnt = 0;
  u32 used = 0;                                // table slots of this pass that hold a term
  // ---- phase A: digits and table chains of every term of this pass
#pragma unroll 1
  for (int b = 0; b < upp; b++) {
    const size_t w = base + (size_t)b * T;
    if (w >= items) break;
    cnt = b + 1;
    const size_t combo = w / (size_t)gpc;
    const int t0 = (int)(w % (size_t)gpc) * g;
    const int tn = (terms - t0 < g) ? terms - t0 : g;
#pragma unroll 1
    for (int tt = 0; tt < tn; tt++) {
      const int s = b * g + tt;
      const size_t term = combo * (size_t)terms + (size_t)(t0 + tt);
      used |= 1u << s;
      u32 k[NW], ord[NW];
      C::scalar_load(k, scalars + term * NW);
      C::order(ord);
      reduce_once<NW>(k, ord);
      u32 flags = 0;
      bool flip = false;
      if constexpr (C::A_IS_ZERO) {
        k256::GlvSplit sp;
        k256::glv_split(sp, k);
        k256::Radix16<4> d1, d2;
        k256::radix16_recode<4>(d1, sp.k1);
        k256::radix16_recode<4>(d2, sp.k2);
#pragma unroll
        for (int q = 0; q < 4; q++) { ws.dig[s][q] = d1.y[q]; ws.dig[s][4 + q] = d2.y[q]; }
        flags = (d1.top & 1u) | ((d2.top & 1u) << 1) | (sp.neg1 ? 4u : 0u) | (sp.neg2 ? 8u : 0u);
      } else {
        u32 t[NW];
        mp_sub<NW>(t, ord, k);
        flip = !mp_geq<NW>(t, k);              // n - k < k: use n - k and -P
        u32 c = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) ws.dig[s][q] = addc(flip ? t[q] : k[q], 0x88888888u, c);
        flags = c & 1u;
      }
      // the point -> Jacobian (a homogeneous X : Y : Z is Jacobian X Z : Y Z^2 : Z)
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
        for (int q = 0; q < 2 * NW; q++) z |= src[q];
        p_inf = (z == 0);
        C::fe_one(p.z);
      }
      if (p_inf) {                              // keep the arithmetic on a valid point; the term is left out of the loop
        typename C::Pt gpt;
        C::pt_generator(gpt);
        p.x = gpt.x; p.y = gpt.y; C::fe_one(p.z);
        flags |= FLAG_SKIP;
      }
      if (flip) C::fe_neg(p.y, p.y);
      ws.dig[s][DW] = flags;
      Fe rx, ry, qx, qy, zacc, h;
      chain_start<C>(rx, ry, zacc, qx, qy, p, pt_fmt != FMT_PROJECTIVE || p_inf);
      ws.tab[s][1].x = rx; ws.tab[s][1].y = ry;
#pragma unroll 1
      for (int e = 2; e < 8; e++) {
        jac::coz_add_update<C>(rx, ry, qx, qy, h);
        ws.tab[s][e].x = rx; ws.tab[s][e].y = ry; ws.zr[s][e] = h;
        C::fe_mul(zacc, zacc, h);              // D_e = D_(e-1) h_e
      }
      ws.tab[s][0].x = qx; ws.tab[s][0].y = qy; ws.zr[s][0] = zacc;       // P over D_7, and D_7 itself
    }
  }
  // ---- phase B: ONE inversion for the D_7 of all tables of this pass (a zero denominator - only for input that is not on the curve -
  //      is replaced by one so that it cannot poison its neighbours); 1 / D_e = (1 / D_7) prod_{i > e} h_i
  {
    Fe acc; C::fe_one(acc);
#pragma unroll 1
    for (int s = 0; s < SLOTS; s++) {
      if (!((used >> s) & 1u)) continue;
      ws.pre[s] = acc;
      Fe z = ws.zr[s][0];
      if (C::fe_is_zero(z)) C::fe_one(z);
      C::fe_mul(acc, acc, z);
    }
    Fe ai;
    C::fe_inv(ai, acc);
#pragma unroll 1
    for (int s = SLOTS - 1; s >= 0; s--) {
      if (!((used >> s) & 1u)) continue;
      AffEntry<C>* tab = ws.tab[s];
      Fe z = ws.zr[s][0], zi7, zi, t, sfx;
      if (C::fe_is_zero(z)) C::fe_one(z);
      C::fe_mul(zi7, ai, ws.pre[s]);
      C::fe_mul(ai, ai, z);
      C::fe_sqr(t, zi7);
      C::fe_mul(tab[7].x, tab[7].x, t);
      C::fe_mul(tab[0].x, tab[0].x, t);
      C::fe_mul(t, t, zi7);
      C::fe_mul(tab[7].y, tab[7].y, t);
      C::fe_mul(tab[0].y, tab[0].y, t);
#pragma unroll 1
      for (int e = 6; e >= 1; e--) {
        if (e == 6) sfx = ws.zr[s][7]; else C::fe_mul(sfx, sfx, ws.zr[s][e + 1]);      // D_7 / D_e
        C::fe_mul(zi, zi7, sfx);
        C::fe_sqr(t, zi);
        C::fe_mul(tab[e].x, tab[e].x, t);
        C::fe_mul(t, t, zi);
        C::fe_mul(tab[e].y, tab[e].y, t);
      }
    }
  }
  // ---- phase C: the window loop of every work item; its terms share the doublings
  Fe beta_;
  if constexpr (C::A_IS_ZERO) k256::beta(beta_);
#pragma unroll 1
  for (int b = 0; b < cnt; b++) {
    const size_t w = base + (size_t)b * T;
    const int t0 = (int)(w % (size_t)gpc) * g;
    const int tn = (terms - t0 < g) ? terms - t0 : g;
    Acc acc;
    C::fe_zero(acc.x); C::fe_zero(acc.y); C::fe_zero(acc.z);                // infinity
#pragma unroll 1
    for (int j = NPOS - 1; j >= 0; j--) {
      if (j != NPOS - 1) {
#pragma unroll 1
        for (int d = 0; d < 4; d++) acc_dbl(acc);
      }
#pragma unroll 1
      for (int tt = 0; tt < tn; tt++) {
        const int s = b * g + tt;
        const u32 flags = ws.dig[s][DW];
        if (flags & FLAG_SKIP) continue;
        if constexpr (C::A_IS_ZERO) {
#pragma unroll 1
          for (int h = 0; h < 2; h++) {
            int sd;
            if (j == NPOS - 1) sd = (int)((flags >> h) & 1u);
            else sd = k256::radix16_digit(ws.dig[s][4 * h + (j >> 3)], j & 7);
            if (sd == 0) continue;
            const AffEntry<C>& e = ws.tab[s][(sd < 0 ? -sd : sd) - 1];
            Fe ex = e.x, ey = e.y;
            if (h) k256::mul(ex, ex, beta_);                                // lambda (x, y) = (beta x, y)
            if ((((flags >> (2 + h)) & 1u) != 0) != (sd < 0)) C::fe_neg(ey, ey);
            acc_add(acc, ex, ey);
          }
        } else {
          int sd;
          if (j == NPOS - 1) sd = (int)(flags & 1u);
          else sd = (int)((ws.dig[s][j >> 3] >> (4 * (j & 7))) & 15u) - 8;
          if (sd == 0) continue;
          const AffEntry<C>& e = ws.tab[s][(sd < 0 ? -sd : sd) - 1];
          Fe ex = e.x, ey = e.y;
          if (sd < 0) C::fe_neg(ey, ey);
          acc_add(acc, ex, ey);
        }
      }
    }
    u32* o = partial + w * 3 * NW;
#pragma unroll
    for (int q = 0; q < NW; q++) { o[q] = acc.x.v[q]; o[NW + q] = acc.y.v[q]; o[2 * NW + q] = acc.z.v[q]; }
  }
}

// Second stage, one pass of one lane: combinations base, base + T, .. (at most BATCH, those below n): the gpc partial sums of each
// are added up (general Jacobian additions, all special cases handled) and the results written with one shared inversion.
template <class C, int BATCH>
ECGPU_HD void fold_pass(const u32* partial, int gpc, u32* out, int out_fmt, uint8_t* out_inf, size_t n, size_t base, size_t T) {
  constexpr int NW = C::NW;
  Jac<C> res[BATCH];
  typename C::Fe pre[BATCH];
  int cnt = 0;
#pragma unroll 1
  for (int b = 0; b < BATCH; b++) {
    const size_t i = base + (size_t)b * T;
    if (i >= n) break;
    cnt = b + 1;
    Jac<C> acc;
#pragma unroll 1
    for (int j = 0; j < gpc; j++) {
      const u32* src = partial + (i * (size_t)gpc + (size_t)j) * 3 * NW;
      Jac<C> q;
#pragma unroll
      for (int w = 0; w < NW; w++) { q.x.v[w] = src[w]; q.y.v[w] = src[NW + w]; q.z.v[w] = src[2 * NW + w]; }
      if (j == 0) acc = q; else jac::add<C>(acc, acc, q);
    }
    res[b] = acc;
  }
  jac::store_batch_affine<C>(res, pre, cnt, base, T, out, out_fmt, out_inf);
}

#ifdef __HIPCC__
template <class C, int WAVES>
__global__ void __launch_bounds__(256, WAVES) lincomb_kernel(const u32* scalars, const u32* points, int pt_fmt, int terms, int g, int gpc, size_t items, LaneWs<C>* ws_all,
                                                             u32* partial, WaveSched sched) {
  LaneWs<C>& ws = ws_all[(size_t)blockIdx.x * blockDim.x + threadIdx.x];
  // every wave draws whole passes (SLOTS / g work items per lane: their tables share one inversion) from one counter (sched.hpp): with more passes than
  // resident lanes the favoured waves of a SIMD take more of them instead of idling at the end
  size_t lo, hi;
  while (wave_next_chunk(sched, lo, hi)) lane_pass<C>(scalars, points, pt_fmt, terms, g, gpc, hi, lo + (threadIdx.x & 63u), 64, ws, partial);
}
template <class C>
__global__ void __launch_bounds__(256) fold_kernel(const u32* partial, int gpc, u32* out, int out_fmt, uint8_t* out_inf, size_t n) {
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * 16) fold_pass<C, 16>(partial, gpc, out, out_fmt, out_inf, n, base, T);
}
#endif

// how a call of n combinations of `terms` terms is cut for `lanes` resident lanes: the largest group size g <= min(16, terms) that
// still gives every lane a work item, balanced over the groups of a combination (17 terms: 9 + 8, not 16 + 1)
static inline void plan(size_t n, size_t terms, size_t lanes, int* g_out, int* gpc_out) {
  size_t g = terms < (size_t)SLOTS ? terms : (size_t)SLOTS;
  while (g > 1 && n * ((terms + g - 1) / g) < lanes) g--;
  const size_t gpc = (terms + g - 1) / g;
  g = (terms + gpc - 1) / gpc;
  *g_out = (int)g;
  *gpc_out = (int)gpc;
}

}  // namespace straus
}  // namespace ecgpu
