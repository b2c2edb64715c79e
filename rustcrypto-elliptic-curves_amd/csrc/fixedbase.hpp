// Fixed-base scalar multiplication k*G, throughput schedule (all curves).
//
// MulByGenerator in the reference: k256 uses 33 lazily built tables of [1..8] * 2^(8i) G and 66
// complete additions (k256/src/arithmetic/mul.rs:396-440); primeorder curves just compute G * k with
// the variable-base window method (primeorder/src/projective.rs:422-431, "TODO: precomputed basepoint
// tables").  The group element k*G is what is specified, so here:
//   * one table per context in HBM (L2-resident: 33 x 128 affine points = 270 KB for 256-bit curves):
//     T[j][d-1] = d * 2^(8j) * G for d = 1..128, built on the device when the context first needs it;
//   * signed 8-bit digits of k (k > n/2 is replaced by n - k and the result negated, so the carry
//     window is almost never touched): one Jacobian mixed addition per non-zero digit, no doublings;
//   * per-lane batched conversion to affine (one inversion per BATCH results).
// Device code only.
#pragma once
#include "jacobian.hpp"
#include "fixedbase_ct.hpp"
#include "kernels.hpp"
#include "sched.hpp"
#include "msm.hpp"                // XYZZ accumulators (8M + 2S per mixed addition)

namespace ecgpu {
namespace fb {

constexpr int W = 8;
constexpr int ENTRIES = 1 << (W - 1);                                   // |digit| in 1..128
template <class C> constexpr int nwin() { return C::NB + 1; }            // one byte per window + the carry window
// wide variants: WB-bit windows, 2^(WB-1) entries per window, 8 NB / WB additions.
//   WB = 16: 17 x 2^15 entries (36 MB per 256-bit curve, served from the Infinity Cache)      - batches >= 2^18
//   WB = 20: 13 x 2^19 entries (436 MB per 256-bit curve, 1 GB for p384, HBM-resident gathers) - batches >= 2^21
//   WB = 24: 11 x 2^23 entries (5.9 GB per 256-bit curve, 13.7 GB for p384; built in ~30 ms)  - batches >= 2^23
//   WB = 26: 10 x 2^25 entries (21.5 GB per 256-bit curve; built in ~85 ms)                   - batches >= 2^24, not p384
//   (ECGPU_FB_MAX_WINDOW caps the width a context will build, ECGPU_FB_WINDOW pins it)
//   (p256, 2^24 results: 16.1 / 14.2 / 13.3 ms with WB = 20 / 24 / 26 - every addition a window saves is 11 of ~140
//   multiplications, and the gathers from a table no cache holds stay hidden behind them)
// When WB divides the scalar width the signed recoding can carry out of the top window (one extra window that only
// ever sees digit 1); otherwise the top window has spare bits and absorbs the carry.
template <int WB> constexpr int wide_entries() { return 1 << (WB - 1); }
template <class C, int WB> constexpr bool wide_carry_window() { return (8 * C::NB) % WB == 0; }
template <class C, int WB> constexpr int nwin_wide() { return (8 * C::NB + WB - 1) / WB + (wide_carry_window<C, WB>() ? 1 : 0); }

// stage A: one lane per window computes d * 2^(8j) G, d = 1..128, in Jacobian coordinates
template <class C>
__global__ void __launch_bounds__(64) table_jac_kernel(Jac<C>* tmp) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nwin<C>()) return;
  typename C::Pt g;
  C::pt_generator(g);                     // affine generator, Z = 1
  Jac<C> base;
  base.x = g.x; base.y = g.y; C::fe_one(base.z);
#pragma unroll 1
  for (int i = 0; i < W * j; i++) jac::dbl<C>(base);
  Jac<C> acc = base;
  tmp[(size_t)j * ENTRIES] = acc;
#pragma unroll 1
  for (int d = 1; d < ENTRIES; d++) {
    jac::add<C>(acc, acc, base);
    tmp[(size_t)j * ENTRIES + d] = acc;
  }
}
// stage B: one lane per entry converts to affine (own inversion; this runs once per context)
template <class C>
__global__ void __launch_bounds__(256) table_affine_kernel(const Jac<C>* tmp, AffEntry<C>* table, int total) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  typename C::Fe zi, t;
  C::fe_inv(zi, tmp[e].z);
  C::fe_sqr(t, zi);
  C::fe_mul(table[e].x, tmp[e].x, t);
  C::fe_mul(t, t, zi);
  C::fe_mul(table[e].y, tmp[e].y, t);
}

// The cnt XYZZ results of one lane (element j is global index base + j * stride) to affine with ONE inversion: with
// w = ZZ ZZZ, 1 / ZZ = ZZZ / w and 1 / ZZZ = ZZ / w (8M per result and a share of the inversion; the Jacobian epilogue,
// jac::store_batch_affine, takes 6M + 1S - the one multiplication more is paid once per result, the squaring the XYZZ
// addition saves is saved once per addition).  Same output formats as jac::store_batch_affine.
template <class C>
ECGPU_HD void store_batch_affine_xyzz(const msm::Xyzz<C>* res, typename C::Fe* pre, int cnt, size_t base, size_t stride, u32* out, int out_fmt,
                                      uint8_t* out_inf, const size_t* idx = nullptr) {
  using Fe = typename C::Fe;
  constexpr int NW = C::NW;
  Fe acc; C::fe_one(acc);
#pragma unroll 1
  for (int j = 0; j < cnt; j++) {
    pre[j] = acc;
    if (C::fe_is_zero(res[j].zz)) continue;
    Fe w;
    C::fe_mul(w, res[j].zz, res[j].zzz);
    C::fe_mul(acc, acc, w);
  }
  Fe ai;
  C::fe_inv(ai, acc);
#pragma unroll 1
  for (int j = cnt - 1; j >= 0; j--) {
    const size_t i = idx ? idx[j] : base + (size_t)j * stride;
    Fe one, zero, wi, t, x, y;
    C::fe_one(one); C::fe_zero(zero);
    const bool zr = C::fe_is_zero(res[j].zz);
    if (zr) { x = zero; y = zero; }
    else {
      C::fe_mul(wi, ai, pre[j]);                  // 1 / (ZZ ZZZ)
      C::fe_mul(t, res[j].zz, res[j].zzz);
      C::fe_mul(ai, ai, t);
      C::fe_mul(t, wi, res[j].zzz);               // 1 / ZZ
      C::fe_mul(x, res[j].x, t);
      C::fe_mul(t, wi, res[j].zz);                // 1 / ZZZ
      C::fe_mul(y, res[j].y, t);
    }
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

// The accumulator of the digit-indexed fixed-base kernels: XYZZ (8M + 2S per addition of an affine entry, this loop has no doubling to
// make the fourth coordinate expensive) where the four coordinates fit the register budget - the 256-bit curves: signing with
// public nonces +3 %, config 3 +1.3 % -, Jacobian (8M + 3S) for P-384, where 48 accumulator registers instead of 36 spill
// (measured: 5.66 against 5.27 ms per 2^20 with XYZZ; tools/ab_round3d.sh).
template <class C, bool XYZZ = (C::NW <= 8)>
struct FbAcc {
  using Pt = msm::Xyzz<C>;
  static ECGPU_HD void set_infinity(Pt& p) { msm::xyzz_set_infinity<C>(p); }
  static ECGPU_HD void add_mixed(Pt& p, const typename C::Fe& x, const typename C::Fe& y) { msm::xyzz_add_mixed<C>(p, x, y); }
  static ECGPU_HD void add_affine(Pt& p, const typename C::Fe& x, const typename C::Fe& y) { msm::xyzz_add_affine<C>(p, x, y); }
  static ECGPU_HD void store(const Pt* res, typename C::Fe* pre, int cnt, size_t base, size_t stride, u32* out, int out_fmt, uint8_t* out_inf,
                             const size_t* idx = nullptr) {
    store_batch_affine_xyzz<C>(res, pre, cnt, base, stride, out, out_fmt, out_inf, idx);
  }
};
template <class C>
struct FbAcc<C, false> {
  using Pt = Jac<C>;
  static ECGPU_HD void set_infinity(Pt& p) { jac::set_infinity<C>(p); }
  static ECGPU_HD void add_mixed(Pt& p, const typename C::Fe& x, const typename C::Fe& y) { jac::add_mixed<C>(p, x, y); }
  static ECGPU_HD void add_affine(Pt& p, const typename C::Fe& x, const typename C::Fe& y) { jac::add_affine<C>(p, x, y); }
  static ECGPU_HD void store(const Pt* res, typename C::Fe* pre, int cnt, size_t base, size_t stride, u32* out, int out_fmt, uint8_t* out_inf,
                             const size_t* idx = nullptr) {
    jac::store_batch_affine<C>(res, pre, cnt, base, stride, out, out_fmt, out_inf, idx);
  }
};

template <class C, int BATCH, int WAVES>
__global__ void __launch_bounds__(256, WAVES) mul_kernel(const u32* scalars, const AffEntry<C>* table, u32* out, int out_fmt,
                                                         uint8_t* out_inf, size_t n) {
  constexpr int NW = C::NW;
  typename FbAcc<C>::Pt res[BATCH];
  typename C::Fe pre[BATCH];
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * BATCH) {
    int cnt = 0;
#pragma unroll 1
    for (int b = 0; b < BATCH; b++) {
      const size_t i = base + (size_t)b * T;
      if (i >= n) break;
      u32 k[NW], ord[NW], t[NW];
      C::scalar_load(k, scalars + i * NW);
      C::order(ord);
      reduce_once<NW>(k, ord);
      mp_sub<NW>(t, ord, k);
      const bool flip = !mp_geq<NW>(t, k);          // n - k < k
#pragma unroll
      for (int w = 0; w < NW; w++) k[w] = flip ? t[w] : k[w];
      typename FbAcc<C>::Pt acc;
      FbAcc<C>::set_infinity(acc);
      u32 carry = 0;
#pragma unroll 1
      for (int j = 0; j < nwin<C>(); j++) {
        u32 word = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) word = (j >> 2) == q ? k[q] : word;
        u32 d = ((j < C::NB) ? ((word >> (8 * (j & 3))) & 0xFFu) : 0u) + carry;
        carry = (d >= 0x80u) ? 1u : 0u;
        const int sd = (int)d - (int)(carry << 8);
        if (sd != 0) {
          const AffEntry<C>* e = table + (size_t)j * ENTRIES + ((sd < 0 ? -sd : sd) - 1);
          typename C::Fe x = e->x, y = e->y;
          if ((sd < 0) != flip) C::fe_neg(y, y);
          FbAcc<C>::add_mixed(acc, x, y);
        }
      }
      res[b] = acc;
      cnt = b + 1;
    }
    FbAcc<C>::store(res, pre, cnt, base, T, out, out_fmt, out_inf);
  }
}


// canonical x||y bytes -> table entries in the internal field form (used to build the wide table
// from the output of the 8-bit-window kernel itself)
template <class C>
__global__ void __launch_bounds__(256) table_from_bytes_kernel(const u32* xy, AffEntry<C>* table, size_t total) {
  ECGPU_GRID_STRIDE(e, total) {
    C::fe_load(table[e].x, xy + e * 2 * C::NW);
    C::fe_load(table[e].y, xy + e * 2 * C::NW + C::NW);
  }
}
// scalars d * 2^(WB j) for j < nwin_wide, d = 1..2^(WB-1) (canonical big-endian bytes), reduced mod n
// (entries first .. first + count of the table: the wide tables are built in chunks so that the scratch stays small)
template <class C, int WB>
__global__ void __launch_bounds__(256) table_scalars_kernel(u32* out, size_t first, size_t count) {
  constexpr int NW = C::NW;
  ECGPU_GRID_STRIDE(idx, count) {
    const size_t e = first + idx;
    const int j = (int)(e / wide_entries<WB>());
    u32 d = (u32)(e % wide_entries<WB>()) + 1;
    // a carry window only ever sees digit 1 (k <= n/2 after the sign fold): its other entries are never read
    const bool carry_window = wide_carry_window<C, WB>() && (j == nwin_wide<C, WB>() - 1);
    if (carry_window && d > 1) d = 1;
    // likewise a top window with spare bits never sees a digit above 2^(bits it covers) (+ carry)
    constexpr int TOP_BITS = 8 * C::NB - WB * (nwin_wide<C, WB>() - 1);
    if (!wide_carry_window<C, WB>() && j == nwin_wide<C, WB>() - 1 && d > (1u << TOP_BITS)) d = 1;
    u32 k[NW + 1];
#pragma unroll
    for (int w = 0; w <= NW; w++) k[w] = 0;
    // d << (WB j): d <= 2^(WB-1) < 2^32 spans at most two 32-bit words
    const int wi = (WB * j) >> 5, sh = (WB * j) & 31;
    const u64 wide = (u64)d << sh;
#pragma unroll
    for (int w = 0; w <= NW; w++) {
      if (w == wi) k[w] |= (u32)wide;
      if (w == wi + 1) k[w] |= (u32)(wide >> 32);
    }
    // the top window can give a value of 32 NW + a few bits, which exceeds the word array: reduce 2^(32 NW) = R mod n by subtraction
    u32 ord[NW];
    C::order(ord);
    if (k[NW]) {
      // value = k[NW] * 2^(32 NW) + low;  2^(32 NW) mod n = 2^(32 NW) - n  (n > 2^(32 NW - 1))
      u32 negn[NW], acc[NW], bw = 0;
#pragma unroll
      for (int w = 0; w < NW; w++) negn[w] = subb(0u, ord[w], bw);      // 2^(32 NW) - n
      mp_zero<NW>(acc);
      for (u32 t = 0; t < k[NW]; t++) { mp_add<NW>(acc, acc, negn); reduce_once<NW>(acc, ord); }
      u32 lo[NW];
#pragma unroll
      for (int w = 0; w < NW; w++) lo[w] = k[w];
      reduce_once<NW>(lo, ord);
      const u32 cy = mp_add<NW>(acc, acc, lo);
      if (cy) { u32 t2[NW]; mp_sub<NW>(t2, acc, ord); mp_copy<NW>(acc, t2); }
      reduce_once<NW>(acc, ord);
#pragma unroll
      for (int w = 0; w < NW; w++) k[w] = acc[w];
    } else {
      reduce_once<NW>(k, ord);
    }
    words_store_be<NW>(out + idx * NW, k);
  }
}

// The accumulator is FbAcc<C>: XYZZ on the 256-bit curves (round 3; Jacobian before), Jacobian on P-384.
template <class C, int WB, int BATCH, int WAVES>
__global__ void __launch_bounds__(256, WAVES) mul_wide_kernel(const u32* scalars, const AffEntry<C>* table, u32* out, int out_fmt,
                                                           uint8_t* out_inf, size_t n_all, WaveSched sched) {
  constexpr int NW = C::NW;
  typename FbAcc<C>::Pt res[BATCH];
  typename C::Fe pre[BATCH];
  size_t idx[BATCH];                 // global index of every buffered result
  // Every wave draws small chunks of 64 x u consecutive scalars (sched.hpp), keeps the results in res[] across chunks and flushes them with one
  // shared inversion when the buffer is full or the work has run out.  A null counter = the static grid stride (the table builds use it).
  const bool dynamic = sched.counter != nullptr;
  const size_t T = dynamic ? (size_t)64 : (size_t)gridDim.x * blockDim.x;
  size_t base = dynamic ? 0 : (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = n_all;
  int cnt = 0, slots = 0;            // results buffered by this lane; per-lane units drawn since the last flush (wave-uniform)
  for (;;) {
    int units = BATCH;
    bool more = true;
    if (dynamic) {
      size_t lo;
      more = wave_next_chunk(sched, lo, n);
      base = lo + (threadIdx.x & 63u);
      units = more ? (int)((n - lo + 63) / 64) : 0;
    } else if (base >= n) {
      break;
    }
#pragma unroll 1
    for (int b = 0; b < units; b++) {
      const size_t i = base + (size_t)b * T;
      if (i >= n) break;
      u32 k[NW], ord[NW], t[NW];
      C::scalar_load(k, scalars + i * NW);
      C::order(ord);
      reduce_once<NW>(k, ord);
      mp_sub<NW>(t, ord, k);
      const bool flip = !mp_geq<NW>(t, k);
#pragma unroll
      for (int w = 0; w < NW; w++) k[w] = flip ? t[w] : k[w];
      typename FbAcc<C>::Pt acc;
      FbAcc<C>::set_infinity(acc);
      int filled = 0;                                     // 0: empty, 1: one entry (affine: Z = 1 / ZZ = ZZZ = 1), 2: a sum
      u32 carry = 0;
      // signed digit of window j (WB bits of k from bit WB j, words beyond the scalar read as zero; updates the recoding carry)
      auto digit = [&](int j) -> int {
        const int wi = (WB * j) >> 5, sh = (WB * j) & 31;
        u32 w0 = 0, w1 = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) { w0 = (wi == q) ? k[q] : w0; w1 = (wi + 1 == q) ? k[q] : w1; }
        const u64 pair = ((u64)w1 << 32) | w0;
        u32 d = ((u32)(pair >> sh) & ((1u << WB) - 1u)) + carry;
        carry = (d >= (1u << (WB - 1))) ? 1u : 0u;      // d in [2^(WB-1), 2^WB] becomes d - 2^WB with a carry
        return (int)d - (int)(carry << WB);
      };
      // Software pipeline over the gathers (round 4): the entry of window j + 1 is in flight during the addition of window j.  16 more
      // live registers (128 VGPRs, 7 spilled on P-256); p256 fixed base 2^24: 12.48-12.59 ms against 12.65-12.76 (-1.6 %, three alternating
      // passes, profiles/r04_ab_measurements.txt); at 3 waves per SIMD with or without it: 13.0-13.3 ms.  P-384's 24 extra registers cost
      // it 18 % (round 2): the plain loop stays there.  ECGPU_FB_NO_PREFETCH: A/B switch.
#ifdef ECGPU_FB_NO_PREFETCH
      constexpr bool PREFETCH = false;
#else
      constexpr bool PREFETCH = (NW <= 8);
#endif
      if constexpr (PREFETCH) {
        int nsd = digit(0);
        typename C::Fe nx, ny;
        C::fe_zero(nx); C::fe_zero(ny);
        if (nsd != 0) { const AffEntry<C>* e = table + ((nsd < 0 ? -nsd : nsd) - 1); nx = e->x; ny = e->y; }
#pragma unroll 1
        for (int j = 0; j < nwin_wide<C, WB>(); j++) {
          const int sd = nsd;
          typename C::Fe x = nx, y = ny;
          if (j + 1 < nwin_wide<C, WB>()) {
            nsd = digit(j + 1);
            if (nsd != 0) { const AffEntry<C>* e = table + (size_t)(j + 1) * wide_entries<WB>() + ((nsd < 0 ? -nsd : nsd) - 1); nx = e->x; ny = e->y; }
          }
          if (sd != 0) {
            if ((sd < 0) != flip) C::fe_neg(y, y);
            // the second entry meets an accumulator with ZZ = ZZZ = 1: 4M + 2S instead of 8M + 2S (one of the nine additions of a
            // 26-bit-window multiplication; the lanes of a wave disagree about `filled` only after a zero digit, 2^-WB per window)
            if (filled == 1) { FbAcc<C>::add_affine(acc, x, y); filled = 2; }
            else { FbAcc<C>::add_mixed(acc, x, y); filled = filled ? 2 : 1; }
          }
        }
      } else {
#pragma unroll 1
        for (int j = 0; j < nwin_wide<C, WB>(); j++) {
          const int sd = digit(j);
          if (sd != 0) {
            const AffEntry<C>* e = table + (size_t)j * wide_entries<WB>() + ((sd < 0 ? -sd : sd) - 1);
            typename C::Fe x = e->x, y = e->y;
            if ((sd < 0) != flip) C::fe_neg(y, y);
            if (filled == 1) { FbAcc<C>::add_affine(acc, x, y); filled = 2; }
            else { FbAcc<C>::add_mixed(acc, x, y); filled = filled ? 2 : 1; }
          }
        }
      }
      res[cnt] = acc;
      idx[cnt] = i;
      cnt++;
    }
    slots += units;
    if (!dynamic || !more || slots + (int)sched.chunk_units > BATCH) {
      if (cnt) FbAcc<C>::store(res, pre, cnt, 0, 0, out, out_fmt, out_inf, idx);
      cnt = 0;
      slots = 0;
    }
    if (dynamic && !more) break;
    if (!dynamic) base += T * BATCH;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Constant-time fixed base for SECRET scalars (signing nonces): the throughput schedule above indexes its tables by the
// digits and branches on them; the reference's own schedule is constant-time but, for the primeorder curves, is the
// generic variable-base multiplication (256 doublings: primeorder/src/projective.rs:422-431, "TODO: precomputed
// basepoint tables").  This kernel gives the same group element k G with no doubling and nothing that depends on k but
// data:
//   (the per-result body is mul_ct_one in fixedbase_ct.hpp, host + device, so that the host twin can trace its table reads)
//   * signed 5-bit digits by branch-free recoding; table T[j][d-1] = d 2^(5j) G, d = 1..16 (52 x 16 entries for a
//     256-bit curve: 53 KB, the wide-table builder with WB = 5);
//   * EVERY entry of window j is read (the address depends on j and the entry number only - one broadcast load for the
//     whole wave) and the digit's one is kept by masks, like LookupTable::select (k256 mul.rs:92-127);
//   * ONE Jacobian mixed addition per window, executed for every digit; a zero digit and an empty accumulator are resolved by
//     masks afterwards, and the bounds on the digits keep the formula off its exceptional cases (argument in
//     fixedbase_ct.hpp; round 2 had the reference's complete mixed addition here); the sign of the digit is a masked negation;
//   * one inversion per BATCH results (Montgomery's trick on the homogeneous Z; a zero Z is masked to 1 and flagged).
// 52 (77) additions of 8M + 3S instead of the reference schedule's 256 (384) doublings and 64 (96) additions.
// ---------------------------------------------------------------------------------------------------------------------
template <class C, int BATCH, int WAVES>
__global__ void __launch_bounds__(256, WAVES) mul_ct_kernel(const u32* scalars, const AffEntry<C>* table, u32* out, int out_fmt, uint8_t* out_inf, size_t n) {
  constexpr int NW = C::NW;
  static_assert(ct_nwin<C>() == nwin_wide<C, CT_WB>() && CT_ENTRIES == wide_entries<CT_WB>(), "the table is the wide-table builder's with WB = 5");
  using Fe = typename C::Fe;
  using Pt = typename C::Pt;
  Pt res[BATCH];
  Fe pre[BATCH];
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * BATCH) {
    int cnt = 0;
#pragma unroll 1
    for (int b = 0; b < BATCH; b++) {
      const size_t i = base + (size_t)b * T;
      if (i >= n) break;
      u32 k[NW], ord[NW];
      C::scalar_load(k, scalars + i * NW);
      C::order(ord);
      reduce_once<NW>(k, ord);
      Pt acc;
      mul_ct_one<C>(acc, k, table);
      res[b] = acc;
      cnt = b + 1;
    }
    // x = X / Z, y = Y / Z with one inversion for the cnt results of this lane
    Fe run, one, zero;
    C::fe_one(run); C::fe_one(one); C::fe_zero(zero);
    u32 infs = 0;
#pragma unroll 1
    for (int b = 0; b < cnt; b++) {
      const bool inf = C::fe_is_zero(res[b].z);
      infs |= (inf ? 1u : 0u) << b;
      C::fe_select(res[b].z, inf, one, res[b].z);
      pre[b] = run;
      C::fe_mul(run, run, res[b].z);
    }
    Fe inv;
    C::fe_inv(inv, run);
#pragma unroll 1
    for (int b = cnt - 1; b >= 0; b--) {
      const size_t i = base + (size_t)b * T;
      const bool inf = (infs >> b) & 1u;
      Fe zi, x, y;
      C::fe_mul(zi, inv, pre[b]);
      C::fe_mul(inv, inv, res[b].z);
      C::fe_mul(x, res[b].x, zi);
      C::fe_mul(y, res[b].y, zi);
      C::fe_select(x, inf, zero, x);
      if (out_fmt == FMT_PROJECTIVE) {
        C::fe_select(y, inf, one, y);
        C::fe_store(out + i * 3 * NW, x);
        C::fe_store(out + i * 3 * NW + NW, y);
        C::fe_store(out + i * 3 * NW + 2 * NW, inf ? zero : one);
      } else {
        C::fe_select(y, inf, zero, y);
        C::fe_store(out + i * 2 * NW, x);
        C::fe_store(out + i * 2 * NW + NW, y);
      }
      if (out_inf) out_inf[i] = inf ? 1 : 0;
    }
  }
}

}  // namespace fb
}  // namespace ecgpu
