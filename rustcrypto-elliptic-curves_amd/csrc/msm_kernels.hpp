// Kernels and host launcher of the Pippenger MSM (see msm.hpp for the schedule), written once over the curve traits and
// the window width.  Device code only.
#pragma once
#include <type_traits>
#include "ecgpu_internal.hpp"
#include "kernels.hpp"
#include "msm.hpp"

namespace ecgpu {
namespace msm {

// ---------------------------------------------------------------------------------------------------------------------
// Geometry of one window width CB (bits of a signed digit): buckets, the two levels of the sort, the sorted entry, the
// reduction tree.  Two widths are instantiated: 16 and 19 (see msm_run for which sizes take which).
// ---------------------------------------------------------------------------------------------------------------------
template <int CB>
struct Geo {
  static constexpr int CBITS = CB;
  static constexpr int NBUCKET = 1 << (CB - 1);              // |digit| in 1 .. 2^(CB-1)
  // sort: NCOARSE coarse bins of NFINE buckets per window
  static constexpr int LOG_FINE = (CB <= 16) ? 6 : 7, NFINE = 1 << LOG_FINE, NCOARSE = NBUCKET / NFINE;
  // sorted entry (32 bits) = term index | low bucket bits << INDEX_BITS | GLV half << 30 | subtract << 31
  static constexpr int INDEX_BITS = 30 - LOG_FINE;
  static constexpr u32 INDEX_MASK = (1u << INDEX_BITS) - 1u;
  static constexpr size_t SLAB_TERMS = (size_t)1 << INDEX_BITS;      // a call is cut into slabs of this many terms
  // reduction: a lane of the first level takes M buckets; 256 of its nodes make a group, NG groups a window
  static constexpr int LOG_M = 3, M = 1 << LOG_M;
  static constexpr int LOG_NG = 4, NG = 1 << LOG_NG;                 // 16 groups = 4096 nodes per window enter the group kernel
  static constexpr int NMID = (CB - 1 - LOG_M - 8 - LOG_NG) / LOG_M; // M-ary levels between the bucket level and the groups (0 or 1)
  static_assert((CB - 1 - LOG_M - 8 - LOG_NG) % LOG_M == 0 && NMID >= 0, "the levels must end at 4096 nodes per window");
  using Mag = typename std::conditional<(CB <= 16), uint16_t, u32>::type;   // storage of a digit magnitude (0 .. 2^(CB-1))
};
// per-curve shape of the digit matrix
template <class C, int CB>
struct Cfg {
  static constexpr int NHALF = C::A_IS_ZERO ? 2 : 1;                 // GLV halves per term (secp256k1 only)
  static constexpr int BITS = C::A_IS_ZERO ? 128 : 32 * C::NW;       // bits of a (half-)scalar magnitude
  static constexpr int NREAL = (BITS + CB - 1) / CB;                 // windows that cover them
  // a window width that divides BITS leaves no room for the carry of the signed recoding: one more window with a single
  // bucket takes it (CB = 16).  Otherwise the windows have room to spare, and the spare bits are taken from the LOW windows,
  // one each (NARROW windows of CB - 1 bits): leaving them all to the top window (14 of 19 bits for secp256k1) would put
  // that window's entries into 1/32 of its buckets and coarse bins - the sort's workgroup-per-bin level then waits for 128
  // bins of sixteen times the size.  With CB - 1 bits a window fills half its buckets, which the sort does not notice.
  static constexpr bool HAS_CARRY = (NREAL * CB == BITS);
  static constexpr int CAP = BITS + (C::A_IS_ZERO ? 1 : 0);          // bits the windows must hold: |k1|, |k2| < 2^128 plus the carry; k <= n/2 < 2^(BITS-1) plus the carry
  static constexpr int SPARE = HAS_CARRY ? 0 : NREAL * CB - CAP;
  static constexpr int NARROW = SPARE < NREAL ? SPARE : NREAL;
  static constexpr int NWIN = NREAL + (HAS_CARRY ? 1 : 0);
  static constexpr int NDIG = NWIN * NHALF;                          // digit columns per term
  static_assert(NDIG <= 32, "one sign bit per digit column");
  static_assert(HAS_CARRY || SPARE >= 0, "the windows must hold the magnitude and the recoding carry");
  __host__ __device__ static constexpr int width(int w) { return CB - (w < NARROW ? 1 : 0); }       // bits of window w
  __host__ __device__ static constexpr int pos(int w) { return w * CB - (w < NARROW ? w : NARROW); }  // its lowest bit
};

// ---------------------------------------------------------------------------------------------------------------------
// 0. The digits of every term, computed ONCE (the first version re-derived all digits of a term in each of the
//    histogram and scatter workgroups that looked at it).
//    secp256k1: the scalar is split by the endomorphism (k = k1 + k2 lambda, magnitudes below 2^128 and two signs); each
//    half gives NREAL signed CB-bit digits in [-2^(CB-1), 2^(CB-1)) and, for CB = 16, a carry digit in {0, 1}.
//    P-256 / P-384: one "half"; k > n/2 is replaced by n - k with the opposite sign.
//    mag[(NHALF w + h) * ns + i] = |digit| of window w of half h (window-major so that a workgroup streams its window's
//    digits, four per load; the row stride ns is n rounded up to a multiple of four),
//    sgn[i] bit NHALF w + h = the entry is subtracted (sign of the digit xor sign of the half).  Terms whose point is the
//    identity are not filtered here: the bucket sums skip them.
// ---------------------------------------------------------------------------------------------------------------------
template <int NWORDS>
__device__ __forceinline__ u32 window_bits(const u32* k, int bit, int width) {
  const int lo = bit >> 5, sh = bit & 31;
  u32 v = (lo < NWORDS) ? (k[lo] >> sh) : 0u;
  if (sh + width > 32 && lo + 1 < NWORDS) v |= k[lo + 1] << (32 - sh);
  return v & ((1u << width) - 1u);
}
template <class C, int CB, int NWORDS>
__device__ __forceinline__ void recode_half(const u32* m, u32 neg, int h, size_t i, size_t ns, typename Geo<CB>::Mag* mag, u32& bits) {
  using K = Cfg<C, CB>;
  using Mag = typename Geo<CB>::Mag;
  constexpr int NHALF = K::NHALF;
  u32 carry = 0;
#pragma unroll
  for (int w = 0; w < K::NREAL; w++) {
    const int wd = K::width(w);
    const u32 v = window_bits<NWORDS>(m, K::pos(w), wd) + carry;
    if (w < K::NREAL - 1 || K::HAS_CARRY) carry = (v >= (1u << (wd - 1))) ? 1u : 0u;       // v in [2^(wd-1), 2^wd] becomes v - 2^wd with a carry
    else carry = 0;                                                                        // top window: room for the last carry, v <= 2^(wd-1)
    const int d = (int)v - (int)(carry << wd);
    mag[(size_t)(NHALF * w + h) * ns + i] = (Mag)(d < 0 ? -d : d);
    bits |= (((d < 0) ? 1u : 0u) ^ neg) << (NHALF * w + h);
  }
  if (K::HAS_CARRY) {
    mag[(size_t)(NHALF * K::NREAL + h) * ns + i] = (Mag)carry;
    bits |= neg << (NHALF * K::NREAL + h);
  }
}
template <class C, int CB>
__global__ void __launch_bounds__(256) digits_kernel(const u32* scalars, size_t n, size_t ns, typename Geo<CB>::Mag* mag, u32* sgn) {
  constexpr int NW = C::NW;
  ECGPU_GRID_STRIDE(i, n) {
    u32 k[NW], ord[NW];
    C::scalar_load(k, scalars + i * NW);
    C::order(ord);
    reduce_once<NW>(k, ord);
    u32 bits = 0;
    if constexpr (Cfg<C, CB>::NHALF == 2) {
      k256::GlvSplit sp;
      k256::glv_split(sp, k);
      recode_half<C, CB, 4>(sp.k1, sp.neg1 ? 1u : 0u, 0, i, ns, mag, bits);
      recode_half<C, CB, 4>(sp.k2, sp.neg2 ? 1u : 0u, 1, i, ns, mag, bits);
    } else {
      u32 t[NW];
      mp_sub<NW>(t, ord, k);                   // n - k
      const bool flip = !mp_geq<NW>(t, k);     // n - k < k
#pragma unroll
      for (int w = 0; w < NW; w++) k[w] = flip ? t[w] : k[w];
      recode_half<C, CB, NW>(k, flip ? 1u : 0u, 0, i, ns, mag, bits);
    }
    sgn[i] = bits;
  }
}

// The points in the field's internal form, 2 NW words each (x limbs, y limbs; the identity - all-zero wire bytes - stays
// all zero): prep[h * n + i].  Half 1 (secp256k1) is lambda P = (beta x, y) (k256 projective.rs:287-293).  The bucket sums
// gather these instead of the wire format: no byte swap and, for the NIST curves, no conversion to Montgomery form per use.
template <class C>
__global__ void __launch_bounds__(256) prepare_points_kernel(const u32* xy, u32* prep, size_t n) {
  constexpr int NW = C::NW;
  ECGPU_GRID_STRIDE(i, n) {
    const uint4* src = (const uint4*)(xy + i * 2 * NW);          // 16-byte loads: a point is 8 NW contiguous bytes per lane
    u32 w[2 * NW];
#pragma unroll
    for (int q = 0; q < NW / 2; q++) { const uint4 v = src[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
    typename C::Fe x, y;
    C::fe_load(x, w);
    C::fe_load(y, w + NW);
    uint4* dst = (uint4*)(prep + i * 2 * NW);
#pragma unroll
    for (int q = 0; q < NW / 4; q++) dst[q] = make_uint4(x.v[4 * q], x.v[4 * q + 1], x.v[4 * q + 2], x.v[4 * q + 3]);
#pragma unroll
    for (int q = 0; q < NW / 4; q++) dst[NW / 4 + q] = make_uint4(y.v[4 * q], y.v[4 * q + 1], y.v[4 * q + 2], y.v[4 * q + 3]);
    if constexpr (C::A_IS_ZERO) {
      FeK256 b;
      k256::beta(b);
      k256::mul(x, x, b);
      uint4* d2 = (uint4*)(prep + (n + i) * 2 * NW);
#pragma unroll
      for (int q = 0; q < NW / 4; q++) d2[q] = make_uint4(x.v[4 * q], x.v[4 * q + 1], x.v[4 * q + 2], x.v[4 * q + 3]);
#pragma unroll
      for (int q = 0; q < NW / 4; q++) d2[NW / 4 + q] = make_uint4(y.v[4 * q], y.v[4 * q + 1], y.v[4 * q + 2], y.v[4 * q + 3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Two-level counting sort of the (half-term, window) entries by bucket, privatised in LDS.
//
// A direct scatter into the buckets of a window (the first version) writes every 4-byte entry to a different cache line,
// and a workgroup comes back to the same line only after it has touched tens of thousands of others: the lines leave
// the L2 partly written, HBM sees 142 M masked partial writes, and the scatter ran at 3.3 ms for 0.57 GB of output.
// Sorting in two levels keeps the set of lines a workgroup is filling small enough for the L2 to merge them:
//   level A  a workgroup owns one window and one contiguous chunk of the terms and splits its entries into NCOARSE
//            coarse bins of NFINE buckets;
//   level B  a workgroup owns one coarse bin (its entries are contiguous after level A) and sorts it by the low bucket
//            bits, which also yields the bucket offsets.
// Every count and every cursor increment is an LDS atomic.
// ---------------------------------------------------------------------------------------------------------------------
// chunk g of nch: boundaries are multiples of four terms (the loops below take four terms per step), the last chunk ends at n
__device__ __forceinline__ void chunk_range(size_t n, int g, int nch, size_t& lo, size_t& hi) {
  lo = (n * (size_t)g / nch) & ~(size_t)3;
  hi = (g == nch - 1) ? n : ((n * (size_t)(g + 1) / nch) & ~(size_t)3);
}
// the four digit magnitudes at terms i .. i + 3 of one row (i a multiple of four: one 8- or 16-byte load)
__device__ __forceinline__ void load_mag4(u32* a, const uint16_t* row, size_t i) {
  const uint2 v = *(const uint2*)(row + i);
  a[0] = v.x & 0xFFFFu; a[1] = v.x >> 16; a[2] = v.y & 0xFFFFu; a[3] = v.y >> 16;
}
__device__ __forceinline__ void load_mag4(u32* a, const u32* row, size_t i) {
  const uint4 v = *(const uint4*)(row + i);
  a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
}
// A1. part[w][g][cb] = number of entries of chunk g of window w in coarse bin cb
template <int CB>
__global__ void __launch_bounds__(1024) coarse_hist_kernel(const typename Geo<CB>::Mag* mag, size_t n, size_t ns, int nhalf, int nch, u32* part) {
  using G = Geo<CB>;
  __shared__ u32 cnt[G::NCOARSE];
  const int w = blockIdx.x / nch, g = blockIdx.x % nch;
  for (int b = threadIdx.x; b < G::NCOARSE; b += 1024) cnt[b] = 0;
  __syncthreads();
  size_t lo, hi;
  chunk_range(n, g, nch, lo, hi);
#pragma unroll 1
  for (int h = 0; h < nhalf; h++) {
    const typename G::Mag* src = mag + (size_t)(nhalf * w + h) * ns;
    for (size_t i = lo + 4 * (size_t)threadIdx.x; i < hi; i += 4096) {       // the row is padded to ns: reading past n within it is safe
      u32 a[4];
      load_mag4(a, src, i);
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (a[q] && i + q < hi) atomicAdd(&cnt[(a[q] - 1) >> G::LOG_FINE], 1u);
    }
  }
  __syncthreads();
  u32* dst = part + ((size_t)w * nch + g) * G::NCOARSE;
  for (int b = threadIdx.x; b < G::NCOARSE; b += 1024) dst[b] = cnt[b];
}
// A2. totals over the chunks, exclusive scan over all ncb = NWIN * NCOARSE coarse bins (one workgroup), cursors per chunk
static __global__ void __launch_bounds__(256) coarse_totals_kernel(const u32* part, int ncb, int ncoarse, int nch, u32* tot) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncb) return;
  const int w = j / ncoarse, cb = j % ncoarse;
  u32 s = 0;
#pragma unroll 1
  for (int g = 0; g < nch; g++) s += part[((size_t)w * nch + g) * ncoarse + cb];
  tot[j] = s;
}
static __global__ void __launch_bounds__(1024) coarse_scan_kernel(const u32* tot, int ncb, u32* coarse_off, u32* total_entries) {
  __shared__ u32 psum[1024];
  const int per = (ncb + 1023) / 1024;
  const int t = threadIdx.x;
  u32 s = 0;
  for (int q = 0; q < per; q++) { const int j = t * per + q; if (j < ncb) s += tot[j]; }
  psum[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    u32 v = (t >= off) ? psum[t - off] : 0;
    __syncthreads();
    psum[t] += v;
    __syncthreads();
  }
  u32 run = (t == 0) ? 0 : psum[t - 1];
  for (int q = 0; q < per; q++) { const int j = t * per + q; if (j < ncb) { coarse_off[j] = run; run += tot[j]; } }
  if (t == 1023) { coarse_off[ncb] = psum[1023]; *total_entries = psum[1023]; }    // one past the end: the number of sorted entries
}
static __global__ void __launch_bounds__(256) coarse_cursors_kernel(u32* part, int ncb, int ncoarse, int nch, const u32* coarse_off) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncb) return;
  const int w = j / ncoarse, cb = j % ncoarse;
  u32 run = coarse_off[j];
#pragma unroll 1
  for (int g = 0; g < nch; g++) {
    u32* p = part + ((size_t)w * nch + g) * ncoarse + cb;
    const u32 c = *p;
    *p = run;
    run += c;
  }
}
// Tile scatter, the write side of both sort levels.  A scattered 4-byte store is its own L2 transaction, and 10^8 of them
// per level were what the sort spent its time on (1.2 TB/s of useful traffic).  A workgroup therefore groups a TILE of
// entries by key in LDS first - count per key (the LDS atomic also hands out the entry's slot within its key), scan,
// place - and then copies the tile out in order: lane i stores element i, so the elements of one key go to consecutive
// addresses from consecutive lanes and a run of r entries costs r / 16 line writes instead of r.
//   PER entries per thread (registers: e[], the key and the slot packed in ks[], key 0xFFFF = no entry)
//   lds: buf[TILE] entries, key[TILE], off[NKEY + 1] tile offsets, cnt[NKEY] (zero on entry and on exit), wtot[NT / 64]
//   cur[NKEY]: where the next entry of each key goes in `dst` (advanced here)
template <int NKEY, int TILE, int NT>
struct TileLds {
  u32 buf[TILE];
  u32 off[NKEY + 1];
  u32 cnt[NKEY];
  u32 cur[NKEY];
  u32 wtot[NT / 64];
  uint16_t key[TILE];
};
//   gcur (optional): cursors shared with other workgroups (global memory); a tile then reserves its room per key with
//   one atomic instead of advancing L.cur
template <int NKEY, int TILE, int NT, int PER>
__device__ __forceinline__ void tile_scatter(TileLds<NKEY, TILE, NT>& L, const u32* e, u32* ks, u32* dst, u32* gcur = nullptr) {
  static_assert(PER * NT == TILE && NT % 64 == 0 && NKEY <= 0xFFFF, "tile shape");
  constexpr int KPT = (NKEY + NT - 1) / NT;                       // counters per thread in the scan
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1. count; the returned old value is the entry's slot within its key.  When every lane of the wave holds the same key
  //    (equal scalars: a whole window in one bucket) one lane counts for all - 64 atomics on one address would serialise.
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const u32 k = ks[j];
    const u32 k0 = __builtin_amdgcn_readfirstlane(k);
    if (__builtin_expect(__all(k == k0) && k0 != 0xFFFFu, 0)) {
      u32 base = 0;
      if (lane == 0) base = atomicAdd(&L.cnt[k0], 64u);
      ks[j] = k | ((__builtin_amdgcn_readfirstlane(base) + (u32)lane) << 16);
    } else if (k != 0xFFFFu) {
      ks[j] = k | (atomicAdd(&L.cnt[k], 1u) << 16);
    }
  }
  __syncthreads();
  // 2. exclusive scan of the counters: per thread, per wave (shuffles), across the waves (LDS)
  u32 v[KPT], local = 0;
#pragma unroll
  for (int q = 0; q < KPT; q++) { const int k = tid * KPT + q; v[q] = (k < NKEY) ? L.cnt[k] : 0u; local += v[q]; }
  u32 incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(incl, o); if (lane >= o) incl += y; }
  if (lane == 63) L.wtot[wave] = incl;
  __syncthreads();
  u32 run = incl - local;
#pragma unroll 1
  for (int w = 0; w < wave; w++) run += L.wtot[w];
#pragma unroll
  for (int q = 0; q < KPT; q++) {
    const int k = tid * KPT + q;
    if (k < NKEY) {
      L.off[k] = run;
      run += v[q];
      if (gcur && v[q]) L.cur[k] = atomicAdd(&gcur[k], v[q]);
    }
  }
  if (tid == NT - 1) L.off[NKEY] = run;
  __syncthreads();
  // 3. place
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const u32 k = ks[j] & 0xFFFFu;
    if (k != 0xFFFFu) { const u32 pos = L.off[k] + (ks[j] >> 16); L.buf[pos] = e[j]; L.key[pos] = (uint16_t)k; }
  }
  __syncthreads();
  // 4. copy out in tile order
  const u32 total = L.off[NKEY];
  for (u32 i = tid; i < total; i += NT) {
    const u32 k = L.key[i];
    dst[L.cur[k] + (i - L.off[k])] = L.buf[i];
  }
  __syncthreads();
  // 5. advance the cursors, clear the counters
#pragma unroll
  for (int q = 0; q < KPT; q++) { const int k = tid * KPT + q; if (k < NKEY) { if (!gcur) L.cur[k] += v[q]; L.cnt[k] = 0; } }
  __syncthreads();
}

// A3. entries into their coarse bins; same workgroup -> (window, chunk) map as the histogram.  A carry window (`carry_win`,
//     -1 if there is none) holds a single bucket, so its entries are final after this level and go straight to `sorted`.
//     (Slots are 16 bits: a tile holds at most 2^16 entries of one key.)
template <int CB>
struct CoarseTile {
  static constexpr int NT = 1024, PER = 16, TILE = NT * PER;
  using Lds = TileLds<Geo<CB>::NCOARSE, TILE, NT>;
};
template <int CB>
__global__ void __launch_bounds__(1024) coarse_scatter_kernel(const typename Geo<CB>::Mag* mag, const u32* sgn, size_t n, size_t ns, int nhalf, int carry_win, int nch,
                                                              const u32* part, u32* mid, u32* sorted) {
  using G = Geo<CB>;
  using T = CoarseTile<CB>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typename T::Lds& L = *reinterpret_cast<typename T::Lds*>(smem);
  const int w = blockIdx.x / nch, g = blockIdx.x % nch, tid = threadIdx.x;
  const u32* src = part + ((size_t)w * nch + g) * G::NCOARSE;
  for (int b = tid; b < G::NCOARSE; b += T::NT) { L.cur[b] = src[b]; L.cnt[b] = 0; }
  __syncthreads();
  size_t lo, hi;
  chunk_range(n, g, nch, lo, hi);
  u32* dst = (w == carry_win) ? sorted : mid;
#pragma unroll 1
  for (int h = 0; h < nhalf; h++) {
    const typename G::Mag* m = mag + (size_t)(nhalf * w + h) * ns;
#pragma unroll 1
    for (size_t t0 = lo; t0 < hi; t0 += T::TILE) {                 // a tile: TILE consecutive terms of this digit column
      u32 e[T::PER], ks[T::PER];
#pragma unroll
      for (int r = 0; r < T::PER / 4; r++) {
        const size_t i = t0 + 4 * ((size_t)r * T::NT + tid);
        u32 a[4] = {0, 0, 0, 0}, sg[4] = {0, 0, 0, 0};
        if (i < hi) {                                              // rows are padded to ns: reading past n within a row is safe
          load_mag4(a, m, i);
          const uint4 sv = *(const uint4*)(sgn + i);
          sg[0] = sv.x; sg[1] = sv.y; sg[2] = sv.z; sg[3] = sv.w;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const bool ok = a[q] != 0 && i + q < hi;
          const u32 b = a[q] - 1;
          ks[4 * r + q] = ok ? (b >> G::LOG_FINE) : 0xFFFFu;
          e[4 * r + q] = (u32)(i + q) | ((b & (G::NFINE - 1)) << G::INDEX_BITS) | ((u32)h << 30) | (((sg[q] >> (nhalf * w + h)) & 1u) << 31);
        }
      }
      tile_scatter<G::NCOARSE, T::TILE, T::NT, T::PER>(L, e, ks, dst);
    }
  }
}
// B. one workgroup per coarse bin: count its entries per bucket, scan the NFINE counts (which are the bucket offsets of the
//    whole sort: offsets[(w * NCOARSE + cb) * NFINE + f] is bucket w * NBUCKET + cb * NFINE + f), place the entries tile by tile.
template <int CB>
struct FineTile {
  static constexpr int NT = 256, PER = 16, TILE = NT * PER;
  using Lds = TileLds<Geo<CB>::NFINE, TILE, NT>;
};
// A bin above BIG_BIN_TILES tiles (equal or clustered scalars: whole windows in a few buckets) would keep one workgroup busy
// for milliseconds; it is only registered here and sorted by many workgroups, one tile each, with counters and cursors in
// global memory (big_count / big_scan / big_place below).
constexpr u32 BIG_BIN_TILES = 64;               // 2^18 entries; a uniform 2^24-term sum has 2^16 per bin
// the fine keys of a tile counted into L.cnt (PER loads in flight per thread, then their LDS atomics)
template <int CB, class L_t>
__device__ __forceinline__ void fine_count_tile(L_t& L, const u32* mid, u32 t0, u32 hi) {
  using G = Geo<CB>;
  using T = FineTile<CB>;
  const int t = threadIdx.x;
  u32 v[T::PER];
#pragma unroll
  for (int r = 0; r < T::PER; r++) { const u32 i = t0 + (u32)r * T::NT + t; v[r] = (i < hi) ? mid[i] : 0xFFFFFFFFu; }
#pragma unroll
  for (int r = 0; r < T::PER; r++) {
    const bool ok = t0 + (u32)r * T::NT + t < hi;
    const u32 k = (v[r] >> G::INDEX_BITS) & (G::NFINE - 1), k0 = __builtin_amdgcn_readfirstlane(k);
    if (__builtin_expect(__all(ok && k == k0), 0)) {         // one bucket for the whole wave (equal scalars): one atomic
      if ((t & 63) == 0) atomicAdd(&L.cnt[k0], 64u);
    } else if (ok) {
      atomicAdd(&L.cnt[k], 1u);
    }
  }
}
// the entries of a tile and their fine keys, consecutive lanes reading consecutive entries
template <int CB>
__device__ __forceinline__ void fine_load_tile(const u32* mid, u32 t0, u32 hi, u32* e, u32* ks) {
  using G = Geo<CB>;
  using T = FineTile<CB>;
#pragma unroll
  for (int r = 0; r < T::PER; r++) {
    const u32 i = t0 + (u32)r * T::NT + threadIdx.x;
    const bool ok = i < hi;
    e[r] = ok ? mid[i] : 0u;
    ks[r] = ok ? ((e[r] >> G::INDEX_BITS) & (G::NFINE - 1)) : 0xFFFFu;
  }
}
template <int CB>
__global__ void __launch_bounds__(256) fine_sort_kernel(const u32* mid, const u32* coarse_off, int carry_win, u32* offsets, u32* sorted, u32* big_ctr, u32* big_list) {
  using G = Geo<CB>;
  using T = FineTile<CB>;
  constexpr int NFINE = G::NFINE;
  __shared__ typename T::Lds L;
  __shared__ u32 scn[NFINE];
  const int j = (int)(gridDim.x - 1 - blockIdx.x), t = threadIdx.x;      // top windows first (measured: 0.39 ms against 0.45 ms in launch order)
  const u32 lo = coarse_off[j], hi = coarse_off[j + 1];
  if (carry_win >= 0 && j >= carry_win * G::NCOARSE) {     // carry window: every entry of the bin is in its first bucket, already in place
    if (t < NFINE) offsets[(size_t)j * NFINE + t] = (t == 0) ? lo : hi;
    return;
  }
  if (hi - lo > BIG_BIN_TILES * T::TILE) {
    if (t == 0) big_list[atomicAdd(big_ctr, 1u)] = (u32)j;
    return;
  }
  if (t < NFINE) L.cnt[t] = 0;
  __syncthreads();
#pragma unroll 1
  for (u32 t0 = lo; t0 < hi; t0 += T::TILE) fine_count_tile<CB>(L, mid, t0, hi);
  __syncthreads();
  // exclusive scan of the NFINE counts
  if (t < NFINE) scn[t] = L.cnt[t];
  __syncthreads();
#pragma unroll 1
  for (int off = 1; off < NFINE; off <<= 1) {
    const u32 v = (t < NFINE && t >= off) ? scn[t - off] : 0u;
    __syncthreads();
    if (t < NFINE) scn[t] += v;
    __syncthreads();
  }
  if (t < NFINE) {
    const u32 start = lo + scn[t] - L.cnt[t];
    L.cur[t] = start;
    L.cnt[t] = 0;
    offsets[(size_t)j * NFINE + t] = start;
  }
  __syncthreads();
#pragma unroll 1
  for (u32 t0 = lo; t0 < hi; t0 += T::TILE) {
    u32 e[T::PER], ks[T::PER];
    fine_load_tile<CB>(mid, t0, hi, e, ks);
    tile_scatter<NFINE, T::TILE, T::NT, T::PER>(L, e, ks, sorted);
  }
}
// big bins, pass 1: every workgroup counts tiles of every registered bin into big_cnt[b][f]
template <int CB>
__global__ void __launch_bounds__(256) big_count_kernel(const u32* mid, const u32* coarse_off, const u32* big_ctr, const u32* big_list, u32* big_cnt) {
  using G = Geo<CB>;
  using T = FineTile<CB>;
  constexpr int NFINE = G::NFINE;
  __shared__ struct { u32 cnt[NFINE]; } L;
  const int t = threadIdx.x;
  const u32 nbig = *big_ctr;
#pragma unroll 1
  for (u32 b = 0; b < nbig; b++) {
    const u32 j = big_list[b], lo = coarse_off[j], hi = coarse_off[j + 1];
    const u32 ntile = (hi - lo + T::TILE - 1) / T::TILE;
#pragma unroll 1
    for (u32 s = blockIdx.x; s < ntile; s += gridDim.x) {
      if (t < NFINE) L.cnt[t] = 0;
      __syncthreads();
      fine_count_tile<CB>(L, mid, lo + s * T::TILE, hi);
      __syncthreads();
      if (t < NFINE && L.cnt[t]) atomicAdd(&big_cnt[(size_t)b * NFINE + t], L.cnt[t]);
      __syncthreads();
    }
  }
}
// pass 2: one workgroup per registered bin: bucket offsets and the shared cursors
template <int CB>
__global__ void __launch_bounds__(256) big_scan_kernel(const u32* coarse_off, const u32* big_ctr, const u32* big_list, const u32* big_cnt, u32* big_cur, u32* offsets) {
  using G = Geo<CB>;
  constexpr int NFINE = G::NFINE;
  __shared__ u32 scn[NFINE];
  const int t = threadIdx.x;
  const u32 nbig = *big_ctr;
#pragma unroll 1
  for (u32 b = blockIdx.x; b < nbig; b += gridDim.x) {
    const u32 j = big_list[b], lo = coarse_off[j];
    const u32 c = (t < NFINE) ? big_cnt[(size_t)b * NFINE + t] : 0u;
    if (t < NFINE) scn[t] = c;
    __syncthreads();
#pragma unroll 1
    for (int off = 1; off < NFINE; off <<= 1) {
      const u32 v = (t < NFINE && t >= off) ? scn[t - off] : 0u;
      __syncthreads();
      if (t < NFINE) scn[t] += v;
      __syncthreads();
    }
    if (t < NFINE) {
      const u32 start = lo + scn[t] - c;
      big_cur[(size_t)b * NFINE + t] = start;
      offsets[(size_t)j * NFINE + t] = start;
    }
    __syncthreads();
  }
}
// pass 3: the tiles again, placed through the shared cursors
template <int CB>
__global__ void __launch_bounds__(256) big_place_kernel(const u32* mid, const u32* coarse_off, const u32* big_ctr, const u32* big_list, u32* big_cur, u32* sorted) {
  using G = Geo<CB>;
  using T = FineTile<CB>;
  constexpr int NFINE = G::NFINE;
  __shared__ typename T::Lds L;
  const int t = threadIdx.x;
  if (t < NFINE) L.cnt[t] = 0;
  __syncthreads();
  const u32 nbig = *big_ctr;
#pragma unroll 1
  for (u32 b = 0; b < nbig; b++) {
    const u32 j = big_list[b], lo = coarse_off[j], hi = coarse_off[j + 1];
    const u32 ntile = (hi - lo + T::TILE - 1) / T::TILE;
#pragma unroll 1
    for (u32 s = blockIdx.x; s < ntile; s += gridDim.x) {
      u32 e[T::PER], ks[T::PER];
      fine_load_tile<CB>(mid, lo + s * T::TILE, hi, e, ks);
      tile_scatter<NFINE, T::TILE, T::NT, T::PER>(L, e, ks, sorted, big_cur + (size_t)b * NFINE);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. Bucket sums over equal runs of the sorted entries.
//    The sorted array is cut into `ntask` runs of `len` consecutive entries (len = task_len(total, ntask)); lane t sums
//    run t in an XYZZ accumulator.  When the run leaves a bucket the accumulator is stored and cleared:
//      - a bucket that begins and ends inside the run is complete: it goes to bucketsX[b];
//      - a bucket that began before the run is the run's HEAD piece (head[t]); it may also extend past the run's end;
//      - a bucket that begins inside the run (or exactly at its start) and extends past its end is the TAIL piece
//        (tail[t]), and the lane appends the bucket to the list of buckets in pieces.
//    A bucket b in pieces that covers runs t0 < t1 is tail[t0] + head[t0 + 1] + .. + head[t1] (span_combine_kernel; a
//    bucket over more than SPAN_MAX runs - equal scalars put every term of a window into one bucket - is summed by
//    whole workgroups).  Every lane performs the same number of additions whatever the bucket sizes: no lane of a wave
//    waits for a longer bucket (one lane per bucket part, the previous version, lost 6 % at 512 entries per bucket and
//    would lose 30 % at 64), and over-full buckets need no separate accumulation path.
// ---------------------------------------------------------------------------------------------------------------------
constexpr u32 MSM_MIN_RUN = 16;                   // shortest run (small sums: fewer, longer runs than lanes)
constexpr u32 SPAN_MAX = 64, HEAVY_CHUNK = 4096;  // pieces one lane folds; pieces per workgroup of the heavy path
__host__ __device__ __forceinline__ u32 task_len(u32 total, u32 ntask) {
  const u32 len = (u32)(((u64)total + ntask - 1) / ntask);
  return len < MSM_MIN_RUN ? MSM_MIN_RUN : len;
}
struct HeavyBucket { u32 bucket, base, chunks; };
struct HeavyChunk { u32 bucket, index; };

// the prepared point an entry names: 2 NW words in the field's internal form (16-byte loads)
// A/B switches of the bucket sums (tools/ab_round3b.sh; DESIGN.md section 4 "MSM: round 3"):
//   MSM_BS_WAVES     occupancy target (waves per SIMD).  3 = 150 VGPRs and no spill instead of 128 + 110 spilled: 2.5 % SLOWER.
//   MSM_HOT_GATHER   DIAGNOSTIC ONLY (wrong sums): the term index of every gather is masked with this value, so the gathers
//                    come from a footprint of (mask + 1) x 64 bytes per half - what the gathers cost, by the cache level they hit.
#ifndef MSM_BS_WAVES
#define MSM_BS_WAVES 4
#endif
template <class C>
struct RawPoint { uint4 v[C::NW / 2]; };
template <class C, int CB>
__device__ __forceinline__ RawPoint<C> entry_point(const u32* prep, size_t n, u32 e) {
#ifdef MSM_HOT_GATHER
  const uint4* src = (const uint4*)(prep + ((size_t)((e >> 30) & 1u) * n + (e & (u32)MSM_HOT_GATHER)) * 2 * C::NW);
#elif defined(MSM_DIAG_ONE_ARRAY)
  // DIAGNOSTIC ONLY (wrong sums, round 4): both GLV halves gather from half 0's array - what halving the gathers' footprint
  // (1 GB -> 512 MB at 2^23 terms) would buy at most, before the sort is made half-aware to do it for real
  const uint4* src = (const uint4*)(prep + (size_t)(e & Geo<CB>::INDEX_MASK) * 2 * C::NW);
#else
  const uint4* src = (const uint4*)(prep + ((size_t)((e >> 30) & 1u) * n + (e & Geo<CB>::INDEX_MASK)) * 2 * C::NW);
#endif
  RawPoint<C> r;
#pragma unroll
  for (int q = 0; q < C::NW / 2; q++) r.v[q] = src[q];
  return r;
}
template <class C>
__device__ __forceinline__ void bucket_add_raw(Xyzz<C>& acc, const RawPoint<C>& r, u32 e) {
  constexpr int NW = C::NW;
  typename C::Fe x, y;
  u32 z = 0;
#pragma unroll
  for (int q = 0; q < NW / 4; q++) {
    x.v[4 * q] = r.v[q].x; x.v[4 * q + 1] = r.v[q].y; x.v[4 * q + 2] = r.v[q].z; x.v[4 * q + 3] = r.v[q].w;
    y.v[4 * q] = r.v[NW / 4 + q].x; y.v[4 * q + 1] = r.v[NW / 4 + q].y; y.v[4 * q + 2] = r.v[NW / 4 + q].z; y.v[4 * q + 3] = r.v[NW / 4 + q].w;
  }
  // the identity (all zero) contributes nothing; a non-identity point has x.v[NW-1] = y.v[NW-1] = 0 with probability 2^-64
  if (__builtin_expect((x.v[NW - 1] | y.v[NW - 1]) == 0, 0)) {
#pragma unroll
    for (int q = 0; q < NW; q++) z |= x.v[q] | y.v[q];
    if (z == 0) return;
  }
  if (e >> 31) C::fe_neg(y, y);
  xyzz_add_mixed<C>(acc, x, y);
}
template <class C>
__device__ __forceinline__ void store_xyzz(Xyzz<C>* dst, const Xyzz<C>& p) {
  constexpr int NW = C::NW;
  uint4* d = (uint4*)dst;
  const typename C::Fe* f[4] = {&p.x, &p.y, &p.zz, &p.zzz};
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int q = 0; q < NW / 4; q++) d[k * (NW / 4) + q] = make_uint4(f[k]->v[4 * q], f[k]->v[4 * q + 1], f[k]->v[4 * q + 2], f[k]->v[4 * q + 3]);
}
template <class C>
__device__ __forceinline__ void load_xyzz_as_jacobian(Jac<C>& r, const Xyzz<C>* src) {
  const Xyzz<C> p = *src;
  xyzz_to_jacobian<C>(r, p);
}

template <class C, int CB>
__global__ void __launch_bounds__(256, MSM_BS_WAVES) bucket_sum_kernel(const u32* prep, size_t n, const u32* offsets, const u32* sorted, int nb, u32 ntask, Xyzz<C>* bucketsX,
                                                            Xyzz<C>* head, Xyzz<C>* tail, u32* span_ctr, u32* span_list) {
  const u32 total = offsets[nb];
  const u32 len = task_len(total, ntask);
  ECGPU_GRID_STRIDE(tt, (size_t)ntask) {
    const u32 t = (u32)tt;
    const u64 s64 = (u64)t * len;
    if (s64 >= total) break;                      // runs are handed out in order: every later one of this lane is empty too
    const u32 s = (u32)s64, e = (total - s < len) ? total : s + len;
    // the bucket of entry s: offsets[b] <= s < offsets[b + 1]
    u32 lo = 0, hi = (u32)nb;
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) >> 1;
      if (offsets[mid] <= s) lo = mid; else hi = mid;
    }
    u32 b = lo, bend = offsets[b + 1];
    bool is_head = offsets[b] < s;                // the first bucket began in an earlier run
    Xyzz<C> acc;
    xyzz_set_infinity<C>(acc);
    u32 en = sorted[s];
    RawPoint<C> pn = entry_point<C, CB>(prep, n, en);
#pragma unroll 1
    for (u32 q = s; q < e; q++) {
      if (q == bend) {                            // bucket b ended at q: it is complete unless it is the head piece
        store_xyzz<C>(is_head ? &head[t] : &bucketsX[b], acc);
        xyzz_set_infinity<C>(acc);
        is_head = false;
        b++;
        bend = offsets[b + 1];
        if (bend <= q) {                          // empty buckets ahead (the unused half of a narrow window: 2^17 of them): bisect
          u32 l = b, h = (u32)nb - 1;             // offsets[l + 1] <= q < offsets[h + 1]
          while (h - l > 1) {
            const u32 mid = (l + h) >> 1;
            if (offsets[mid + 1] <= q) l = mid; else h = mid;
          }
          b = h;
          bend = offsets[b + 1];
        }
      }
      const u32 ec = en;
      const RawPoint<C> pc = pn;
      if (q + 1 < e) {                            // software pipeline: the gather of entry q + 1 is in flight during the addition of entry q
        en = sorted[q + 1];
        pn = entry_point<C, CB>(prep, n, en);
      }
      bucket_add_raw<C>(acc, pc, ec);
    }
    // the last bucket of the run
    if (is_head) {
      store_xyzz<C>(&head[t], acc);
    } else if (bend > e) {
      store_xyzz<C>(&tail[t], acc);
      span_list[atomicAdd(span_ctr, 1u)] = b;
    } else {
      store_xyzz<C>(&bucketsX[b], acc);
    }
  }
}
// piece p of a bucket that starts in run t0: p = 0 is tail[t0], p >= 1 is head[t0 + p]
template <class C>
__device__ __forceinline__ void load_piece(Jac<C>& r, const Xyzz<C>* head, const Xyzz<C>* tail, u32 t0, u32 p) {
  load_xyzz_as_jacobian<C>(r, p == 0 ? &tail[t0] : &head[t0 + p]);
}
// buckets in pieces: fold the pieces (few) or register the bucket for the workgroup path (many)
template <class C, int CB>
__global__ void __launch_bounds__(256) span_combine_kernel(const u32* offsets, int nb, u32 ntask, const u32* span_ctr, const u32* span_list, const Xyzz<C>* head,
                                                           const Xyzz<C>* tail, Xyzz<C>* bucketsX, u32* heavy_ctr, HeavyBucket* heavy, HeavyChunk* chunks) {
  const u32 len = task_len(offsets[nb], ntask);
  const u32 count = *span_ctr;
  ECGPU_GRID_STRIDE(i, (size_t)count) {
    const u32 b = span_list[i];
    const u32 t0 = offsets[b] / len, t1 = (offsets[b + 1] - 1) / len, np = t1 - t0 + 1;
    if (np > SPAN_MAX) {
      const u32 k = (np + HEAVY_CHUNK - 1) / HEAVY_CHUNK;
      const u32 idx = atomicAdd(&heavy_ctr[0], 1u);
      const u32 base = atomicAdd(&heavy_ctr[1], k);
      heavy[idx] = HeavyBucket{b, base, k};
#pragma unroll 1
      for (u32 q = 0; q < k; q++) chunks[base + q] = HeavyChunk{b, q};
      continue;
    }
    Xyzz<C> acc = tail[t0];                      // piece 0; pieces p >= 1 are head[t0 + p]
#pragma unroll 1
    for (u32 p = 1; p < np; p++) {
      const Xyzz<C> pc = head[t0 + p];
      xyzz_add<C>(acc, pc);
    }
    store_xyzz<C>(&bucketsX[b], acc);
  }
}

// sum over the 256 lanes of a workgroup through LDS
template <class C>
__device__ __forceinline__ void lds_tree_sum(Jac<C>* sh, Jac<C>& v, int lane, int count) {
  sh[lane] = v;
  __syncthreads();
  for (int off = count >> 1; off >= 1; off >>= 1) {
    if (lane < off) {
      Jac<C> a = sh[lane], b = sh[lane + off];
      pt_add<C>(a, a, b);
      sh[lane] = a;
    }
    __syncthreads();
  }
  v = sh[0];
  __syncthreads();
}
// one workgroup per chunk of HEAVY_CHUNK pieces of a heavy bucket: lanes stride over the pieces, LDS tree sum
template <class C, int CB>
__global__ void __launch_bounds__(256) heavy_chunk_kernel(const u32* offsets, int nb, u32 ntask, const u32* heavy_ctr, const HeavyChunk* chunks, const Xyzz<C>* head,
                                                          const Xyzz<C>* tail, Jac<C>* partial) {
  __shared__ Jac<C> sh[256];
  const u32 len = task_len(offsets[nb], ntask);
  const u32 total = heavy_ctr[1];
  for (u32 c = blockIdx.x; c < total; c += gridDim.x) {
    const HeavyChunk ch = chunks[c];
    const u32 t0 = offsets[ch.bucket] / len, t1 = (offsets[ch.bucket + 1] - 1) / len, np = t1 - t0 + 1;
    const u32 plo = ch.index * HEAVY_CHUNK, phi = (np - plo > HEAVY_CHUNK) ? plo + HEAVY_CHUNK : np;
    Jac<C> acc, pc;
    jac::set_infinity<C>(acc);
#pragma unroll 1
    for (u32 p = plo + threadIdx.x; p < phi; p += 256) {
      load_piece<C>(pc, head, tail, t0, p);
      pt_add<C>(acc, acc, pc);
    }
    lds_tree_sum<C>(sh, acc, threadIdx.x, 256);
    if (threadIdx.x == 0) partial[c] = acc;
  }
}
// one workgroup per heavy bucket: fold its chunk sums
template <class C>
__global__ void __launch_bounds__(256) heavy_finish_kernel(const u32* heavy_ctr, const HeavyBucket* heavy, const Jac<C>* partial, Xyzz<C>* bucketsX) {
  __shared__ Jac<C> sh[256];
  const u32 total = heavy_ctr[0];
  for (u32 h = blockIdx.x; h < total; h += gridDim.x) {
    const HeavyBucket hb = heavy[h];
    Jac<C> acc;
    jac::set_infinity<C>(acc);
#pragma unroll 1
    for (u32 j = threadIdx.x; j < hb.chunks; j += 256) pt_add<C>(acc, acc, partial[hb.base + j]);
    lds_tree_sum<C>(sh, acc, threadIdx.x, 256);
    if (threadIdx.x == 0) {
      Xyzz<C> r;
      jacobian_to_xyzz<C>(r, acc);
      store_xyzz<C>(&bucketsX[hb.bucket], r);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 4. Weighted sums S_w = sum_j (j + 1) B_j of a window's buckets, in three kernels with a short dependent chain.
//    a) level0: a lane takes M = 8 consecutive buckets: T = sum B_j, W = sum (j' + 1) B_j (j' = position inside the node)
//       by running sums from the top (run += B_j, W += run): 2 M additions.
//       With the nodes numbered i = 0 .. N-1 in a window:  S_w = sum_i W_i + M * sum_i i T_i.
//    (19-bit windows: one M-ary level of the same kind in between, W' = sum W_i + size sum i T_i, so that 4096 nodes per
//    window are left.)
//    b) group: a workgroup takes 256 consecutive nodes (i = 256 g + l) with three teams of 256 lanes working side by side:
//         A_g = sum_l W_i,   P_g = sum_l T_i,   Q_g = sum_l l T_i   (l T_i by double-and-add over 8 bits)
//       each by an LDS tree over its team, so that   sum_i i T_i = 256 sum_g g P_g + sum_g Q_g.
//    c) window: one workgroup per window, three teams again: sum_g A_g, sum_g g P_g (double-and-add over the bits of g),
//       sum_g Q_g, then S_w = A + M (256 P + Q) on one lane.
//    The dependent chain after level0 is about 700 field multiplications (the earlier tree of M-ary running-sum levels
//    finished by one workgroup had twice that, and it is latency, not throughput, that this stage costs).
// ---------------------------------------------------------------------------------------------------------------------
template <class C, int CB>
__global__ void __launch_bounds__(64) level0_kernel(const Xyzz<C>* bucketsX, const u32* offsets, Jac<C>* out_t, Jac<C>* out_w, int total) {
  constexpr int M = Geo<CB>::M;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  Xyzz<C> run, wt;                                 // the buckets are XYZZ: the running sums stay in that form (12M + 2S per addition)
  xyzz_set_infinity<C>(run);
  wt = run;
#pragma unroll 1
  for (int t = M - 1; t >= 0; t--) {
    const size_t b = (size_t)s * M + t;
    if (offsets[b + 1] != offsets[b]) {            // an empty bucket was never written
      const Xyzz<C> bj = bucketsX[b];
      xyzz_add<C>(run, bj);
    }
    xyzz_add<C>(wt, run);
  }
  Jac<C> r;
  xyzz_to_jacobian<C>(r, run);
  out_t[s] = r;
  xyzz_to_jacobian<C>(r, wt);
  out_w[s] = r;
}
// an M-ary level: children i = 0 .. M-1 of `size` = 2^log_size buckets each: T' = sum T_i, W' = sum W_i + size sum i T_i
// (running sums from the top: run += T_i, acc += run for i = M-1 .. 1)
template <class C, int CB>
__global__ void __launch_bounds__(64) level_kernel(const Jac<C>* in_t, const Jac<C>* in_w, Jac<C>* out_t, Jac<C>* out_w, int log_size, int total) {
  constexpr int M = Geo<CB>::M;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  const Jac<C>* T = in_t + (size_t)s * M;
  const Jac<C>* W = in_w + (size_t)s * M;
  Jac<C> run, acc, ws;
  jac::set_infinity<C>(run);
  acc = run;
  ws = W[0];
#pragma unroll 1
  for (int i = M - 1; i >= 1; i--) {
    pt_add<C>(run, run, T[i]);
    pt_add<C>(acc, acc, run);
    pt_add<C>(ws, ws, W[i]);
  }
  pt_add<C>(run, run, T[0]);
#pragma unroll 1
  for (int j = 0; j < log_size; j++) pt_dbl<C>(acc);
  pt_add<C>(ws, ws, acc);
  out_t[s] = run;
  out_w[s] = ws;
}
// sum over a team of `count` lanes (a power of two, a multiple of 128) through sh[count / 2]: the upper half of the lanes
// that are still active hands its values to the lower half
template <class C>
__device__ __forceinline__ void team_tree_sum(Jac<C>* sh, Jac<C>& v, int lane, int count) {
  for (int half = count >> 1; half >= 1; half >>= 1) {
    if (lane >= half && lane < 2 * half) sh[lane - half] = v;
    __syncthreads();
    if (lane < half) { const Jac<C> b = sh[lane]; pt_add<C>(v, v, b); }
    __syncthreads();
  }
}
// k * p by double-and-add over `bits` bits (k is lane-dependent: every lane executes every step)
template <class C>
__device__ __forceinline__ void small_multiple(Jac<C>& r, const Jac<C>& p, int k, int bits) {
  jac::set_infinity<C>(r);
#pragma unroll 1
  for (int bit = bits - 1; bit >= 0; bit--) {
    pt_dbl<C>(r);
    if ((k >> bit) & 1) pt_add<C>(r, r, p);
  }
}
constexpr int GROUP_NODES = 256, LOG_GROUP_NODES = 8;
template <class C>
__global__ void __launch_bounds__(3 * GROUP_NODES) group_kernel(const Jac<C>* in_t, const Jac<C>* in_w, Jac<C>* out) {
  __shared__ Jac<C> sh[3][GROUP_NODES / 2];
  const int team = threadIdx.x / GROUP_NODES, l = threadIdx.x % GROUP_NODES;      // teams are whole waves
  const size_t node = (size_t)blockIdx.x * GROUP_NODES + l;
  Jac<C> v;
  if (team == 0) v = in_w[node];
  else if (team == 1) v = in_t[node];
  else { const Jac<C> t = in_t[node]; small_multiple<C>(v, t, l, LOG_GROUP_NODES); }
  team_tree_sum<C>(sh[team], v, l, GROUP_NODES);
  if (l == 0) out[(size_t)blockIdx.x * 3 + team] = v;
}
// per window: `ng` groups (a power of two, at most 256; teams are padded to a whole number of waves)
template <class C>
__global__ void __launch_bounds__(3 * 256) window_kernel(const Jac<C>* grp, int ng, int log_ng, int team_lanes, int log_size, Jac<C>* win) {
  __shared__ Jac<C> sh[3][128];
  const int w = blockIdx.x, team = threadIdx.x / team_lanes, g = threadIdx.x % team_lanes;     // blockDim.x == 3 * team_lanes
  Jac<C> v;
  jac::set_infinity<C>(v);
  if (g < ng) {
    const Jac<C> x = grp[((size_t)w * ng + g) * 3 + team];
    if (team == 1) small_multiple<C>(v, x, g, log_ng);
    else v = x;
  }
  team_tree_sum<C>(sh[team], v, g, team_lanes);
  if (g == 0) sh[team][0] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    Jac<C> r = sh[1][0];                           // P
#pragma unroll 1
    for (int j = 0; j < LOG_GROUP_NODES; j++) pt_dbl<C>(r);
    pt_add<C>(r, r, sh[2][0]);                     // 256 P + Q
#pragma unroll 1
    for (int j = 0; j < log_size; j++) pt_dbl<C>(r);
    pt_add<C>(r, r, sh[0][0]);                     // A + size (256 P + Q)
    win[w] = r;
  }
}
// window sums of a further slab of terms are added to the running window sums
template <class C>
__global__ void __launch_bounds__(64) windows_accumulate_kernel(Jac<C>* total, const Jac<C>* slab, int nwin) {
  const int w = threadIdx.x;
  if (w < nwin) { Jac<C> a = total[w]; pt_add<C>(a, a, slab[w]); total[w] = a; }
}

// Small sums (n below SMALL_MSM_TERMS): the bucket method has a fixed cost of ~1.3 ms (the buckets to reduce, the serial
// doublings), more than n plain scalar multiplications take, so those run through the variable-base kernel and
// the n products are summed here: one workgroup per slice (lanes stride, LDS tree), then one workgroup over the slice
// sums.  Measured (ms, this path / 16-bit buckets) - k256: 2^12 1.06 / 1.54, 2^14 1.11 / 1.30, 2^16 1.21 / 1.31,
// 2^17 1.86 / 1.40; p256: 2^14 1.92 / 2.32, 2^16 2.08 / 2.56, 2^17 3.39 / 2.34.
constexpr size_t SMALL_MSM_TERMS = (size_t)5 << 14;
template <class C>
__global__ void __launch_bounds__(256) sum_affine_kernel(const u32* xy, size_t n, Jac<C>* partial) {
  __shared__ Jac<C> sh[256];
  constexpr int NW = C::NW;
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = (lo + per < n) ? lo + per : n;
  Xyzz<C> xacc;
  xyzz_set_infinity<C>(xacc);
#pragma unroll 1
  for (size_t j = lo + threadIdx.x; j < hi; j += 256) {         // wire-format affine points (the output of the scalar multiplications)
    const u32* src = xy + j * 2 * NW;
    u32 z = 0;
#pragma unroll
    for (int q = 0; q < 2 * NW; q++) z |= src[q];
    if (z == 0) continue;
    typename C::Fe x, y;
    C::fe_load(x, src);
    C::fe_load(y, src + NW);
    xyzz_add_mixed<C>(xacc, x, y);
  }
  Jac<C> acc;
  xyzz_to_jacobian<C>(acc, xacc);
  lds_tree_sum<C>(sh, acc, threadIdx.x, 256);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
template <class C>
__global__ void __launch_bounds__(256) sum_partials_kernel(const Jac<C>* partial, int count, Jac<C>* win) {
  __shared__ Jac<C> sh[256];
  Jac<C> acc;
  jac::set_infinity<C>(acc);
#pragma unroll 1
  for (int j = threadIdx.x; j < count; j += 256) pt_add<C>(acc, acc, partial[j]);
  lds_tree_sum<C>(sh, acc, threadIdx.x, 256);
  if (threadIdx.x == 0) win[0] = acc;                  // finish_kernel with nwin = 1 converts and stores it
}

// One doubling of a secp256k1 point shared by the lanes of a wave (every lane holds the same p and leaves with the same
// 2p).  The formula of k256::jac_double has seven multiplications in three dependent steps - {X^2, Y^2, Y Z}, then
// {X B, B^2, (3A)^2}, then E (D - X3) - so three lanes take one product each per step and swap results by shuffles: a lone
// lane's doubling is seven multiplication latencies, this one three (the final Horner pass is 110 doublings of one point).
__device__ __forceinline__ void fe_from_lane(FeK256& r, const FeK256& a, int src) {
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl(a.v[i], src);
}
__device__ __forceinline__ void fe_pick(FeK256& r, int role, const FeK256& a0, const FeK256& a1, const FeK256& a2) {
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = role == 0 ? a0.v[i] : (role == 1 ? a1.v[i] : a2.v[i]);
}
__device__ __forceinline__ void wave_double_k256(Jac<CurveK256>& p, int role) {
  FeK256 o1, o2, m, a, b, yz, e, t, xb, c, f, d;
  fe_pick(o1, role, p.x, p.y, p.y);
  fe_pick(o2, role, p.x, p.y, p.z);
  k256::mul(m, o1, o2);
  fe_from_lane(a, m, 0); fe_from_lane(b, m, 1); fe_from_lane(yz, m, 2);      // A = X^2, B = Y^2, Y Z
  k256::shl<1>(t, a); k256::add(e, t, a);                                    // E = 3A
  fe_pick(o1, role, p.x, b, e);
  fe_pick(o2, role, b, b, e);
  k256::mul(m, o1, o2);
  fe_from_lane(xb, m, 0); fe_from_lane(c, m, 1); fe_from_lane(f, m, 2);      // X B, C = B^2, E^2
  k256::shl<2>(d, xb);                                                       // D = 4 X B
  k256::sub(t, f, d); k256::sub(p.x, t, d);                                  // X3 = E^2 - 2D
  k256::sub(t, d, p.x); k256::mul(t, e, t);                                  // E (D - X3): every lane
  k256::shl<3>(c, c); k256::sub(p.y, t, c);                                  // Y3 = E (D - X3) - 8C
  k256::shl<1>(p.z, yz);                                                     // Z3 = 2 Y Z
}

// 5. Horner over the windows (window w is `cbits` bits wide, the lowest `narrow` windows one bit less), conversion to
//    affine, output.  One wave; secp256k1 shares the doublings among its lanes (above), lane 0 stores.
template <class C>
__global__ void __launch_bounds__(64) finish_kernel(const Jac<C>* win, int nwin, int cbits, int narrow, u32* out, int out_fmt) {
  if (blockIdx.x != 0) return;
  constexpr int NW = C::NW;
  using Fe = typename C::Fe;
  Jac<C> r = win[nwin - 1];
  if constexpr (C::A_IS_ZERO) {
    const int role = threadIdx.x < 2 ? (int)threadIdx.x : 2;
#pragma unroll 1
    for (int w = nwin - 2; w >= 0; w--) {
#pragma unroll 1
      for (int j = (w < narrow) ? 1 : 0; j < cbits; j++) wave_double_k256(r, role);      // times 2^(width of window w)
      pt_add<C>(r, r, win[w]);
    }
    if (threadIdx.x != 0) return;
  } else {
    if (threadIdx.x != 0) return;
#pragma unroll 1
    for (int w = nwin - 2; w >= 0; w--) {
#pragma unroll 1
      for (int j = (w < narrow) ? 1 : 0; j < cbits; j++) pt_dbl<C>(r);
      pt_add<C>(r, r, win[w]);
    }
  }
  const bool inf = C::fe_is_zero(r.z);
  Fe zi, zi2, zi3, x, y, one, zero;
  C::fe_one(one); C::fe_zero(zero);
  C::fe_inv(zi, r.z);
  C::fe_sqr(zi2, zi); C::fe_mul(zi3, zi2, zi);
  C::fe_mul(x, r.x, zi2); C::fe_mul(y, r.y, zi3);
  if (inf) { x = zero; y = (out_fmt == FMT_PROJECTIVE) ? one : zero; }
  C::fe_store(out, x);
  C::fe_store(out + NW, y);
  if (out_fmt == FMT_PROJECTIVE) C::fe_store(out + 2 * NW, inf ? zero : one);
}

// homogeneous projective input -> affine (one inversion per lane; only used when the caller hands X:Y:Z)
template <class C>
__global__ void __launch_bounds__(256) to_affine_kernel(const u32* xyz, u32* xy, size_t n) {
  ECGPU_GRID_STRIDE(i, n) {
    typename C::Pt p;
    load_point<C>(p, xyz + i * 3 * C::NW, FMT_PROJECTIVE);
    store_affine_from_projective<C>(xy + i * 2 * C::NW, nullptr, p);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side: all stages on c->stream out of one grow-only workspace.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef MSM_ROUNDS
#define MSM_ROUNDS 6                 // bucket-sum runs per resident lane (ECGPU_MSM_ROUNDS)
#endif
static inline size_t msm_align(size_t x) { return (x + 255) & ~(size_t)255; }
static int msm_reserve(ecgpu_ctx* c, size_t need) {
  if (need > c->msm_ws_cap) {
    if (c->msm_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->msm_ws)); c->msm_ws = nullptr; c->msm_ws_cap = 0; }
    HIPCHK(c, hipMalloc(&c->msm_ws, need));
    c->msm_ws_cap = need;
  }
  return 0;
}

// the bucket method with CB-bit windows
template <class C, int CB>
static int msm_buckets(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t n, u32* out, int out_fmt) {
  using G = Geo<CB>;
  using K = Cfg<C, CB>;
  using J = Jac<C>;
  using X = Xyzz<C>;
  using Mag = typename G::Mag;
  constexpr int NW = C::NW, NHALF = K::NHALF, NWIN = K::NWIN, NDIG = K::NDIG;
  constexpr int NCB = NWIN * G::NCOARSE;
  constexpr int CARRY_WIN = K::HAS_CARRY ? NWIN - 1 : -1;
  static_assert((size_t)NDIG * G::SLAB_TERMS < ((size_t)1 << 32), "32-bit offsets within a slab");
  auto al = msm_align;
  const size_t nb = (size_t)NWIN * G::NBUCKET;
  // Large sums run in slabs of at most SLAB_TERMS terms (a sorted entry keeps the term index in INDEX_BITS bits); every
  // slab goes through the whole pipeline down to its NWIN window sums, which are added up before the final Horner pass.
  // ECGPU_OPT_MSM_SLAB_TERMS overrides the slab size (tests exercise the slab loop on small inputs).
  size_t slab = c->opt[ECGPU_OPT_MSM_SLAB_TERMS] ? (size_t)c->opt[ECGPU_OPT_MSM_SLAB_TERMS] : G::SLAB_TERMS;
  if (slab < 1024 || slab > G::SLAB_TERMS) slab = G::SLAB_TERMS;
  const size_t m = n < slab ? n : slab;                // terms of the largest slab: sizes the workspace
  // bucket-sum runs: `rounds` per lane the chip holds at that kernel's occupancy (4 workgroups of 256 per CU)
  const int rounds = (c->opt[ECGPU_OPT_MSM_ROUNDS] >= 1 && c->opt[ECGPU_OPT_MSM_ROUNDS] <= 64) ? (int)c->opt[ECGPU_OPT_MSM_ROUNDS] : MSM_ROUNDS;
  const u32 ntask = (u32)rounds * (u32)c->num_cus * 256u * MSM_BS_WAVES;
  const size_t sz_aff = (pt_fmt == FMT_PROJECTIVE) ? al(m * 8 * NW) : 0, sz_prep = al((size_t)NHALF * m * 8 * NW);
  const size_t sz_off = al((nb + 1) * 4), sz_coff = al((size_t)(NCB + 1) * 4), sz_tot = al((size_t)NCB * 4), sz_sorted = al((size_t)NDIG * m * 4 + 32);
  // level A of the sort: one 1024-thread workgroup per CU, the chunks of a window side by side
  const int nch = (c->num_cus - 1) / NWIN > 0 ? (c->num_cus - 1) / NWIN : 1;
  const size_t sz_part = al((size_t)NWIN * nch * G::NCOARSE * 4);
  const size_t ms = (m + 3) & ~(size_t)3;              // row stride of the digit arrays: four terms per load
  const size_t sz_mag = al((size_t)NDIG * ms * sizeof(Mag)), sz_sgn = al(ms * 4);
  const size_t sz_bx = al(nb * sizeof(X)), sz_piece = al((size_t)ntask * sizeof(X)), sz_span = al((size_t)ntask * 4 + 8);
  const size_t n0 = nb / G::M;                         // nodes of the level that reads the buckets
  const size_t sz_l0 = al(n0 * sizeof(J)), sz_l1 = al(n0 / G::M * sizeof(J)), sz_grp = al(n0 / GROUP_NODES * 3 * sizeof(J)), sz_win = al(NWIN * sizeof(J));
  constexpr int NG = G::NG, LOG_NG = G::LOG_NG, TEAM = NG < 64 ? 64 : NG;        // lanes per team of the window kernel: whole waves
  // a run leaves at most two pieces (head and tail), so there are at most 2 ntask pieces: at most 2 ntask / SPAN_MAX buckets
  // over more than SPAN_MAX runs, cut into at most 2 ntask / HEAVY_CHUNK + one chunk each
  const size_t hmax = 2 * (size_t)ntask / SPAN_MAX + 1, cmax = 2 * (size_t)ntask / HEAVY_CHUNK + hmax + 1;
  // sort bins above BIG_BIN_TILES tiles: at most (entries of a slab) / (entries of such a bin) of them
  const size_t maxbig = (size_t)NDIG * m / (BIG_BIN_TILES * FineTile<CB>::TILE) + 1;
  const size_t sz_big = al(maxbig * 4) + 2 * al(maxbig * G::NFINE * 4);
  const size_t sz_ctr = al(16), sz_heavy = al(hmax * sizeof(HeavyBucket)), sz_chunks = al(cmax * sizeof(HeavyChunk)), sz_partial = al(cmax * sizeof(J));
  const size_t need = sz_aff + sz_prep + sz_mag + sz_sgn + sz_off + sz_coff + sz_tot + sz_part + 2 * sz_sorted + sz_bx + 2 * sz_piece + sz_span + 2 * sz_l0 + 2 * sz_l1 + sz_grp +
                      2 * sz_win + sz_ctr + sz_heavy + sz_chunks + sz_partial + sz_big;
  int rc = msm_reserve(c, need);
  if (rc) return rc;
  // the coarse scatter groups its tiles in more LDS than the 64 KB a kernel gets by default (set per call: the attribute
  // belongs to the function on the current device, and a process may hold contexts on several devices)
  HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&coarse_scatter_kernel<CB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sizeof(typename CoarseTile<CB>::Lds)));
  char* p = (char*)c->msm_ws;
  u32* aff = (u32*)p; p += sz_aff;
  u32* prep = (u32*)p; p += sz_prep;
  Mag* mag = (Mag*)p; p += sz_mag;
  u32* sgn = (u32*)p; p += sz_sgn;
  u32* offsets = (u32*)p; p += sz_off;
  u32* coarse_off = (u32*)p; p += sz_coff;
  u32* tot = (u32*)p; p += sz_tot;
  u32* part = (u32*)p; p += sz_part;
  u32* mid = (u32*)p; p += sz_sorted;
  u32* sorted = (u32*)p; p += sz_sorted;
  X* bucketsX = (X*)p; p += sz_bx;
  X* head = (X*)p; p += sz_piece;
  X* tail = (X*)p; p += sz_piece;
  u32* span_list = (u32*)p; p += sz_span;
  J* ta = (J*)p; p += sz_l0;
  J* wa = (J*)p; p += sz_l0;
  J* tb = (J*)p; p += sz_l1;
  J* wb = (J*)p; p += sz_l1;
  J* grp = (J*)p; p += sz_grp;
  J* win = (J*)p; p += sz_win;
  J* win_slab = (J*)p; p += sz_win;
  u32* ctr = (u32*)p; p += sz_ctr;                     // [0] buckets in pieces, [1] big sort bins, [2] heavy buckets, [3] heavy chunks
  HeavyBucket* heavy = (HeavyBucket*)p; p += sz_heavy;
  HeavyChunk* chunks = (HeavyChunk*)p; p += sz_chunks;
  J* partial = (J*)p; p += sz_partial;
  u32* big_cnt = (u32*)p; p += al(maxbig * G::NFINE * 4);          // zeroed together with the list behind it: see the memset below
  u32* big_list = (u32*)p; p += al(maxbig * 4);
  u32* big_cur = (u32*)p;
  const size_t pin = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * (size_t)NW;      // 32-bit words per input point
  const unsigned cb_grid = (unsigned)((NCB + 255) / 256);
  for (size_t s0 = 0; s0 < n; s0 += slab) {
    const size_t cnt = (n - s0 < slab) ? n - s0 : slab;
    const u32* ssc = sc + s0 * NW;
    const u32* xy = pts + s0 * pin;
    if (pt_fmt == FMT_PROJECTIVE) {
      hipLaunchKernelGGL((to_affine_kernel<C>), dim3(ecgpu_grid_for(c, cnt, 8)), dim3(256), 0, c->stream, xy, aff, cnt);
      xy = aff;
    }
    J* wdst = (s0 == 0) ? win : win_slab;
    hipLaunchKernelGGL((digits_kernel<C, CB>), dim3(ecgpu_grid_for(c, cnt, 8)), dim3(256), 0, c->stream, ssc, cnt, ms, mag, sgn);
    hipLaunchKernelGGL((prepare_points_kernel<C>), dim3(ecgpu_grid_for(c, cnt, 8)), dim3(256), 0, c->stream, xy, prep, cnt);
    hipLaunchKernelGGL((coarse_hist_kernel<CB>), dim3((unsigned)(NWIN * nch)), dim3(1024), 0, c->stream, (const Mag*)mag, cnt, ms, NHALF, nch, part);
    hipLaunchKernelGGL(coarse_totals_kernel, dim3(cb_grid), dim3(256), 0, c->stream, (const u32*)part, NCB, G::NCOARSE, nch, tot);
    hipLaunchKernelGGL(coarse_scan_kernel, dim3(1), dim3(1024), 0, c->stream, (const u32*)tot, NCB, coarse_off, offsets + nb);
    hipLaunchKernelGGL(coarse_cursors_kernel, dim3(cb_grid), dim3(256), 0, c->stream, part, NCB, G::NCOARSE, nch, (const u32*)coarse_off);
    hipLaunchKernelGGL((coarse_scatter_kernel<CB>), dim3((unsigned)(NWIN * nch)), dim3(1024), sizeof(typename CoarseTile<CB>::Lds), c->stream, (const Mag*)mag, (const u32*)sgn,
                       cnt, ms, NHALF, CARRY_WIN, nch, (const u32*)part, mid, sorted);
    HIPCHK(c, hipMemsetAsync(ctr, 0, 16, c->stream));
    HIPCHK(c, hipMemsetAsync(big_cnt, 0, al(maxbig * G::NFINE * 4), c->stream));
    hipLaunchKernelGGL((fine_sort_kernel<CB>), dim3((unsigned)NCB), dim3(256), 0, c->stream, (const u32*)mid, (const u32*)coarse_off, CARRY_WIN, offsets, sorted, ctr + 1,
                       big_list);
    hipLaunchKernelGGL((big_count_kernel<CB>), dim3((unsigned)c->num_cus * 4), dim3(256), 0, c->stream, (const u32*)mid, (const u32*)coarse_off, (const u32*)(ctr + 1),
                       (const u32*)big_list, big_cnt);
    hipLaunchKernelGGL((big_scan_kernel<CB>), dim3(64), dim3(256), 0, c->stream, (const u32*)coarse_off, (const u32*)(ctr + 1), (const u32*)big_list, (const u32*)big_cnt,
                       big_cur, offsets);
    hipLaunchKernelGGL((big_place_kernel<CB>), dim3((unsigned)c->num_cus * 4), dim3(256), 0, c->stream, (const u32*)mid, (const u32*)coarse_off, (const u32*)(ctr + 1),
                       (const u32*)big_list, big_cur, sorted);
    hipLaunchKernelGGL((bucket_sum_kernel<C, CB>), dim3(ntask / 256), dim3(256), 0, c->stream, (const u32*)prep, cnt, (const u32*)offsets, (const u32*)sorted, (int)nb, ntask,
                       bucketsX, head, tail, ctr, span_list);
    hipLaunchKernelGGL((span_combine_kernel<C, CB>), dim3(ecgpu_grid_for(c, ntask, 8)), dim3(256), 0, c->stream, (const u32*)offsets, (int)nb, ntask, (const u32*)ctr,
                       (const u32*)span_list, (const X*)head, (const X*)tail, bucketsX, ctr + 2, heavy, chunks);
    hipLaunchKernelGGL((heavy_chunk_kernel<C, CB>), dim3((unsigned)c->num_cus * 4), dim3(256), 0, c->stream, (const u32*)offsets, (int)nb, ntask, (const u32*)(ctr + 2),
                       (const HeavyChunk*)chunks, (const X*)head, (const X*)tail, partial);
    hipLaunchKernelGGL((heavy_finish_kernel<C>), dim3((unsigned)c->num_cus), dim3(256), 0, c->stream, (const u32*)(ctr + 2), (const HeavyBucket*)heavy, (const J*)partial,
                       bucketsX);
    // buckets -> nb / 8 nodes (-> / 8 for the wide windows) -> groups of 256 nodes -> window sums
    size_t nodes = n0;
    int log_size = G::LOG_M;
    J *it = ta, *iw = wa;
    hipLaunchKernelGGL((level0_kernel<C, CB>), dim3((unsigned)((nodes + 63) / 64)), dim3(64), 0, c->stream, (const X*)bucketsX, (const u32*)offsets, ta, wa, (int)nodes);
    for (int lv = 0; lv < G::NMID; lv++) {
      nodes /= G::M;
      hipLaunchKernelGGL((level_kernel<C, CB>), dim3((unsigned)((nodes + 63) / 64)), dim3(64), 0, c->stream, (const J*)it, (const J*)iw, tb, wb, log_size, (int)nodes);
      it = tb; iw = wb;
      log_size += G::LOG_M;
    }
    hipLaunchKernelGGL((group_kernel<C>), dim3((unsigned)(nodes / GROUP_NODES)), dim3(3 * GROUP_NODES), 0, c->stream, (const J*)it, (const J*)iw, grp);
    hipLaunchKernelGGL((window_kernel<C>), dim3(NWIN), dim3(3 * TEAM), 0, c->stream, (const J*)grp, NG, LOG_NG, TEAM, log_size, wdst);
    if (s0 != 0) hipLaunchKernelGGL((windows_accumulate_kernel<C>), dim3(1), dim3(64), 0, c->stream, win, (const J*)win_slab, NWIN);
  }
  hipLaunchKernelGGL((finish_kernel<C>), dim3(1), dim3(64), 0, c->stream, (const J*)win, (int)NWIN, CB, K::NARROW, out, out_fmt);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// `mul` is the curve's batch scalar multiplication (used for small sums).
// ECGPU_OPT_MSM_SMALL_PATH = 0 forces the bucket method for every size, ECGPU_OPT_MSM_WINDOW_BITS = 16 | 19 one window width
// (per-context options, include/ecgpu.h: measurements, tests of the bucket paths on small inputs).
// 19-bit windows pay between these sizes (tools/msm_sizes.sh; ms at 16 / 19 bits - k256: 2^20 2.72 / 2.84, 2^21 3.96 / 3.90,
// 2^22 6.34 / 6.17, 2^23 10.96 / 10.2, 2^24 20.06 / 20.17; p256: 2^21 4.98 / 5.51, 2^22 7.94 / 7.73, 2^23 12.93 / 12.33,
// 2^24 22.96 / 23.48; p384: 2^21 18.1 / 19.1, 2^22 28.8 / 28.4): below, the eight times larger bucket tree costs more than
// the bucket additions saved; from 2^24 terms on the 19-bit path needs two slabs (23-bit term index in a sorted entry)
// where the 16-bit path needs one.
constexpr size_t WIDE_WINDOW_MAX = (size_t)1 << 24;
template <class C>
constexpr size_t wide_window_min() { return C::A_IS_ZERO ? (size_t)1 << 21 : (size_t)1 << 22; }
template <class C, class MulFn>
static int msm_run(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t n, u32* out, int out_fmt, MulFn mul) {
  constexpr int NW = C::NW;
  using J = Jac<C>;
  if ((uintptr_t)sc & 3) return ecgpu_set_err(c, ECGPU_ERR_ARG, "ecgpu_msm: the scalars must be 4-byte aligned");
  if ((uintptr_t)pts & 3) return ecgpu_set_err(c, ECGPU_ERR_ARG, "ecgpu_msm: the points must be 4-byte aligned");
  const bool small_path = c->opt[ECGPU_OPT_MSM_SMALL_PATH] != 0;
  const bool small = small_path && n > 0 && n < SMALL_MSM_TERMS;
  // the bucket path reads AFFINE input points with 16-byte loads (projective input is normalised into the workspace first)
  if (!small && pt_fmt == FMT_AFFINE && ((uintptr_t)pts & 15))
    return ecgpu_set_err(c, ECGPU_ERR_ARG, "ecgpu_msm: affine device points must be 16-byte aligned for sums of %zu terms and more", (size_t)SMALL_MSM_TERMS);
  if (small) {
    // n scalar multiplications on the throughput kernel, then a two-level sum of the products
    auto al = msm_align;
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    const size_t sz_prod = al(n * 8 * NW), sz_part = al((size_t)blocks * sizeof(J)), sz_win = al(sizeof(J));
    int rc = msm_reserve(c, sz_prod + sz_part + sz_win);
    if (rc) return rc;
    char* p = (char*)c->msm_ws;
    u32* prod = (u32*)p; p += sz_prod;
    J* partial = (J*)p; p += sz_part;
    J* win = (J*)p;
    if ((rc = mul(sc, pts, pt_fmt, prod, n))) return rc;
    hipLaunchKernelGGL((sum_affine_kernel<C>), dim3(blocks), dim3(256), 0, c->stream, (const u32*)prod, n, partial);
    hipLaunchKernelGGL((sum_partials_kernel<C>), dim3(1), dim3(256), 0, c->stream, (const J*)partial, blocks, win);
    hipLaunchKernelGGL((finish_kernel<C>), dim3(1), dim3(64), 0, c->stream, (const J*)win, 1, 0, 0, out, out_fmt);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  const int cb = (int)c->opt[ECGPU_OPT_MSM_WINDOW_BITS];
  const bool wide = (cb == 19) || (cb != 16 && n >= wide_window_min<C>() && n < WIDE_WINDOW_MAX);
  return wide ? msm_buckets<C, 19>(c, sc, pts, pt_fmt, n, out, out_fmt) : msm_buckets<C, 16>(c, sc, pts, pt_fmt, n, out, out_fmt);
}

}  // namespace msm
}  // namespace ecgpu
