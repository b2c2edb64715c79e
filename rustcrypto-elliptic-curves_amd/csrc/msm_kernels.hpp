// Kernels and host launcher of the Pippenger MSM (see msm.hpp for the schedule), written once over the curve traits.
// Device code only.
#pragma once
#include "ecgpu_internal.hpp"
#include "kernels.hpp"
#include "msm.hpp"

namespace ecgpu {
namespace msm {

// ---------------------------------------------------------------------------------------------------------------------
// 0. The digits of every term, computed ONCE (the first version re-derived all digits of a term in each of the
//    histogram and scatter workgroups that looked at it).
//    secp256k1: the scalar is split by the endomorphism (k = k1 + k2 lambda, magnitudes below 2^128 and two signs); each
//    half gives 8 signed 16-bit digits in [-2^15, 2^15) and a carry digit in {0, 1} (window 8, a single bucket).
//    P-256 / P-384: one "half"; k > n/2 is replaced by n - k with the opposite sign (the carry window of the recoding is
//    then almost always empty; without this half of all terms land in its single bucket).
//    mag[(NHALF w + h) * ns + i] = |digit| of window w of half h (0 .. 2^15; window-major so that a workgroup streams its
//    window's digits, four 16-bit words per load; the row stride ns is n rounded up to a multiple of four),
//    sgn[i] bit NHALF w + h = the entry is subtracted (sign of the digit xor sign of the half).  Terms whose point is the
//    identity are not filtered here: the bucket sums skip them.
// ---------------------------------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(256) digits_kernel(const u32* scalars, size_t n, size_t ns, uint16_t* mag, u32* sgn) {
  constexpr int NW = C::NW, NHALF = Cfg<C>::NHALF, NWIN = Cfg<C>::NWIN;
  static_assert(Cfg<C>::NDIG <= 32, "one sign bit per digit column");
  ECGPU_GRID_STRIDE(i, n) {
    u32 k[NW], ord[NW];
    C::scalar_load(k, scalars + i * NW);
    C::order(ord);
    reduce_once<NW>(k, ord);
    u32 bits = 0;
    if constexpr (NHALF == 2) {
      k256::GlvSplit sp;
      k256::glv_split(sp, k);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const u32* m = h ? sp.k2 : sp.k1;
        const u32 neg = (h ? sp.neg2 : sp.neg1) ? 1u : 0u;
        u32 carry = 0;
#pragma unroll
        for (int w = 0; w < NWIN - 1; w++) {
          const u32 v = ((m[w >> 1] >> (16 * (w & 1))) & 0xFFFFu) + carry;
          carry = (v >= 0x8000u) ? 1u : 0u;      // v in [2^15, 2^16] becomes v - 2^16 with a carry
          const int d = (int)v - (int)(carry << 16);
          mag[(size_t)(2 * w + h) * ns + i] = (uint16_t)(d < 0 ? -d : d);
          bits |= (((d < 0) ? 1u : 0u) ^ neg) << (2 * w + h);
        }
        mag[(size_t)(2 * (NWIN - 1) + h) * ns + i] = (uint16_t)carry;
        bits |= neg << (2 * (NWIN - 1) + h);
      }
    } else {
      u32 t[NW];
      mp_sub<NW>(t, ord, k);                   // n - k
      const bool flip = !mp_geq<NW>(t, k);     // n - k < k
#pragma unroll
      for (int w = 0; w < NW; w++) k[w] = flip ? t[w] : k[w];
      const u32 neg = flip ? 1u : 0u;
      u32 carry = 0;
#pragma unroll
      for (int w = 0; w < NWIN - 1; w++) {
        const u32 v = ((k[w >> 1] >> (16 * (w & 1))) & 0xFFFFu) + carry;
        carry = (v >= 0x8000u) ? 1u : 0u;
        const int d = (int)v - (int)(carry << 16);
        mag[(size_t)w * ns + i] = (uint16_t)(d < 0 ? -d : d);
        bits |= (((d < 0) ? 1u : 0u) ^ neg) << w;
      }
      mag[(size_t)(NWIN - 1) * ns + i] = (uint16_t)carry;
      bits |= neg << (NWIN - 1);
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
    if constexpr (Cfg<C>::NHALF == 2) {
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
// A direct scatter into the 2^15 buckets of a window (the first version) writes every 4-byte entry to a different
// cache line, and a workgroup comes back to the same line only after it has touched ~32 768 others: the lines leave the
// L2 partly written, HBM sees 142 M masked partial writes, and the scatter ran at 3.3 ms for 0.57 GB of output.
// Sorting in two levels keeps the set of lines a workgroup is filling small enough for the L2 to merge them:
//   level A  a workgroup owns one window and one contiguous chunk of the terms and splits its entries into NCOARSE = 512
//            coarse bins of NFINE = 64 buckets (512 open lines per workgroup);
//   level B  a workgroup owns one coarse bin (its entries are contiguous after level A) and sorts it by the low six bits
//            of the bucket number (64 open lines), which also yields the bucket offsets.
// Every count and every cursor increment is an LDS atomic.  An entry is 32 bits: term index (24 bits, so a call is cut
// into slabs of 2^24 terms), the low six bucket bits (needed by level B only), the GLV half and the subtract flag.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int LOG_FINE = 6, NFINE = 1 << LOG_FINE, NCOARSE = NBUCKET / NFINE;
constexpr size_t SLAB_TERMS = (size_t)1 << 24;
constexpr u32 ENTRY_INDEX_MASK = 0x00FFFFFFu;          // entry = index | fine << 24 | half << 30 | subtract << 31
// chunk g of nch: boundaries are multiples of four terms (the loops below take four terms per step), the last chunk ends at n
__device__ __forceinline__ void chunk_range(size_t n, int g, int nch, size_t& lo, size_t& hi) {
  lo = (n * (size_t)g / nch) & ~(size_t)3;
  hi = (g == nch - 1) ? n : ((n * (size_t)(g + 1) / nch) & ~(size_t)3);
}
// the four 16-bit digits at terms i .. i + 3 of one row (i a multiple of four: one 8-byte load)
__device__ __forceinline__ void load_mag4(u32* a, const uint16_t* row, size_t i) {
  const uint2 v = *(const uint2*)(row + i);
  a[0] = v.x & 0xFFFFu; a[1] = v.x >> 16; a[2] = v.y & 0xFFFFu; a[3] = v.y >> 16;
}
// A1. part[w][g][cb] = number of entries of chunk g of window w in coarse bin cb
static __global__ void __launch_bounds__(1024) coarse_hist_kernel(const uint16_t* mag, size_t n, size_t ns, int nhalf, int nch, u32* part) {
  __shared__ u32 cnt[NCOARSE];
  const int w = blockIdx.x / nch, g = blockIdx.x % nch;
  for (int b = threadIdx.x; b < NCOARSE; b += 1024) cnt[b] = 0;
  __syncthreads();
  size_t lo, hi;
  chunk_range(n, g, nch, lo, hi);
#pragma unroll 1
  for (int h = 0; h < nhalf; h++) {
    const uint16_t* src = mag + (size_t)(nhalf * w + h) * ns;
    for (size_t i = lo + 4 * (size_t)threadIdx.x; i < hi; i += 4096) {       // the row is padded to ns: reading past n within it is safe
      u32 a[4];
      load_mag4(a, src, i);
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (a[q] && i + q < hi) atomicAdd(&cnt[(a[q] - 1) >> LOG_FINE], 1u);
    }
  }
  __syncthreads();
  u32* dst = part + ((size_t)w * nch + g) * NCOARSE;
  for (int b = threadIdx.x; b < NCOARSE; b += 1024) dst[b] = cnt[b];
}
// A2. totals over the chunks, exclusive scan over all ncb = NWIN * NCOARSE coarse bins (one workgroup), cursors per chunk
static __global__ void __launch_bounds__(256) coarse_totals_kernel(const u32* part, int ncb, int nch, u32* tot) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncb) return;
  const int w = j / NCOARSE, cb = j % NCOARSE;
  u32 s = 0;
#pragma unroll 1
  for (int g = 0; g < nch; g++) s += part[((size_t)w * nch + g) * NCOARSE + cb];
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
static __global__ void __launch_bounds__(256) coarse_cursors_kernel(u32* part, int ncb, int nch, const u32* coarse_off) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncb) return;
  const int w = j / NCOARSE, cb = j % NCOARSE;
  u32 run = coarse_off[j];
#pragma unroll 1
  for (int g = 0; g < nch; g++) {
    u32* p = part + ((size_t)w * nch + g) * NCOARSE + cb;
    const u32 c = *p;
    *p = run;
    run += c;
  }
}
// A3. entries into their coarse bins; same workgroup -> (window, chunk) map as the histogram.  The carry window (the last
//     one) holds a single bucket, so its entries are final after this level and go straight to `sorted`.
static __global__ void __launch_bounds__(1024) coarse_scatter_kernel(const uint16_t* mag, const u32* sgn, size_t n, size_t ns, int nhalf, int nwin, int nch,
                                                              const u32* part, u32* mid, u32* sorted) {
  __shared__ u32 cur[NCOARSE];
  const int w = blockIdx.x / nch, g = blockIdx.x % nch;
  const u32* src = part + ((size_t)w * nch + g) * NCOARSE;
  for (int b = threadIdx.x; b < NCOARSE; b += 1024) cur[b] = src[b];
  __syncthreads();
  size_t lo, hi;
  chunk_range(n, g, nch, lo, hi);
  u32* dst = (w == nwin - 1) ? sorted : mid;
#pragma unroll 1
  for (int h = 0; h < nhalf; h++) {
    const uint16_t* m = mag + (size_t)(nhalf * w + h) * ns;
    for (size_t i = lo + 4 * (size_t)threadIdx.x; i < hi; i += 4096) {       // four terms per step: the loads go out together
      u32 a[4];
      load_mag4(a, m, i);
      const uint4 sv = *(const uint4*)(sgn + i);                                  // sgn is padded to ns words as well
      const u32 sg[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (a[q] && i + q < hi) {
          const u32 b = a[q] - 1;
          const u32 pos = atomicAdd(&cur[b >> LOG_FINE], 1u);
          dst[pos] = (u32)(i + q) | ((b & (NFINE - 1)) << 24) | ((u32)h << 30) | (((sg[q] >> (nhalf * w + h)) & 1u) << 31);
        }
      }
    }
  }
}
// B. one workgroup per coarse bin: count its entries per bucket, scan the 64 counts (which are the bucket offsets of the
//    whole sort: offsets[(w * NCOARSE + cb) * NFINE + f] is bucket w * NBUCKET + cb * NFINE + f), place the entries.
static __global__ void __launch_bounds__(256) fine_sort_kernel(const u32* mid, const u32* coarse_off, int nwin, u32* offsets, u32* sorted) {
  __shared__ u32 cnt[NFINE], cur[NFINE];
  const int j = blockIdx.x, t = threadIdx.x;
  const u32 lo = coarse_off[j], hi = coarse_off[j + 1];
  if (j >= (nwin - 1) * NCOARSE) {             // carry window: every entry of the bin is in its first bucket, already in place
    if (t < NFINE) offsets[(size_t)j * NFINE + t] = (t == 0) ? lo : hi;
    return;
  }
  if (t < NFINE) cnt[t] = 0;
  __syncthreads();
  // entries lo .. hi: a scalar head up to the next multiple of four, then 16-byte loads
  const u32 lo4 = (lo + 3u) & ~3u, head = (lo4 < hi ? lo4 : hi);
  if (lo + t < head) atomicAdd(&cnt[(mid[lo + t] >> 24) & (NFINE - 1)], 1u);
  for (u32 e = head + 4 * t; e < hi; e += 1024) {
    const uint4 v4 = *(const uint4*)(mid + e);                  // mid is padded by four entries
    const u32 v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (e + q < hi) atomicAdd(&cnt[(v[q] >> 24) & (NFINE - 1)], 1u);
  }
  __syncthreads();
  if (t < NFINE) {                             // exclusive scan of 64 counts in the first wave
    const u32 c = cnt[t];
    u32 incl = c;
#pragma unroll
    for (int off = 1; off < NFINE; off <<= 1) {
      const u32 v = __shfl_up(incl, off);
      if (t >= off) incl += v;
    }
    const u32 start = lo + incl - c;
    cur[t] = start;
    offsets[(size_t)j * NFINE + t] = start;
  }
  __syncthreads();
  if (lo + t < head) {
    const u32 v = mid[lo + t];
    sorted[atomicAdd(&cur[(v >> 24) & (NFINE - 1)], 1u)] = v;
  }
  for (u32 e = head + 4 * t; e < hi; e += 1024) {
    const uint4 v4 = *(const uint4*)(mid + e);
    const u32 v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (e + q < hi) sorted[atomicAdd(&cur[(v[q] >> 24) & (NFINE - 1)], 1u)] = v[q];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 4. bucket sums.  A bucket with more than `cap` entries would serialise the launch on a few lanes (equal scalars - a
//    plain sum of points is an MSM with all scalars 1 - put every term of a window into one bucket), so such buckets are
//    only registered here: they are cut into chunks of `cap` entries, each chunk is summed by a whole workgroup
//    (heavy_chunk_kernel) and the chunk sums are folded per bucket (heavy_finish_kernel).  With uniform scalars only the
//    carry windows' single buckets are heavy.
// ---------------------------------------------------------------------------------------------------------------------
struct HeavyBucket { u32 bucket, base, chunks; };
struct HeavyChunk { u32 bucket, index; };

// the prepared point an entry names: 2 NW words in the field's internal form (16-byte loads)
template <class C>
struct RawPoint { uint4 v[C::NW / 2]; };
template <class C>
__device__ __forceinline__ RawPoint<C> entry_point(const u32* prep, size_t n, u32 e) {
  const uint4* src = (const uint4*)(prep + ((size_t)((e >> 30) & 1u) * n + (e & ENTRY_INDEX_MASK)) * 2 * C::NW);
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
// acc += the points of entries s .. e, software-pipelined: the gather of entry q + 1 is in flight during the addition of entry q
template <class C>
__device__ __forceinline__ void bucket_accumulate_run(Xyzz<C>& acc, const u32* prep, size_t n, const u32* sorted, u32 s, u32 e) {
  if (s >= e) return;
  u32 en = sorted[s];
  RawPoint<C> pn = entry_point<C>(prep, n, en);
#pragma unroll 1
  for (u32 q = s; q < e; q++) {
    const u32 ec = en;
    const RawPoint<C> pc = pn;
    if (q + 1 < e) {
      en = sorted[q + 1];
      pn = entry_point<C>(prep, n, en);
    }
    bucket_add_raw<C>(acc, pc, ec);
  }
}
// One lane per (bucket, part): a bucket's run of entries is cut into `split` equal parts summed by `split` neighbouring
// lanes, and bucket_combine_kernel adds the parts.  With one lane per bucket the 294 912 buckets of a 2^23-term k256 sum
// were 1.125 x the 262 144 lanes the chip holds at this kernel's occupancy - a second, almost empty round as long as the
// first; eight parts per bucket make it nine full rounds of shorter tasks, handed out by the dispatcher as CUs free up.
template <class C>
__global__ void __launch_bounds__(256, 4) bucket_sum_kernel(const u32* prep, size_t n, const u32* offsets, const u32* sorted, Jac<C>* parts, int nb, int split,
                                                            u32 cap, u32* heavy_ctr, HeavyBucket* heavy, HeavyChunk* chunks) {
  ECGPU_GRID_STRIDE(t, (size_t)nb * split) {
    const size_t b = t / split;
    const u32 j = (u32)(t % split);
    Xyzz<C> acc;
    xyzz_set_infinity<C>(acc);
    const u32 lo = offsets[b], hi = offsets[b + 1], len = hi - lo;
    if (len > cap) {
      if (j == 0) {
        const u32 k = (len + cap - 1) / cap;
        const u32 idx = atomicAdd(&heavy_ctr[0], 1u);
        const u32 base = atomicAdd(&heavy_ctr[1], k);
        heavy[idx] = HeavyBucket{(u32)b, base, k};
#pragma unroll 1
        for (u32 q = 0; q < k; q++) chunks[base + q] = HeavyChunk{(u32)b, q};
      }
      continue;                                   // parts of a heavy bucket are not read: heavy_finish_kernel writes its sum
    }
    const u32 s = lo + (u32)(((u64)len * j) / split), e = lo + (u32)(((u64)len * (j + 1)) / split);
    bucket_accumulate_run<C>(acc, prep, n, sorted, s, e);
    Jac<C> r;
    xyzz_to_jacobian<C>(r, acc);
    parts[t] = r;
  }
}
// buckets[b] = sum of its parts (general additions; a heavy bucket's sum is written by heavy_finish_kernel)
template <class C>
__global__ void __launch_bounds__(256) bucket_combine_kernel(const Jac<C>* parts, Jac<C>* buckets, int nb, int split, const u32* offsets, u32 cap) {
  ECGPU_GRID_STRIDE(b, (size_t)nb) {
    if (offsets[b + 1] - offsets[b] > cap) continue;
    Jac<C> acc = parts[b * split];
#pragma unroll 1
    for (int j = 1; j < split; j++) pt_add<C>(acc, acc, parts[b * split + j]);
    buckets[b] = acc;
  }
}

// sum over the lanes of a workgroup through LDS (count = blockDim.x, a power of two)
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
// 4b. one workgroup per chunk of a heavy bucket: lanes stride over the chunk, LDS tree sum
template <class C>
__global__ void __launch_bounds__(256) heavy_chunk_kernel(const u32* prep, size_t n, const u32* offsets, const u32* sorted, u32 cap, const u32* heavy_ctr,
                                                          const HeavyChunk* chunks, Jac<C>* partial) {
  __shared__ Jac<C> sh[256];
  const u32 total = heavy_ctr[1];
  for (u32 c = blockIdx.x; c < total; c += gridDim.x) {
    const HeavyChunk ch = chunks[c];
    const u32 lo = offsets[ch.bucket] + ch.index * cap;
    const u32 end = offsets[ch.bucket + 1];
    const u32 hi = (end - lo > cap) ? lo + cap : end;
    Xyzz<C> xacc;
    xyzz_set_infinity<C>(xacc);
#pragma unroll 1
    for (u32 j = lo + threadIdx.x; j < hi; j += 256) { const u32 e = sorted[j]; bucket_add_raw<C>(xacc, entry_point<C>(prep, n, e), e); }
    Jac<C> acc;
    xyzz_to_jacobian<C>(acc, xacc);
    lds_tree_sum<C>(sh, acc, threadIdx.x, 256);
    if (threadIdx.x == 0) partial[c] = acc;
  }
}
// 4c. one workgroup per heavy bucket: fold its chunk sums
template <class C>
__global__ void __launch_bounds__(256) heavy_finish_kernel(const u32* heavy_ctr, const HeavyBucket* heavy, const Jac<C>* partial, Jac<C>* buckets) {
  __shared__ Jac<C> sh[256];
  const u32 total = heavy_ctr[0];
  for (u32 h = blockIdx.x; h < total; h += gridDim.x) {
    const HeavyBucket hb = heavy[h];
    Jac<C> acc;
    jac::set_infinity<C>(acc);
#pragma unroll 1
    for (u32 j = threadIdx.x; j < hb.chunks; j += 256) pt_add<C>(acc, acc, partial[hb.base + j]);
    lds_tree_sum<C>(sh, acc, threadIdx.x, 256);
    if (threadIdx.x == 0) buckets[hb.bucket] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 5. Weighted sums by a tree of running sums.  For a run of L points B_0..B_{L-1},
//      T = sum B_t   and   Wt = sum (t + 1) B_t
//    come from 2L additions (run += B_t from the top, wt += run).  Level 0 does this for every SEG0
//    consecutive buckets; level 1 for every SEG1 consecutive level-0 results; the window kernel
//    finishes.  With weights nested as j + 1 = (s1 * SEG1 + s0) * SEG0 + t + 1:
//      S = sum_j (j+1) B_j = sum Wt0 + SEG0 * [ sum_{s1} ( (Wt1 - T1) ) + SEG1 * sum_{s1} s1 * T1 ]
//    where T1/Wt1 are the level-1 sums over the level-0 totals T0 (Wt1 weights them 1..SEG1).
// ---------------------------------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(64) segment_kernel(const Jac<C>* in, Jac<C>* out_t, Jac<C>* out_w, int len, int total) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  const Jac<C>* B = in + (size_t)s * len;
  Jac<C> run, wt;
  jac::set_infinity<C>(run);
  wt = run;
#pragma unroll 1
  for (int t = len - 1; t >= 0; t--) {
    pt_add<C>(run, run, B[t]);
    pt_add<C>(wt, wt, run);
  }
  out_t[s] = run;
  out_w[s] = wt;
}
// plain sums of `len` consecutive points (for the sum of the level-0 weighted parts)
template <class C>
__global__ void __launch_bounds__(64) sum_kernel(const Jac<C>* in, Jac<C>* out, int len, int total) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  Jac<C> acc;
  jac::set_infinity<C>(acc);
#pragma unroll 1
  for (int t = 0; t < len; t++) pt_add<C>(acc, acc, in[(size_t)s * len + t]);
  out[s] = acc;
}

// 6. per window: combine NSEG1 level-1 results
//      S_w = sumW0 + SEG0 * ( sum_{s1} (Wt1 - T1)  +  SEG1 * sum_{s1} s1 * T1 )
//    One workgroup per window, lane s1 owns one level-1 segment (and one of the NSUMW partial sums of the level-0
//    weighted parts); the three sums over the lanes are LDS tree reductions, so the dependent chain is
//    log2(NSEG1) additions instead of NSEG1 x 4.
template <class C>
__global__ void __launch_bounds__(NSEG1) window_kernel(const Jac<C>* t1, const Jac<C>* w1, const Jac<C>* sumw0, Jac<C>* win) {
  static_assert(NSUMW == NSEG1 && NSEG1 <= 1024, "one lane per level-1 segment and per partial sum");
  __shared__ Jac<C> sh[NSEG1];
  const int w = blockIdx.x, s = threadIdx.x;          // blockDim.x == NSEG1
  Jac<C> inner, acc, sw;
  jac::set_infinity<C>(acc);
  const Jac<C> T = t1[w * NSEG1 + s];
  Jac<C> neg = T;
  C::fe_neg(neg.y, neg.y);
  pt_add<C>(inner, w1[w * NSEG1 + s], neg);            // Wt1 - T1
  // s * T by double-and-add over the bits of s
#pragma unroll 1
  for (int bit = LOG_NSEG1 - 1; bit >= 0; bit--) {
    pt_dbl<C>(acc);
    if ((s >> bit) & 1) pt_add<C>(acc, acc, T);
  }
  sw = sumw0[w * NSUMW + s];
  lds_tree_sum<C>(sh, inner, s, NSEG1);
  lds_tree_sum<C>(sh, acc, s, NSEG1);
  lds_tree_sum<C>(sh, sw, s, NSEG1);
  if (s == 0) {
#pragma unroll 1
    for (int j = 0; j < LOG_SEG1; j++) pt_dbl<C>(acc);
    pt_add<C>(acc, acc, inner);
#pragma unroll 1
    for (int j = 0; j < LOG_SEG0; j++) pt_dbl<C>(acc);
    pt_add<C>(acc, acc, sw);
    win[w] = acc;
  }
}
// window sums of a further slab of terms are added to the running window sums
template <class C>
__global__ void __launch_bounds__(64) windows_accumulate_kernel(Jac<C>* total, const Jac<C>* slab, int nwin) {
  const int w = threadIdx.x;
  if (w < nwin) { Jac<C> a = total[w]; pt_add<C>(a, a, slab[w]); total[w] = a; }
}

// Small sums (n below SMALL_MSM_TERMS): the bucket method has a fixed cost of ~2 ms (the buckets to reduce, the serial
// doublings), more than n plain scalar multiplications take, so those run through the variable-base kernel and
// the n products are summed here: one workgroup per slice (lanes stride, LDS tree), then one workgroup over the slice
// sums.  Measured on k256 (ms, this path / buckets): 2^10 1.1 / 2.2, 2^14 1.2 / 2.3, 2^16 1.3 / 2.5, 2^18 3.1 / 2.9.
constexpr size_t SMALL_MSM_TERMS = (size_t)3 << 16;
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

// 7. Horner over the windows, conversion to affine, output
template <class C>
__global__ void __launch_bounds__(64) finish_kernel(const Jac<C>* win, int nwin, u32* out, int out_fmt) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  constexpr int NW = C::NW;
  using Fe = typename C::Fe;
  Jac<C> r = win[nwin - 1];
#pragma unroll 1
  for (int w = nwin - 2; w >= 0; w--) {
#pragma unroll 1
    for (int j = 0; j < CBITS; j++) pt_dbl<C>(r);
    pt_add<C>(r, r, win[w]);
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
// Host side: all stages on c->stream out of one grow-only workspace.  `mul` is the curve's batch scalar multiplication
// (used for small sums).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef MSM_BUCKET_WGS_PER_CU
#define MSM_BUCKET_WGS_PER_CU 64     // cap on bucket-sum workgroups per CU: above tasks / 256, so every lane takes one task and the dispatcher balances the CUs
#endif
#ifndef MSM_BUCKET_SPLIT
#define MSM_BUCKET_SPLIT 8           // lanes per bucket (bucket_sum_kernel)
#endif
template <class C, class MulFn>
static int msm_run(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t n, u32* out, int out_fmt, MulFn mul) {
  constexpr int NW = C::NW, NHALF = Cfg<C>::NHALF, NWIN = Cfg<C>::NWIN, NDIG = Cfg<C>::NDIG;
  constexpr int NCB = NWIN * NCOARSE;
  using J = Jac<C>;
  static_assert((size_t)NDIG * SLAB_TERMS < ((size_t)1 << 32) && SLAB_TERMS - 1 <= ENTRY_INDEX_MASK, "32-bit offsets and 24-bit term indices within a slab");
  if (((uintptr_t)pts & 15) || ((uintptr_t)sc & 3)) return ecgpu_set_err(c, ECGPU_ERR_ARG, "ecgpu_msm: device points must be 16-byte aligned");
  const size_t nb = (size_t)NWIN * NBUCKET;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  auto reserve = [&](size_t need) -> int {
    if (need > c->msm_ws_cap) {
      if (c->msm_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->msm_ws)); c->msm_ws = nullptr; c->msm_ws_cap = 0; }
      HIPCHK(c, hipMalloc(&c->msm_ws, need));
      c->msm_ws_cap = need;
    }
    return 0;
  };
  // ECGPU_MSM_SMALL = 0 forces the bucket method for every size (measurements, tests of the bucket path on small inputs)
  // (read per call so that one process can exercise both paths)
  const char* small_env = getenv("ECGPU_MSM_SMALL");
  const bool small_path = !(small_env && atoi(small_env) == 0);
  if (small_path && n > 0 && n < SMALL_MSM_TERMS) {
    // n scalar multiplications on the throughput kernel, then a two-level sum of the products
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    const size_t sz_prod = al(n * 8 * NW), sz_part = al((size_t)blocks * sizeof(J)), sz_win = al(sizeof(J));
    int rc = reserve(sz_prod + sz_part + sz_win);
    if (rc) return rc;
    char* p = (char*)c->msm_ws;
    u32* prod = (u32*)p; p += sz_prod;
    J* partial = (J*)p; p += sz_part;
    J* win = (J*)p;
    if ((rc = mul(sc, pts, pt_fmt, prod, n))) return rc;
    hipLaunchKernelGGL((sum_affine_kernel<C>), dim3(blocks), dim3(256), 0, c->stream, (const u32*)prod, n, partial);
    hipLaunchKernelGGL((sum_partials_kernel<C>), dim3(1), dim3(256), 0, c->stream, (const J*)partial, blocks, win);
    hipLaunchKernelGGL((finish_kernel<C>), dim3(1), dim3(64), 0, c->stream, (const J*)win, 1, out, out_fmt);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  // Large sums run in slabs of at most SLAB_TERMS terms (a sorted entry keeps the term index in 24 bits); every slab goes
  // through the whole pipeline down to its NWIN window sums, which are added up before the final Horner pass.
  // ECGPU_MSM_SLAB overrides the slab size (tests exercise the slab loop on small inputs).
  const char* slab_env = getenv("ECGPU_MSM_SLAB");
  size_t slab = slab_env ? (size_t)atoll(slab_env) : SLAB_TERMS;
  if (slab < 1024 || slab > SLAB_TERMS) slab = SLAB_TERMS;
  const size_t m = n < slab ? n : slab;                // terms of the largest slab: sizes the workspace
  const size_t sz_aff = (pt_fmt == FMT_PROJECTIVE) ? al(m * 8 * NW) : 0, sz_prep = al((size_t)NHALF * m * 8 * NW);
  const size_t sz_off = al((nb + 1) * 4), sz_coff = al((size_t)(NCB + 1) * 4), sz_tot = al((size_t)NCB * 4), sz_sorted = al((size_t)NDIG * m * 4 + 32);
  // level A of the sort: one 1024-thread workgroup per CU, the chunks of a window side by side
  // (ECGPU_MSM_CHUNK_WGS workgroups per CU, default 1: measurements)
  static const int chunk_wgs = [] { const char* e = getenv("ECGPU_MSM_CHUNK_WGS"); int v = e ? atoi(e) : 1; return (v < 1 || v > 8) ? 1 : v; }();
  const int nch = (chunk_wgs * c->num_cus - 1) / NWIN > 0 ? (chunk_wgs * c->num_cus - 1) / NWIN : 1;
  const size_t sz_part = al((size_t)NWIN * nch * NCOARSE * 4);
  const size_t ms = (m + 3) & ~(size_t)3;              // row stride of the digit arrays: four terms per load
  const size_t sz_mag = al((size_t)NDIG * ms * 2), sz_sgn = al(ms * 4);
  static const int wgs_per_cu = [] { const char* e = getenv("ECGPU_MSM_WGS"); int v = e ? atoi(e) : MSM_BUCKET_WGS_PER_CU; return (v < 1 || v > 1024) ? MSM_BUCKET_WGS_PER_CU : v; }();
  static const int split = [] { const char* e = getenv("ECGPU_MSM_SPLIT"); int v = e ? atoi(e) : MSM_BUCKET_SPLIT; return (v < 1 || v > 64) ? MSM_BUCKET_SPLIT : v; }();
  const size_t sz_buckets = al(nb * sizeof(J)), sz_parts = al(nb * split * sizeof(J));
  const size_t n0 = (size_t)NWIN * NSEG0, n1 = (size_t)NWIN * NSEG1, nsw = (size_t)NWIN * NSUMW;
  const size_t sz_l0 = al(n0 * sizeof(J)), sz_l1 = al(n1 * sizeof(J)), sz_sw = al(nsw * sizeof(J)), sz_win = al(NWIN * sizeof(J));
  // heavy buckets (more than `cap` entries): at most L / cap of them, at most 2 L / cap + 1 chunks
  const size_t L = (size_t)NDIG * m;
  const u32 cap = (u32)((8 * NHALF * (m / NBUCKET) > 2048) ? 8 * NHALF * (m / NBUCKET) : 2048);
  const size_t hmax = L / cap + 1, cmax = 2 * (L / cap) + 2;
  const size_t sz_hctr = al(8), sz_heavy = al(hmax * sizeof(HeavyBucket)), sz_chunks = al(cmax * sizeof(HeavyChunk)), sz_partial = al(cmax * sizeof(J));
  const size_t need = sz_aff + sz_prep + sz_mag + sz_sgn + sz_off + sz_coff + sz_tot + sz_part + 2 * sz_sorted + sz_buckets + sz_parts + 2 * sz_l0 + 2 * sz_l1 +
                      sz_sw + 2 * sz_win + sz_hctr + sz_heavy + sz_chunks + sz_partial;
  int rc = reserve(need);
  if (rc) return rc;
  char* p = (char*)c->msm_ws;
  u32* aff = (u32*)p; p += sz_aff;
  u32* prep = (u32*)p; p += sz_prep;
  uint16_t* mag = (uint16_t*)p; p += sz_mag;
  u32* sgn = (u32*)p; p += sz_sgn;
  u32* offsets = (u32*)p; p += sz_off;
  u32* coarse_off = (u32*)p; p += sz_coff;
  u32* tot = (u32*)p; p += sz_tot;
  u32* part = (u32*)p; p += sz_part;
  u32* mid = (u32*)p; p += sz_sorted;
  u32* sorted = (u32*)p; p += sz_sorted;
  J* buckets = (J*)p; p += sz_buckets;
  J* parts = (J*)p; p += sz_parts;
  J* t0 = (J*)p; p += sz_l0;
  J* w0 = (J*)p; p += sz_l0;
  J* t1 = (J*)p; p += sz_l1;
  J* w1 = (J*)p; p += sz_l1;
  J* sumw0 = (J*)p; p += sz_sw;
  J* win = (J*)p; p += sz_win;
  J* win_slab = (J*)p; p += sz_win;
  u32* heavy_ctr = (u32*)p; p += sz_hctr;
  HeavyBucket* heavy = (HeavyBucket*)p; p += sz_heavy;
  HeavyChunk* chunks = (HeavyChunk*)p; p += sz_chunks;
  J* partial = (J*)p;
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
    hipLaunchKernelGGL((digits_kernel<C>), dim3(ecgpu_grid_for(c, cnt, 8)), dim3(256), 0, c->stream, ssc, cnt, ms, mag, sgn);
    hipLaunchKernelGGL((prepare_points_kernel<C>), dim3(ecgpu_grid_for(c, cnt, 8)), dim3(256), 0, c->stream, xy, prep, cnt);
    hipLaunchKernelGGL(coarse_hist_kernel, dim3((unsigned)(NWIN * nch)), dim3(1024), 0, c->stream, (const uint16_t*)mag, cnt, ms, NHALF, nch, part);
    hipLaunchKernelGGL(coarse_totals_kernel, dim3(cb_grid), dim3(256), 0, c->stream, (const u32*)part, NCB, nch, tot);
    hipLaunchKernelGGL(coarse_scan_kernel, dim3(1), dim3(1024), 0, c->stream, (const u32*)tot, NCB, coarse_off, offsets + nb);
    hipLaunchKernelGGL(coarse_cursors_kernel, dim3(cb_grid), dim3(256), 0, c->stream, part, NCB, nch, (const u32*)coarse_off);
    hipLaunchKernelGGL(coarse_scatter_kernel, dim3((unsigned)(NWIN * nch)), dim3(1024), 0, c->stream, (const uint16_t*)mag, (const u32*)sgn, cnt, ms, NHALF, NWIN, nch,
                       (const u32*)part, mid, sorted);
    hipLaunchKernelGGL(fine_sort_kernel, dim3((unsigned)NCB), dim3(256), 0, c->stream, (const u32*)mid, (const u32*)coarse_off, NWIN, offsets, sorted);
    HIPCHK(c, hipMemsetAsync(heavy_ctr, 0, 8, c->stream));
    hipLaunchKernelGGL((bucket_sum_kernel<C>), dim3(ecgpu_grid_for(c, nb * split, wgs_per_cu)), dim3(256), 0, c->stream, (const u32*)prep, cnt, (const u32*)offsets,
                       (const u32*)sorted, parts, (int)nb, split, cap, heavy_ctr, heavy, chunks);
    hipLaunchKernelGGL((heavy_chunk_kernel<C>), dim3((unsigned)c->num_cus * 8), dim3(256), 0, c->stream, (const u32*)prep, cnt, (const u32*)offsets, (const u32*)sorted, cap,
                       (const u32*)heavy_ctr, (const HeavyChunk*)chunks, partial);
    hipLaunchKernelGGL((heavy_finish_kernel<C>), dim3((unsigned)c->num_cus), dim3(256), 0, c->stream, (const u32*)heavy_ctr, (const HeavyBucket*)heavy, (const J*)partial,
                       buckets);
    hipLaunchKernelGGL((bucket_combine_kernel<C>), dim3(ecgpu_grid_for(c, nb, 8)), dim3(256), 0, c->stream, (const J*)parts, buckets, (int)nb, split, (const u32*)offsets, cap);
    hipLaunchKernelGGL((segment_kernel<C>), dim3((unsigned)((n0 + 63) / 64)), dim3(64), 0, c->stream, (const J*)buckets, t0, w0, SEG0, (int)n0);
    hipLaunchKernelGGL((segment_kernel<C>), dim3((unsigned)((n1 + 63) / 64)), dim3(64), 0, c->stream, (const J*)t0, t1, w1, SEG1, (int)n1);
    hipLaunchKernelGGL((sum_kernel<C>), dim3((unsigned)((nsw + 63) / 64)), dim3(64), 0, c->stream, (const J*)w0, sumw0, SUMW_LEN, (int)nsw);
    hipLaunchKernelGGL((window_kernel<C>), dim3(NWIN), dim3(NSEG1), 0, c->stream, (const J*)t1, (const J*)w1, (const J*)sumw0, wdst);
    if (s0 != 0) hipLaunchKernelGGL((windows_accumulate_kernel<C>), dim3(1), dim3(64), 0, c->stream, win, (const J*)win_slab, NWIN);
  }
  hipLaunchKernelGGL((finish_kernel<C>), dim3(1), dim3(64), 0, c->stream, (const J*)win, (int)NWIN, out, out_fmt);
  HIPCHK(c, hipGetLastError());
  return 0;
}

}  // namespace msm
}  // namespace ecgpu
