// Kernels of the k256 Pippenger MSM (see msm_k256.hpp for the schedule).  Device code only.
#pragma once
#include "kernels.hpp"
#include "msm_k256.hpp"

namespace ecgpu {
namespace msm {

// 0. The digits of every term, computed ONCE (the first version re-derived all digits of a term in each of the
//    histogram and scatter workgroups that looked at it).  The scalar is split by the endomorphism (k = k1 + k2 lambda,
//    magnitudes below 2^128 and two signs); each half gives 8 signed 16-bit digits in [-2^15, 2^15) and a carry digit
//    in {0, 1} (window 8, a single bucket).  mag[(2 w + h) * ns + i] = |digit| of window w of half h (0 .. 2^15; window-
//    major so that a workgroup streams its window's digits, four 16-bit words per load; the row stride ns is n rounded
//    up to a multiple of four), sgn[i] bit 2 w + h = the entry
//    is subtracted (sign of the digit xor sign of the half).  Terms whose point is the identity are not filtered here:
//    the bucket sums skip them.
__global__ void __launch_bounds__(256) digits_kernel(const u32* scalars, size_t n, size_t ns, uint16_t* mag, u32* sgn) {
  ECGPU_GRID_STRIDE(i, n) {
    u32 k[8];
    words_load_be<8>(k, scalars + i * 8);
    k256::scalar_reduce_once(k);
    k256::GlvSplit sp;
    k256::glv_split(sp, k);
    u32 bits = 0;
#pragma unroll
    for (int h = 0; h < NHALF; h++) {
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
    sgn[i] = bits;
  }
}
// lambda P = (beta x, y) for every point (k256 projective.rs:287-293), in the wire format of the points themselves
// (the identity, all zeros, stays all zeros)
__global__ void __launch_bounds__(256) endo_points_kernel(const u32* xy, u32* out, size_t n) {
  ECGPU_GRID_STRIDE(i, n) {
    const uint4* src = (const uint4*)(xy + i * 16);          // 16-byte loads: a point is 64 contiguous bytes per lane
    uint4* dst = (uint4*)(out + i * 16);
    const uint4 a0 = src[0], a1 = src[1], y0 = src[2], y1 = src[3];
    const u32 w[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    FeK256 x, b;
    k256::from_be_words(x, w);
    k256::beta(b);
    k256::mul(x, x, b);
    u32 o[8];
    CurveK256::fe_store(o, x);
    dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
    dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    dst[2] = y0;
    dst[3] = y1;
  }
}

// Two-level counting sort of the (half-term, window) entries by bucket, privatised in LDS.
//
// A direct scatter into the 2^15 buckets of a window (the previous version) writes every 4-byte entry to a different
// cache line, and a workgroup comes back to the same line only after it has touched ~32 768 others: the lines leave the
// L2 partly written, HBM sees 142 M masked partial writes, and the scatter ran at 3.3 ms for 0.57 GB of output
// (profiles/r02_msm_*).  Sorting in two levels keeps the set of lines a workgroup is filling small enough for the L2 to
// merge them:
//   level A  a workgroup owns one window and one contiguous chunk of the terms and splits its entries into NCOARSE = 512
//            coarse bins of NFINE = 64 buckets (512 open lines per workgroup);
//   level B  a workgroup owns one coarse bin (its entries are contiguous after level A) and sorts it by the low six bits
//            of the bucket number (64 open lines), which also yields the bucket offsets.
// Every count and every cursor increment is an LDS atomic.  An entry is 32 bits: term index (24 bits, so a call is cut
// into slabs of 2^24 terms), the low six bucket bits (needed by level B only), the GLV half and the subtract flag.
constexpr int LOG_FINE = 6, NFINE = 1 << LOG_FINE, NCOARSE = NBUCKET / NFINE;
constexpr int NCB = NWIN * NCOARSE;                    // coarse bins over all windows
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
__global__ void __launch_bounds__(1024) coarse_hist_kernel(const uint16_t* mag, size_t n, size_t ns, int nch, u32* part) {
  __shared__ u32 cnt[NCOARSE];
  const int w = blockIdx.x / nch, g = blockIdx.x % nch;
  for (int b = threadIdx.x; b < NCOARSE; b += 1024) cnt[b] = 0;
  __syncthreads();
  size_t lo, hi;
  chunk_range(n, g, nch, lo, hi);
#pragma unroll 1
  for (int h = 0; h < NHALF; h++) {
    const uint16_t* src = mag + (size_t)(2 * w + h) * ns;
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
// A2. totals over the chunks, exclusive scan over all coarse bins (one workgroup: 4 608 values), cursors per chunk
__global__ void __launch_bounds__(256) coarse_totals_kernel(const u32* part, int nch, u32* tot) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= NCB) return;
  const int w = j / NCOARSE, cb = j % NCOARSE;
  u32 s = 0;
#pragma unroll 1
  for (int g = 0; g < nch; g++) s += part[((size_t)w * nch + g) * NCOARSE + cb];
  tot[j] = s;
}
__global__ void __launch_bounds__(1024) coarse_scan_kernel(const u32* tot, u32* coarse_off, u32* total_entries) {
  __shared__ u32 psum[1024];
  constexpr int PER = (NCB + 1023) / 1024;
  const int t = threadIdx.x;
  u32 s = 0;
#pragma unroll
  for (int q = 0; q < PER; q++) { const int j = t * PER + q; if (j < NCB) s += tot[j]; }
  psum[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    u32 v = (t >= off) ? psum[t - off] : 0;
    __syncthreads();
    psum[t] += v;
    __syncthreads();
  }
  u32 run = (t == 0) ? 0 : psum[t - 1];
#pragma unroll
  for (int q = 0; q < PER; q++) { const int j = t * PER + q; if (j < NCB) { coarse_off[j] = run; run += tot[j]; } }
  if (t == 1023) { coarse_off[NCB] = psum[1023]; *total_entries = psum[1023]; }    // one past the end: the number of sorted entries
}
__global__ void __launch_bounds__(256) coarse_cursors_kernel(u32* part, int nch, const u32* coarse_off) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= NCB) return;
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
// A3. entries into their coarse bins; same workgroup -> (window, chunk) map as the histogram.  The carry window holds a
//     single bucket, so its entries are final after this level and go straight to `sorted`.
__global__ void __launch_bounds__(1024) coarse_scatter_kernel(const uint16_t* mag, const u32* sgn, size_t n, size_t ns, int nch, const u32* part, u32* mid,
                                                              u32* sorted) {
  __shared__ u32 cur[NCOARSE];
  const int w = blockIdx.x / nch, g = blockIdx.x % nch;
  const u32* src = part + ((size_t)w * nch + g) * NCOARSE;
  for (int b = threadIdx.x; b < NCOARSE; b += 1024) cur[b] = src[b];
  __syncthreads();
  size_t lo, hi;
  chunk_range(n, g, nch, lo, hi);
  u32* dst = (w == NWIN - 1) ? sorted : mid;
#pragma unroll 1
  for (int h = 0; h < NHALF; h++) {
    const uint16_t* m = mag + (size_t)(2 * w + h) * ns;
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
          dst[pos] = (u32)(i + q) | ((b & (NFINE - 1)) << 24) | ((u32)h << 30) | (((sg[q] >> (2 * w + h)) & 1u) << 31);
        }
      }
    }
  }
}
// B. one workgroup per coarse bin: count its entries per bucket, scan the 64 counts (which are the bucket offsets of the
//    whole sort: offsets[(w * NCOARSE + cb) * NFINE + f] is bucket w * NBUCKET + cb * NFINE + f), place the entries.
__global__ void __launch_bounds__(256) fine_sort_kernel(const u32* mid, const u32* coarse_off, u32* offsets, u32* sorted) {
  __shared__ u32 cnt[NFINE], cur[NFINE];
  const int j = blockIdx.x, t = threadIdx.x;
  const u32 lo = coarse_off[j], hi = coarse_off[j + 1];
  if (j >= (NWIN - 1) * NCOARSE) {             // carry window: every entry of the bin is in its first bucket, already in place
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
// window sums of a further slab of terms are added to the running window sums
__global__ void __launch_bounds__(64) windows_accumulate_kernel(JacK256* total, const JacK256* slab) {
  const int w = threadIdx.x;
  if (w < NWIN) { JacK256 a = total[w]; jac_add(a, a, slab[w]); total[w] = a; }
}

// 4. bucket sums.  One lane per bucket sums its points (Jacobian accumulator, mixed additions, gathered points).
//    A bucket with more than `cap` entries would serialise the whole launch on one lane (equal scalars - a plain sum
//    of points is an MSM with all scalars 1 - put every term of a window into one bucket), so such buckets are only
//    registered here: they are cut into chunks of `cap` entries, each chunk is summed by a whole workgroup
//    (heavy_chunk_kernel) and the chunk sums are folded per bucket (heavy_finish_kernel).  Uniform scalars have no
//    heavy buckets and the two extra kernels find an empty list.
struct HeavyBucket { u32 bucket, base, chunks; };
struct HeavyChunk { u32 bucket, index; };

// the 64 bytes of the point an entry names (16-byte loads)
struct RawPoint { uint4 v[4]; };
__device__ __forceinline__ RawPoint entry_point(const u32* points_xy, const u32* endo_xy, u32 e) {
  const uint4* src = (const uint4*)((((e >> 30) & 1u) ? endo_xy : points_xy) + (size_t)(e & ENTRY_INDEX_MASK) * 16);
  RawPoint r;
#pragma unroll
  for (int q = 0; q < 4; q++) r.v[q] = src[q];
  return r;
}
__device__ __forceinline__ void bucket_add_raw(XyzzK256& acc, const RawPoint& r, u32 e) {
  const u32 w[16] = {r.v[0].x, r.v[0].y, r.v[0].z, r.v[0].w, r.v[1].x, r.v[1].y, r.v[1].z, r.v[1].w,
                     r.v[2].x, r.v[2].y, r.v[2].z, r.v[2].w, r.v[3].x, r.v[3].y, r.v[3].z, r.v[3].w};
  u32 z = 0;
#pragma unroll
  for (int q = 0; q < 16; q++) z |= w[q];
  if (z == 0) return;                          // the identity (affine zeros) contributes nothing
  FeK256 x, y;
  k256::from_be_words(x, w);
  k256::from_be_words(y, w + 8);
  if (e >> 31) k256::neg(y, y);
  xyzz_add_mixed(acc, x, y);
}
__device__ __forceinline__ void bucket_accumulate(XyzzK256& acc, const u32* points_xy, const u32* endo_xy, u32 e) {
  bucket_add_raw(acc, entry_point(points_xy, endo_xy, e), e);
}
// acc += the points of entries s .. e, software-pipelined: the gather of entry q + 1 is in flight during the addition of entry q
__device__ __forceinline__ void bucket_accumulate_run(XyzzK256& acc, const u32* points_xy, const u32* endo_xy, const u32* sorted, u32 s, u32 e) {
  if (s >= e) return;
  u32 en = sorted[s];
  RawPoint pn = entry_point(points_xy, endo_xy, en);
#pragma unroll 1
  for (u32 q = s; q < e; q++) {
    const u32 ec = en;
    const RawPoint pc = pn;
    if (q + 1 < e) {
      en = sorted[q + 1];
      pn = entry_point(points_xy, endo_xy, en);
    }
    bucket_add_raw(acc, pc, ec);
  }
}
// One lane per (bucket, part): a bucket's run of entries is cut into `split` equal parts summed by `split` neighbouring
// lanes, and bucket_combine_kernel adds the parts.  With one lane per bucket the 294 912 buckets of a 2^23-term sum were
// 1.125 x the 262 144 lanes the chip holds at this kernel's occupancy - a second, almost empty round as long as the first
// (measured: 9.4 ms, 56 % of the rate the same additions reach when the chip stays full); eight parts per bucket make it
// nine full rounds of shorter tasks.
__global__ void __launch_bounds__(256, 4) bucket_sum_kernel(const u32* points_xy, const u32* endo_xy, const u32* offsets, const u32* sorted, JacK256* parts,
                                                            int nb, int split, u32 cap, u32* heavy_ctr, HeavyBucket* heavy, HeavyChunk* chunks) {
  ECGPU_GRID_STRIDE(t, (size_t)nb * split) {
    const size_t b = t / split;
    const u32 j = (u32)(t % split);
    XyzzK256 acc;
    xyzz_set_infinity(acc);
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
      JacK256 z;                                  // parts of a heavy bucket stay empty: heavy_finish_kernel adds its sum
      k256::set_zero(z.x); k256::set_zero(z.y); k256::set_zero(z.z);
      parts[t] = z;
      continue;
    }
    const u32 s = lo + (u32)(((u64)len * j) / split), e = lo + (u32)(((u64)len * (j + 1)) / split);
    bucket_accumulate_run(acc, points_xy, endo_xy, sorted, s, e);
    JacK256 r;
    xyzz_to_jacobian(r, acc);
    parts[t] = r;
  }
}
// buckets[b] = sum of its parts (general additions; a heavy bucket's sum is written by heavy_finish_kernel, which runs before)
__global__ void __launch_bounds__(256) bucket_combine_kernel(const JacK256* parts, JacK256* buckets, int nb, int split, const u32* offsets, u32 cap) {
  ECGPU_GRID_STRIDE(b, (size_t)nb) {
    if (offsets[b + 1] - offsets[b] > cap) continue;
    JacK256 acc = parts[b * split];
#pragma unroll 1
    for (int j = 1; j < split; j++) jac_add(acc, acc, parts[b * split + j]);
    buckets[b] = acc;
  }
}

// sum over the lanes of a workgroup through LDS (count = blockDim.x, a power of two)
__device__ __forceinline__ void lds_tree_sum(JacK256* sh, JacK256& v, int lane, int count) {
  sh[lane] = v;
  __syncthreads();
  for (int off = count >> 1; off >= 1; off >>= 1) {
    if (lane < off) {
      JacK256 a = sh[lane], b = sh[lane + off];
      jac_add(a, a, b);
      sh[lane] = a;
    }
    __syncthreads();
  }
  v = sh[0];
  __syncthreads();
}
// 4b. one workgroup per chunk of a heavy bucket: lanes stride over the chunk, LDS tree sum
__global__ void __launch_bounds__(256) heavy_chunk_kernel(const u32* points_xy, const u32* endo_xy, const u32* offsets, const u32* sorted, u32 cap,
                                                          const u32* heavy_ctr, const HeavyChunk* chunks, JacK256* partial) {
  __shared__ JacK256 sh[256];
  const u32 total = heavy_ctr[1];
  for (u32 c = blockIdx.x; c < total; c += gridDim.x) {
    const HeavyChunk ch = chunks[c];
    const u32 lo = offsets[ch.bucket] + ch.index * cap;
    const u32 end = offsets[ch.bucket + 1];
    const u32 hi = (end - lo > cap) ? lo + cap : end;
    XyzzK256 xacc;
    xyzz_set_infinity(xacc);
#pragma unroll 1
    for (u32 j = lo + threadIdx.x; j < hi; j += 256) bucket_accumulate(xacc, points_xy, endo_xy, sorted[j]);
    JacK256 acc;
    xyzz_to_jacobian(acc, xacc);
    lds_tree_sum(sh, acc, threadIdx.x, 256);
    if (threadIdx.x == 0) partial[c] = acc;
  }
}
// 4c. one workgroup per heavy bucket: fold its chunk sums
__global__ void __launch_bounds__(256) heavy_finish_kernel(const u32* heavy_ctr, const HeavyBucket* heavy, const JacK256* partial, JacK256* buckets) {
  __shared__ JacK256 sh[256];
  const u32 total = heavy_ctr[0];
  for (u32 h = blockIdx.x; h < total; h += gridDim.x) {
    const HeavyBucket hb = heavy[h];
    JacK256 acc;
    k256::set_zero(acc.x); k256::set_zero(acc.y); k256::set_zero(acc.z);
#pragma unroll 1
    for (u32 j = threadIdx.x; j < hb.chunks; j += 256) jac_add(acc, acc, partial[hb.base + j]);
    lds_tree_sum(sh, acc, threadIdx.x, 256);
    if (threadIdx.x == 0) buckets[hb.bucket] = acc;
  }
}

// 5. Weighted sums by a tree of running sums.  For a run of L points B_0..B_{L-1},
//      T = sum B_t   and   Wt = sum (t + 1) B_t
//    come from 2L additions (run += B_t from the top, wt += run).  Level 0 does this for every SEG0
//    consecutive buckets; level 1 for every SEG1 consecutive level-0 results; the window kernel
//    finishes.  With weights nested as j + 1 = (s1 * SEG1 + s0) * SEG0 + t + 1:
//      S = sum_j (j+1) B_j = sum Wt0 + SEG0 * [ sum_{s1} ( (Wt1 - T1) ) + SEG1 * sum_{s1} s1 * T1 ]
//    where T1/Wt1 are the level-1 sums over the level-0 totals T0 (Wt1 weights them 1..SEG1).
__global__ void __launch_bounds__(64) segment_kernel(const JacK256* in, JacK256* out_t, JacK256* out_w, int len, int total) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  const JacK256* B = in + (size_t)s * len;
  JacK256 run, wt;
  k256::set_zero(run.x); k256::set_zero(run.y); k256::set_zero(run.z);
  wt = run;
#pragma unroll 1
  for (int t = len - 1; t >= 0; t--) {
    jac_add(run, run, B[t]);
    jac_add(wt, wt, run);
  }
  out_t[s] = run;
  out_w[s] = wt;
}
// plain sums of `len` consecutive points (for the sum of the level-0 weighted parts)
__global__ void __launch_bounds__(64) sum_kernel(const JacK256* in, JacK256* out, int len, int total) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  JacK256 acc;
  k256::set_zero(acc.x); k256::set_zero(acc.y); k256::set_zero(acc.z);
#pragma unroll 1
  for (int t = 0; t < len; t++) jac_add(acc, acc, in[(size_t)s * len + t]);
  out[s] = acc;
}

// 6. per window: combine NSEG1 level-1 results
//      S_w = sumW0 + SEG0 * ( sum_{s1} (Wt1 - T1)  +  SEG1 * sum_{s1} s1 * T1 )
//    One workgroup per window, lane s1 owns one level-1 segment (and one of the NSUMW partial sums of the level-0
//    weighted parts); the three sums over the lanes are LDS tree reductions, so the dependent chain is
//    log2(NSEG1) additions instead of NSEG1 x 4.
__global__ void __launch_bounds__(NSEG1) window_kernel(const JacK256* t1, const JacK256* w1, const JacK256* sumw0, JacK256* win) {
  static_assert(NSUMW == NSEG1 && NSEG1 <= 1024, "one lane per level-1 segment and per partial sum");
  __shared__ JacK256 sh[NSEG1];
  const int w = blockIdx.x, s = threadIdx.x;          // blockDim.x == NSEG1
  JacK256 inner, acc, sw;
  k256::set_zero(acc.x); k256::set_zero(acc.y); k256::set_zero(acc.z);
  const JacK256 T = t1[w * NSEG1 + s];
  JacK256 neg = T;
  k256::neg(neg.y, neg.y);
  jac_add(inner, w1[w * NSEG1 + s], neg);            // Wt1 - T1
  // s * T by double-and-add over the bits of s
#pragma unroll 1
  for (int bit = LOG_NSEG1 - 1; bit >= 0; bit--) {
    k256::jac_double(acc);
    if ((s >> bit) & 1) jac_add(acc, acc, T);
  }
  sw = sumw0[w * NSUMW + s];
  lds_tree_sum(sh, inner, s, NSEG1);
  lds_tree_sum(sh, acc, s, NSEG1);
  lds_tree_sum(sh, sw, s, NSEG1);
  if (s == 0) {
#pragma unroll 1
    for (int j = 0; j < LOG_SEG1; j++) k256::jac_double(acc);
    jac_add(acc, acc, inner);
#pragma unroll 1
    for (int j = 0; j < LOG_SEG0; j++) k256::jac_double(acc);
    jac_add(acc, acc, sw);
    win[w] = acc;
  }
}

// Small sums (n below SMALL_MSM_TERMS): the bucket method has a fixed cost of ~2.2 ms (17 x 2^15 buckets to reduce, 256
// serial doublings), more than n plain scalar multiplications take, so those run through the variable-base kernel and
// the n products are summed here: one workgroup per slice (lanes stride, LDS tree), then one workgroup over the slice
// sums.  Measured (ms, this path / buckets): 2^10 1.1 / 2.2, 2^14 1.2 / 2.3, 2^16 1.3 / 2.5, 2^18 3.1 / 2.9.
constexpr size_t SMALL_MSM_TERMS = (size_t)3 << 16;
__global__ void __launch_bounds__(256) sum_affine_kernel(const u32* xy, size_t n, JacK256* partial) {
  __shared__ JacK256 sh[256];
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = (lo + per < n) ? lo + per : n;
  XyzzK256 xacc;
  xyzz_set_infinity(xacc);
#pragma unroll 1
  for (size_t j = lo + threadIdx.x; j < hi; j += 256) bucket_accumulate(xacc, xy, xy, (u32)j);      // plain sum: indices below 2^30, no flags
  JacK256 acc;
  xyzz_to_jacobian(acc, xacc);
  lds_tree_sum(sh, acc, threadIdx.x, 256);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(256) sum_partials_kernel(const JacK256* partial, int count, JacK256* win) {
  __shared__ JacK256 sh[256];
  JacK256 acc;
  k256::set_zero(acc.x); k256::set_zero(acc.y); k256::set_zero(acc.z);
#pragma unroll 1
  for (int j = threadIdx.x; j < count; j += 256) jac_add(acc, acc, partial[j]);
  lds_tree_sum(sh, acc, threadIdx.x, 256);
  if (threadIdx.x == 0) win[0] = acc;                  // finish_kernel with nwin = 1 converts and stores it
}

// 7. Horner over the windows, conversion to affine, output
__global__ void __launch_bounds__(64) finish_kernel(const JacK256* win, int nwin, u32* out, int out_fmt) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  JacK256 r = win[nwin - 1];
#pragma unroll 1
  for (int w = nwin - 2; w >= 0; w--) {
#pragma unroll 1
    for (int j = 0; j < C; j++) k256::jac_double(r);
    jac_add(r, r, win[w]);
  }
  const bool inf = k256::is_zero(r.z);
  FeK256 zi, zi2, zi3, x, y, one, zero;
  k256::set_one(one); k256::set_zero(zero);
  k256::inv(zi, r.z);
  k256::sqr(zi2, zi); k256::mul(zi3, zi2, zi);
  k256::mul(x, r.x, zi2); k256::mul(y, r.y, zi3);
  if (inf) { x = zero; y = (out_fmt == FMT_PROJECTIVE) ? one : zero; }
  CurveK256::fe_store(out, x);
  CurveK256::fe_store(out + 8, y);
  if (out_fmt == FMT_PROJECTIVE) CurveK256::fe_store(out + 16, inf ? zero : one);
}

// homogeneous projective input -> affine (one inversion per lane; only used when the caller hands X:Y:Z)
__global__ void __launch_bounds__(256) to_affine_kernel(const u32* xyz, u32* xy, size_t n) {
  ECGPU_GRID_STRIDE(i, n) {
    PtK256 p;
    load_point<CurveK256>(p, xyz + i * 24, FMT_PROJECTIVE);
    store_affine_from_projective<CurveK256>(xy + i * 16, nullptr, p);
  }
}

}  // namespace msm
}  // namespace ecgpu
