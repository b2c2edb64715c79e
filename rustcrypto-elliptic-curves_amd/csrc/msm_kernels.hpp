// Kernels of the k256 Pippenger MSM (see msm_k256.hpp for the schedule).  Device code only.
#pragma once
#include "kernels.hpp"
#include "msm_k256.hpp"

namespace ecgpu {
namespace msm {

// Signed 16-bit digits of one term (registers only).  k > n/2 is replaced by n - k with the opposite sign
// (then k < 2^255 and the carry window of the recoding is almost always empty; without this half of all
// terms land in its single bucket and one lane sums them).  d[w] in [-2^15, 2^15), d[16] in {0, 1};
// returns the sign flip.  Terms whose point is the identity are not filtered here: the bucket sums skip them.
__device__ __forceinline__ bool term_digits(int* d, const u32* scalars, size_t i) {
  u32 k[8];
  words_load_be<8>(k, scalars + i * 8);
  k256::scalar_reduce_once(k);
  bool flip;
  {
    u32 nn[8], t[8];
    k256::order(nn);
    mp_sub<8>(t, nn, k);                 // n - k
    flip = !mp_geq<8>(t, k);             // n - k < k
#pragma unroll
    for (int w = 0; w < 8; w++) k[w] = flip ? t[w] : k[w];
  }
  u32 carry = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    const u32 v = ((k[w >> 1] >> (16 * (w & 1))) & 0xFFFFu) + carry;
    carry = (v >= 0x8000u) ? 1u : 0u;      // v in [2^15, 2^16] becomes v - 2^16 with a carry
    d[w] = (int)v - (int)(carry << 16);
  }
  d[16] = (int)carry;
  return flip;
}
// digit `w` (uniform over the workgroup) of term i and the sign of its bucket entry
__device__ __forceinline__ int term_digit(const u32* scalars, size_t i, int w, bool& negative) {
  int d[NWIN];
  const bool flip = term_digits(d, scalars, i);
  int v = d[0];
#pragma unroll
  for (int q = 1; q < NWIN; q++) v = (w == q) ? d[q] : v;
  negative = (v < 0) != flip;
  return v < 0 ? -v : v;
}

// Counting sort of the (term, window) pairs by bucket, privatised in LDS.  A workgroup owns one window and one
// contiguous chunk of the terms (grid = NWIN x SORT_CHUNKS, one workgroup per CU: the 2^15 counters of a window
// are 128 KB of its LDS), so every count and every cursor increment is an LDS atomic; global memory sees only
// the per-workgroup histograms (coalesced) and the scattered 4-byte index writes.  (The first version issued
// 2 x 17 global atomics per term: 17.5 of the 32 ms of a 2^23-term MSM.)
__device__ __forceinline__ void chunk_range(size_t n, int g, size_t& lo, size_t& hi) {
  lo = n * (size_t)g / SORT_CHUNKS;
  hi = n * (size_t)(g + 1) / SORT_CHUNKS;
}
// 1. part[w][g][b] = number of terms of chunk g whose window-w digit has magnitude b + 1
__global__ void __launch_bounds__(1024) hist_kernel(const u32* scalars, size_t n, u32* part) {
  __shared__ u32 cnt[NBUCKET];
  const int w = blockIdx.x / SORT_CHUNKS, g = blockIdx.x % SORT_CHUNKS;
  for (int b = threadIdx.x; b < NBUCKET; b += 1024) cnt[b] = 0;
  __syncthreads();
  size_t lo, hi;
  chunk_range(n, g, lo, hi);
  for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
    bool neg;
    const int a = term_digit(scalars, i, w, neg);
    if (a) atomicAdd(&cnt[a - 1], 1u);
  }
  __syncthreads();
  u32* dst = part + (size_t)blockIdx.x * NBUCKET;
  for (int b = threadIdx.x; b < NBUCKET; b += 1024) dst[b] = cnt[b];
}
// 2a. bucket totals: hist[w * NBUCKET + b] = sum over the chunks
__global__ void __launch_bounds__(256) totals_kernel(const u32* part, u32* hist) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= NWIN * NBUCKET) return;
  const int w = j / NBUCKET, b = j % NBUCKET;
  u32 s = 0;
#pragma unroll 1
  for (int g = 0; g < SORT_CHUNKS; g++) s += part[((size_t)(w * SORT_CHUNKS + g)) * NBUCKET + b];
  hist[j] = s;
}

// 2b. exclusive scan of the bucket totals: one workgroup per window scans its 2^15 counters (the window's own offset is
//     added by scan_base_kernel), so the 17 windows run side by side instead of one workgroup walking all 557 056.
__global__ void __launch_bounds__(1024) scan_kernel(const u32* hist, u32* offsets, u32* win_total) {
  __shared__ u32 psum[1024];
  const int w = blockIdx.x, t = threadIdx.x;
  constexpr int PER = NBUCKET / 1024;
  const u32* src = hist + (size_t)w * NBUCKET + t * PER;
  u32 s = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) s += src[j];
  psum[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    u32 v = (t >= off) ? psum[t - off] : 0;
    __syncthreads();
    psum[t] += v;
    __syncthreads();
  }
  u32 run = (t == 0) ? 0 : psum[t - 1];
  u32* dst = offsets + (size_t)w * NBUCKET + t * PER;
#pragma unroll
  for (int j = 0; j < PER; j++) { dst[j] = run; run += src[j]; }
  if (t == 1023) win_total[w] = psum[1023];
}
__global__ void __launch_bounds__(256) scan_base_kernel(u32* offsets, const u32* win_total) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > NWIN * NBUCKET) return;
  const int w = (j == NWIN * NBUCKET) ? NWIN : j / NBUCKET;
  u32 base = 0;
#pragma unroll 1
  for (int v = 0; v < w; v++) base += win_total[v];
  if (j == NWIN * NBUCKET) offsets[j] = base;          // one past the end: the number of sorted entries
  else offsets[j] += base;
}
// 2c. part[w][g][b] becomes the first output slot of chunk g inside bucket (w, b)
__global__ void __launch_bounds__(256) cursors_kernel(u32* part, const u32* offsets) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= NWIN * NBUCKET) return;
  const int w = j / NBUCKET, b = j % NBUCKET;
  u32 run = offsets[j];
#pragma unroll 1
  for (int g = 0; g < SORT_CHUNKS; g++) {
    u32* p = part + ((size_t)(w * SORT_CHUNKS + g)) * NBUCKET + b;
    const u32 c = *p;
    *p = run;
    run += c;
  }
}
// 3. scatter the (term, sign) pairs into their buckets; same workgroup -> (window, chunk) map as the histogram
__global__ void __launch_bounds__(1024) scatter_kernel(const u32* scalars, size_t n, const u32* part, u32* sorted) {
  __shared__ u32 cur[NBUCKET];
  const int w = blockIdx.x / SORT_CHUNKS, g = blockIdx.x % SORT_CHUNKS;
  const u32* src = part + (size_t)blockIdx.x * NBUCKET;
  for (int b = threadIdx.x; b < NBUCKET; b += 1024) cur[b] = src[b];
  __syncthreads();
  size_t lo, hi;
  chunk_range(n, g, lo, hi);
  for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
    bool neg;
    const int a = term_digit(scalars, i, w, neg);
    if (a) {
      const u32 pos = atomicAdd(&cur[a - 1], 1u);
      sorted[pos] = (u32)i | (neg ? 0x80000000u : 0u);
    }
  }
}

// 4. bucket sums.  One lane per bucket sums its points (Jacobian accumulator, mixed additions, gathered points).
//    A bucket with more than `cap` entries would serialise the whole launch on one lane (equal scalars - a plain sum
//    of points is an MSM with all scalars 1 - put every term of a window into one bucket), so such buckets are only
//    registered here: they are cut into chunks of `cap` entries, each chunk is summed by a whole workgroup
//    (heavy_chunk_kernel) and the chunk sums are folded per bucket (heavy_finish_kernel).  Uniform scalars have no
//    heavy buckets and the two extra kernels find an empty list.
struct HeavyBucket { u32 bucket, base, chunks; };
struct HeavyChunk { u32 bucket, index; };

__device__ __forceinline__ void bucket_accumulate(JacK256& acc, const u32* points_xy, u32 e) {
  const u32* src = points_xy + (size_t)(e & 0x7FFFFFFFu) * 16;
  u32 z = 0;
#pragma unroll
  for (int q = 0; q < 16; q++) z |= src[q];
  if (z == 0) return;                          // the identity (affine zeros) contributes nothing
  FeK256 x, y;
  k256::from_be_words(x, src);
  k256::from_be_words(y, src + 8);
  if (e >> 31) k256::neg(y, y);
  k256::jac_add_mixed(acc, x, y, nullptr);
}
__global__ void __launch_bounds__(256, 4) bucket_sum_kernel(const u32* points_xy, const u32* offsets, const u32* sorted, JacK256* buckets, int nb,
                                                            u32 cap, u32* heavy_ctr, HeavyBucket* heavy, HeavyChunk* chunks) {
  ECGPU_GRID_STRIDE(b, (size_t)nb) {
    JacK256 acc;
    k256::set_zero(acc.x); k256::set_zero(acc.y); k256::set_zero(acc.z);
    const u32 lo = offsets[b], hi = offsets[b + 1];
    if (hi - lo > cap) {
      const u32 k = (hi - lo + cap - 1) / cap;
      const u32 idx = atomicAdd(&heavy_ctr[0], 1u);
      const u32 base = atomicAdd(&heavy_ctr[1], k);
      heavy[idx] = HeavyBucket{(u32)b, base, k};
#pragma unroll 1
      for (u32 j = 0; j < k; j++) chunks[base + j] = HeavyChunk{(u32)b, j};
      continue;                                 // buckets[b] is written by heavy_finish_kernel
    }
#pragma unroll 1
    for (u32 j = lo; j < hi; j++) bucket_accumulate(acc, points_xy, sorted[j]);
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
__global__ void __launch_bounds__(256) heavy_chunk_kernel(const u32* points_xy, const u32* offsets, const u32* sorted, u32 cap, const u32* heavy_ctr,
                                                          const HeavyChunk* chunks, JacK256* partial) {
  __shared__ JacK256 sh[256];
  const u32 total = heavy_ctr[1];
  for (u32 c = blockIdx.x; c < total; c += gridDim.x) {
    const HeavyChunk ch = chunks[c];
    const u32 lo = offsets[ch.bucket] + ch.index * cap;
    const u32 end = offsets[ch.bucket + 1];
    const u32 hi = (end - lo > cap) ? lo + cap : end;
    JacK256 acc;
    k256::set_zero(acc.x); k256::set_zero(acc.y); k256::set_zero(acc.z);
#pragma unroll 1
    for (u32 j = lo + threadIdx.x; j < hi; j += 256) bucket_accumulate(acc, points_xy, sorted[j]);
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
  JacK256 acc;
  k256::set_zero(acc.x); k256::set_zero(acc.y); k256::set_zero(acc.z);
#pragma unroll 1
  for (size_t j = lo + threadIdx.x; j < hi; j += 256) bucket_accumulate(acc, xy, (u32)j);
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
__global__ void finish_kernel(const JacK256* win, int nwin, u32* out, int out_fmt) {
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
