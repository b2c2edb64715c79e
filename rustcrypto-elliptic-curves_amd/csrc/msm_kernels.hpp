// Kernels of the k256 Pippenger MSM (see msm_k256.hpp for the schedule).  Device code only.
#pragma once
#include "kernels.hpp"
#include "msm_k256.hpp"

namespace ecgpu {
namespace msm {

// Signed 16-bit digits of one term (registers only).  k > n/2 is replaced by n - k with the opposite sign
// (then k < 2^255 and the carry window of the recoding is almost always empty; without this half of all
// terms land in its single bucket and one lane sums them).  d[w] in [-2^15, 2^15), d[16] in {0, 1};
// returns the sign flip.  A term whose point is the identity gets all-zero digits.
__device__ __forceinline__ bool term_digits(int* d, const u32* scalars, const u32* points_xy, size_t i) {
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
  u32 z = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) z |= points_xy[i * 16 + w];
  const bool skip = (z == 0);
  u32 carry = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    const u32 v = ((k[w >> 1] >> (16 * (w & 1))) & 0xFFFFu) + carry;
    carry = (v >= 0x8000u) ? 1u : 0u;      // v in [2^15, 2^16] becomes v - 2^16 with a carry
    d[w] = skip ? 0 : (int)v - (int)(carry << 16);
  }
  d[16] = skip ? 0 : (int)carry;
  return flip;
}

// 1. bucket histogram (one lane per term, 17 atomics)
__global__ void __launch_bounds__(256) hist_kernel(const u32* scalars, const u32* points_xy, size_t n, u32* hist) {
  ECGPU_GRID_STRIDE(i, n) {
    int d[NWIN];
    (void)term_digits(d, scalars, points_xy, i);
#pragma unroll
    for (int w = 0; w < NWIN; w++)
      if (d[w] != 0) atomicAdd(&hist[w * NBUCKET + (d[w] < 0 ? -d[w] : d[w]) - 1], 1u);
  }
}

// 2. exclusive scan of the histogram (one workgroup; 17 * 2^15 counters)
__global__ void __launch_bounds__(1024) scan_kernel(const u32* hist, u32* offsets, u32* cursor, int total) {
  __shared__ u32 part[1024];
  const int t = threadIdx.x;
  const int per = (total + 1023) / 1024;
  const int lo = t * per, hi = (lo + per < total) ? lo + per : total;
  u32 s = 0;
  for (int j = lo; j < hi; j++) s += hist[j];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    u32 v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  u32 run = (t == 0) ? 0 : part[t - 1];
  for (int j = lo; j < hi; j++) { offsets[j] = run; cursor[j] = run; run += hist[j]; }
  if (t == 1023) offsets[total] = part[1023];
}

// 3. scatter the (term, sign) pairs into their buckets (digits recomputed: cheaper than storing and
//    re-reading 34 bytes per term)
__global__ void __launch_bounds__(256) scatter_kernel(const u32* scalars, const u32* points_xy, size_t n, u32* cursor, u32* sorted) {
  ECGPU_GRID_STRIDE(i, n) {
    int d[NWIN];
    const bool flip = term_digits(d, scalars, points_xy, i);
    // all 17 returning atomics are issued before any result is used, so their latencies overlap
    u32 pos[NWIN];
#pragma unroll
    for (int w = 0; w < NWIN; w++) {
      const int a = d[w] < 0 ? -d[w] : d[w];
      pos[w] = a ? atomicAdd(&cursor[w * NBUCKET + a - 1], 1u) : 0u;
    }
#pragma unroll
    for (int w = 0; w < NWIN; w++)
      if (d[w] != 0) sorted[pos[w]] = (u32)i | (((d[w] < 0) != flip) ? 0x80000000u : 0u);
  }
}

// 4. one lane per bucket: sum of its points (Jacobian accumulator, mixed additions, gathered points)
__global__ void __launch_bounds__(256, 4) bucket_sum_kernel(const u32* points_xy, const u32* offsets, const u32* sorted, JacK256* buckets, int nb) {
  ECGPU_GRID_STRIDE(b, (size_t)nb) {
    JacK256 acc;
    k256::set_zero(acc.x); k256::set_zero(acc.y); k256::set_zero(acc.z);
    const u32 lo = offsets[b], hi = offsets[b + 1];
#pragma unroll 1
    for (u32 j = lo; j < hi; j++) {
      const u32 e = sorted[j];
      const u32* src = points_xy + (size_t)(e & 0x7FFFFFFFu) * 16;
      FeK256 x, y;
      k256::from_be_words(x, src);
      k256::from_be_words(y, src + 8);
      if (e >> 31) k256::neg(y, y);
      k256::jac_add_mixed(acc, x, y, nullptr);
    }
    buckets[b] = acc;
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
__global__ void __launch_bounds__(64) window_kernel(const JacK256* t1, const JacK256* w1, const JacK256* sumw0, JacK256* win) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= NWIN) return;
  JacK256 run, acc, inner, neg;
  k256::set_zero(run.x); k256::set_zero(run.y); k256::set_zero(run.z);
  acc = run; inner = run;
#pragma unroll 1
  for (int s = NSEG1 - 1; s >= 0; s--) {
    jac_add(inner, inner, w1[w * NSEG1 + s]);
    neg = t1[w * NSEG1 + s];
    k256::neg(neg.y, neg.y);
    jac_add(inner, inner, neg);                    // Wt1 - T1
    if (s >= 1) {
      jac_add(run, run, t1[w * NSEG1 + s]);
      jac_add(acc, acc, run);                      // sum s1 * T1
    }
  }
#pragma unroll 1
  for (int j = 0; j < LOG_SEG1; j++) k256::jac_double(acc);
  jac_add(acc, acc, inner);
#pragma unroll 1
  for (int j = 0; j < LOG_SEG0; j++) k256::jac_double(acc);
  JacK256 sw;
  k256::set_zero(sw.x); k256::set_zero(sw.y); k256::set_zero(sw.z);
#pragma unroll 1
  for (int g = 0; g < NSUMW; g++) jac_add(sw, sw, sumw0[w * NSUMW + g]);
  jac_add(acc, acc, sw);
  win[w] = acc;
}

// 7. Horner over the windows, conversion to affine, output
__global__ void finish_kernel(const JacK256* win, u32* out, int out_fmt) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  JacK256 r = win[NWIN - 1];
#pragma unroll 1
  for (int w = NWIN - 2; w >= 0; w--) {
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
