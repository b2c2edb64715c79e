// Kernels of the k256 Pippenger MSM (see msm_k256.hpp for the schedule).  Device code only.
#pragma once
#include "kernels.hpp"
#include "msm_k256.hpp"

namespace ecgpu {
namespace msm {

// 1. signed 16-bit digits + bucket histogram.  digits is window-major ([w][i]) so that the scatter
//    pass reads it coalesced.  A term whose point is the identity contributes nothing.
__global__ void __launch_bounds__(256) digits_hist_kernel(const u32* scalars, const u32* points_xy, size_t n, int16_t* digits, uint8_t* flips,
                                                          u32* hist) {
  ECGPU_GRID_STRIDE(i, n) {
    u32 k[8];
    words_load_be<8>(k, scalars + i * 8);
    k256::scalar_reduce_once(k);
    // k > n/2: use (n - k, -P).  Then k < 2^255 and the carry window of the signed recoding is (almost
    // always) empty; without this half of all terms land in its single bucket and one lane sums them.
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
      u32 d = ((k[w >> 1] >> (16 * (w & 1))) & 0xFFFFu) + carry;
      carry = (d >= 0x8000u) ? 1u : 0u;      // d in [2^15, 2^16] becomes d - 2^16 with a carry
      int sd = (int)d - (int)(carry << 16);   // in [-2^15, 2^15): fits int16; the flip is applied by the scatter pass
      if (skip) sd = 0;
      digits[(size_t)w * n + i] = (int16_t)sd;
      if (sd != 0) atomicAdd(&hist[w * NBUCKET + (sd < 0 ? -sd : sd) - 1], 1u);
    }
    int top = skip ? 0 : (int)carry;
    flips[i] = flip ? 1 : 0;
    digits[(size_t)16 * n + i] = (int16_t)top;
    if (top) atomicAdd(&hist[16 * NBUCKET + 0], 1u);
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

// 3. scatter the (term, sign) pairs into their buckets
__global__ void __launch_bounds__(256) scatter_kernel(const int16_t* digits, const uint8_t* flips, size_t n, u32* cursor, u32* sorted) {
  const size_t total = (size_t)NWIN * n;
  ECGPU_GRID_STRIDE(e, total) {
    const int w = (int)(e / n);
    const size_t i = e - (size_t)w * n;
    const int sd = digits[e];
    if (sd != 0) {
      const u32 pos = atomicAdd(&cursor[w * NBUCKET + (sd < 0 ? -sd : sd) - 1], 1u);
      const bool negate = (sd < 0) != (flips[i] != 0);
      sorted[pos] = (u32)i | (negate ? 0x80000000u : 0u);
    }
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

// 5. per (window, segment): T = sum_t B_t and Wt = sum_t t * B_t over SEG consecutive buckets
__global__ void __launch_bounds__(64) segment_kernel(const JacK256* buckets, JacK256* seg_t, JacK256* seg_w) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;       // window * NSEG + segment
  if (s >= NWIN * NSEG) return;
  const JacK256* B = buckets + (size_t)s * SEG;
  JacK256 run, wt;
  k256::set_zero(run.x); k256::set_zero(run.y); k256::set_zero(run.z);
  wt = run;
#pragma unroll 1
  for (int t = SEG - 1; t >= 0; t--) {
    jac_add(run, run, B[t]);
    jac_add(wt, wt, run);
  }
  seg_t[s] = run;
  seg_w[s] = wt;
}

// 6. per window: S_w = sum_j j * B_j = SEG * sum_s s * T_s + sum_s Wt_s
__global__ void __launch_bounds__(64) window_kernel(const JacK256* seg_t, const JacK256* seg_w, JacK256* win) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= NWIN) return;
  JacK256 run, acc, sw;
  k256::set_zero(run.x); k256::set_zero(run.y); k256::set_zero(run.z);
  acc = run; sw = run;
#pragma unroll 1
  for (int s = NSEG - 1; s >= 0; s--) {
    jac_add(sw, sw, seg_w[w * NSEG + s]);
    if (s >= 1) {
      jac_add(run, run, seg_t[w * NSEG + s]);
      jac_add(acc, acc, run);
    }
  }
#pragma unroll 1
  for (int j = 0; j < 8; j++) k256::jac_double(acc);          // * SEG = 2^8
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
