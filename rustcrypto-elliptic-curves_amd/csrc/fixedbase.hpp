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
#include "kernels.hpp"

namespace ecgpu {
namespace fb {

constexpr int W = 8;
constexpr int ENTRIES = 1 << (W - 1);                                   // |digit| in 1..128
template <class C> constexpr int nwin() { return C::NB + 1; }            // one byte per window + the carry window

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

template <class C, int BATCH, int WAVES>
__global__ void __launch_bounds__(256, WAVES) mul_kernel(const u32* scalars, const AffEntry<C>* table, u32* out, int out_fmt,
                                                         uint8_t* out_inf, size_t n) {
  constexpr int NW = C::NW;
  Jac<C> res[BATCH];
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
      Jac<C> acc;
      jac::set_infinity<C>(acc);
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
          jac::add_mixed<C>(acc, x, y);
        }
      }
      res[b] = acc;
      cnt = b + 1;
    }
    // batched conversion to affine and output
    typename C::Fe acc; C::fe_one(acc);
#pragma unroll 1
    for (int b = 0; b < cnt; b++) {
      pre[b] = acc;
      typename C::Fe z = res[b].z;
      if (C::fe_is_zero(z)) C::fe_one(z);
      C::fe_mul(acc, acc, z);
    }
    typename C::Fe ai;
    C::fe_inv(ai, acc);
#pragma unroll 1
    for (int b = cnt - 1; b >= 0; b--) {
      const size_t i = base + (size_t)b * T;
      typename C::Fe z = res[b].z, one, zero, zi, t, x, y;
      C::fe_one(one); C::fe_zero(zero);
      const bool zr = C::fe_is_zero(z);
      if (zr) z = one;
      C::fe_mul(zi, ai, pre[b]);
      C::fe_mul(ai, ai, z);
      C::fe_sqr(t, zi);
      C::fe_mul(x, res[b].x, t);
      C::fe_mul(t, t, zi);
      C::fe_mul(y, res[b].y, t);
      if (zr) { x = zero; y = zero; }
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
}

}  // namespace fb
}  // namespace ecgpu
