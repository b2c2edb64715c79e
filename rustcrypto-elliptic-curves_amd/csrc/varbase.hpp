// Variable-base scalar multiplication k*P, throughput schedule for the curves without an efficient
// endomorphism (P-256, P-384).  The reference computes this with complete homogeneous formulas, an
// unsigned 4-bit window and a 16-entry table (primeorder/src/projective.rs:106-150: 4361 / 6473
// field multiplications); the group element is what is specified, so here:
//   * Jacobian coordinates (doubling 4M+4S with a halving for a = -3, general addition 11M+5S);
//   * signed 4-bit digits (k > n/2 is replaced by n - k and -P): table [P .. 8P] of 8 Jacobian points
//     per lane in a lane-contiguous global workspace, built with 4 doublings and 3 mixed additions;
//   * per-lane batched conversion to affine (one inversion per BATCH results).
// (The common-Z "effective affine" table used for k256 needs a = 0: on the isomorphic curve the
// a = -3 doubling shortcut no longer holds.)  Device code only.
#pragma once
#include "jacobian.hpp"
#include "kernels.hpp"

namespace ecgpu {
namespace vb {

template <class C> constexpr int nwin() { return 2 * C::NB + 1; }   // nibbles + the carry digit

template <class C, int BATCH, int WAVES>
__global__ void __launch_bounds__(256, WAVES) mul_kernel(const u32* scalars, const u32* points, int pt_fmt, u32* out, int out_fmt,
                                                         uint8_t* out_inf, size_t n, Jac<C>* tab_ws) {
  constexpr int NW = C::NW;
  using Fe = typename C::Fe;
  // table [P .. 8P] of this lane: 8 consecutive entries of a global workspace (one block per resident lane).  In
  // the private segment a lane-divergent index turns every entry read into scattered dword rows (measured on the
  // k256 kernel: 3.5x the fetch traffic).
  Jac<C>* tab = tab_ws + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  Jac<C> res[BATCH];
  Fe pre[BATCH];
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int pw = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * NW;
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
      // input point -> Jacobian (homogeneous X:Y:Z is Jacobian XZ : YZ^2 : Z)
      const u32* src = points + i * pw;
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
      if (flip) C::fe_neg(p.y, p.y);
      if (p_inf) jac::set_infinity<C>(p);
      // table [P, 2P, .., 8P]
      {
        Jac<C> p2 = p, t, u;
        jac::dbl<C>(p2);                       // 2P
        tab[0] = p; tab[1] = p2;
        jac::add<C>(t, p2, p);                 // 3P
        tab[2] = t;
        u = t; jac::dbl<C>(u);                 // 6P
        tab[5] = u;
        jac::add<C>(u, u, p);                  // 7P
        tab[6] = u;
        jac::dbl<C>(p2);                       // 4P
        tab[3] = p2;
        jac::add<C>(t, p2, p);                 // 5P
        tab[4] = t;
        jac::dbl<C>(p2);                       // 8P
        tab[7] = p2;
      }
      // signed nibbles: digit_j = nibble_j(k + 0x88..8) - 8, the carry out of the top nibble is the last digit
      u32 y[NW], c = 0;
#pragma unroll
      for (int w = 0; w < NW; w++) y[w] = addc(k[w], 0x88888888u, c);
      Jac<C> acc;
      jac::set_infinity<C>(acc);
      if (c) acc = tab[0];
#pragma unroll 1
      for (int j = 8 * NW - 1; j >= 0; j--) {
#pragma unroll 1
        for (int d = 0; d < 4; d++) jac::dbl<C>(acc);
        u32 word = y[0];
#pragma unroll
        for (int q = 1; q < NW; q++) word = (j >> 3) == q ? y[q] : word;
        const int sd = (int)((word >> (4 * (j & 7))) & 15u) - 8;
        if (sd != 0) {
          Jac<C> e = tab[(sd < 0 ? -sd : sd) - 1];
          if (sd < 0) C::fe_neg(e.y, e.y);
          jac::add<C>(acc, acc, e);
        }
      }
      res[b] = acc;
      cnt = b + 1;
    }
    // batched conversion to affine and output
    Fe accz; C::fe_one(accz);
#pragma unroll 1
    for (int b = 0; b < cnt; b++) {
      pre[b] = accz;
      Fe z = res[b].z;
      if (C::fe_is_zero(z)) C::fe_one(z);
      C::fe_mul(accz, accz, z);
    }
    Fe ai;
    C::fe_inv(ai, accz);
#pragma unroll 1
    for (int b = cnt - 1; b >= 0; b--) {
      const size_t i = base + (size_t)b * T;
      Fe z = res[b].z, one, zero, zi, t, x, yy;
      C::fe_one(one); C::fe_zero(zero);
      const bool zr = C::fe_is_zero(z);
      if (zr) z = one;
      C::fe_mul(zi, ai, pre[b]);
      C::fe_mul(ai, ai, z);
      C::fe_sqr(t, zi);
      C::fe_mul(x, res[b].x, t);
      C::fe_mul(t, t, zi);
      C::fe_mul(yy, res[b].y, t);
      if (zr) { x = zero; yy = zero; }
      if (out_fmt == FMT_PROJECTIVE) {
        if (zr) yy = one;
        u32* o = out + i * 3 * NW;
        C::fe_store(o, x); C::fe_store(o + NW, yy); C::fe_store(o + 2 * NW, zr ? zero : one);
      } else {
        u32* o = out + i * 2 * NW;
        C::fe_store(o, x); C::fe_store(o + NW, yy);
        if (out_inf) out_inf[i] = zr ? 1 : 0;
      }
    }
  }
}

}  // namespace vb
}  // namespace ecgpu
