// GroupEncoding::{to_bytes, from_bytes} for batches: the fixed-width compressed SEC1 representation
// (1 + NB bytes: tag 0x02 / 0x03 and x; the identity is all zeros - "technically an invalid SEC1 encoding",
// k256/src/arithmetic/affine.rs:213-238, primeorder/src/affine.rs:256-278), and ToEncodedPoint / FromEncodedPoint with
// the uncompressed form 0x04 || x || y (k256 affine.rs:241-284, primeorder affine.rs:161-195, 340-358) in fixed-width
// records (the identity: tag 0x00 and zero padding).  Device code only.
#pragma once
#include "kernels.hpp"

namespace ecgpu {
namespace sec1 {

// affine x || y (zeros = identity) -> tag || x [|| y].  Records are 33 / 65 / 49 / 97 bytes, so a lane's own stores would be
// byte stores at odd addresses (1.1x10^9 points/s for the 65-byte form); instead a workgroup lays its 256 records out in LDS
// and copies the tile to HBM as consecutive dwords (a tile starts at a multiple of 256 records, hence 4-byte aligned).
template <class C>
__global__ void __launch_bounds__(256) to_bytes_kernel(const u32* xy, const uint8_t* inf, uint8_t* out, size_t n, int uncompressed) {
  constexpr int NB = C::NB, NW = C::NW;
  const int body = uncompressed ? 2 * NB : NB, rec = body + 1;
  __shared__ __align__(16) uint8_t tile[256 * (2 * NB + 1)];
  for (size_t t0 = (size_t)blockIdx.x * 256; t0 < n; t0 += (size_t)gridDim.x * 256) {
    const size_t i = t0 + threadIdx.x;
    if (i < n) {
      u32 w[2 * NW], z = 0;
#pragma unroll
      for (int j = 0; j < 2 * NW; j++) { w[j] = xy[i * 2 * NW + j]; z |= w[j]; }
      const bool id = (z == 0) || (inf && inf[i]);
      uint8_t* o = tile + (size_t)threadIdx.x * rec;
      // big-endian bytes in memory order: byte b of the record body is byte (b & 3) of word b >> 2; y's last byte decides the tag
      o[0] = id ? 0 : (uncompressed ? (uint8_t)4 : (uint8_t)(2 + ((w[2 * NW - 1] >> 24) & 1)));
#pragma unroll
      for (int j = 0; j < 2 * NW; j++) {
        if (4 * j < body) {
          const u32 v = id ? 0u : w[j];
          o[1 + 4 * j] = (uint8_t)v; o[2 + 4 * j] = (uint8_t)(v >> 8); o[3 + 4 * j] = (uint8_t)(v >> 16); o[4 + 4 * j] = (uint8_t)(v >> 24);
        }
      }
    }
    __syncthreads();
    const size_t cnt = (n - t0 < 256) ? n - t0 : 256, bytes = cnt * rec;
    uint8_t* dst = out + t0 * rec;
    if ((((uintptr_t)dst) & 3) == 0) {
      for (size_t q = threadIdx.x; q < bytes / 4; q += 256) ((u32*)dst)[q] = ((const u32*)tile)[q];
      for (size_t b = (bytes & ~(size_t)3) + threadIdx.x; b < bytes; b += 256) dst[b] = tile[b];
    } else {
      for (size_t b = threadIdx.x; b < bytes; b += 256) dst[b] = tile[b];
    }
    __syncthreads();
  }
}

// tag || x -> affine x || y, ok.  Accepted: 0x02 / 0x03 (compressed), 0x05 (compact: the even root, what
// EncodedPoint::from_bytes makes of a 1 + NB byte string with that tag), all zeros (identity).
// record_bytes = 1 + NB: the compressed forms only; 1 + 2 NB: also 0x04 || x || y (coordinates below p and on the curve),
// and the short forms must be followed by NB zero bytes.
template <class C>
__global__ void __launch_bounds__(256) from_bytes_kernel(const uint8_t* in, u32* out_xy, uint8_t* ok, size_t n, int record_bytes) {
  constexpr int NB = C::NB, NW = C::NW;
  const bool wide = record_bytes == 1 + 2 * NB;
  ECGPU_GRID_STRIDE(i, n) {
    const uint8_t* s = in + i * (size_t)record_bytes;
    const uint8_t tag = s[0];
    u32 raw[NW], any = 0, tail_any = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) {
      raw[j] = (u32)s[1 + 4 * j] | ((u32)s[2 + 4 * j] << 8) | ((u32)s[3 + 4 * j] << 16) | ((u32)s[4 + 4 * j] << 24);
      any |= raw[j];
    }
    u32 rawy[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) {
      rawy[j] = wide ? ((u32)s[1 + NB + 4 * j] | ((u32)s[2 + NB + 4 * j] << 8) | ((u32)s[3 + NB + 4 * j] << 16) | ((u32)s[4 + NB + 4 * j] << 24)) : 0u;
      tail_any |= rawy[j];
    }
    u32* o = out_xy + i * 2 * NW;
    const bool tagged = (tag == 2 || tag == 3 || tag == 5) && tail_any == 0;
    bool good = false;
    typename C::Fe x, y, zero;
    C::fe_zero(zero);
    x = zero; y = zero;
    if (wide && tag == 4) {
      u32 lx[NW], ly[NW], p[NW];
      words_load_be<NW>(lx, raw);
      words_load_be<NW>(ly, rawy);
      C::modulus(p);
      const bool canon = !mp_geq<NW>(lx, p) && !mp_geq<NW>(ly, p);
      typename C::Fe l, r, d;
      C::fe_load(x, raw);
      C::fe_load(y, rawy);
      C::fe_sqr(l, y);
      C::curve_rhs(r, x);
      C::fe_sub(d, l, r);
      good = canon && C::fe_is_zero(d);
      if (!good) { x = zero; y = zero; }
    } else if (tagged) {
      u32 lx[NW], p[NW];
      words_load_be<NW>(lx, raw);
      C::modulus(p);
      const bool canon = !mp_geq<NW>(lx, p);
      typename C::Fe rhs, ny;
      C::fe_load(x, raw);
      C::curve_rhs(rhs, x);
      const bool has = C::fe_sqrt(y, rhs);
      C::fe_neg(ny, y);
      const bool odd = C::fe_is_odd(y);
      bool keep = (odd == (tag == 3));
      if (tag == 5 && !C::A_IS_ZERO) {
        // primeorder's decompact takes the numerically smaller of y and -y (primeorder/src/affine.rs:66-77,157-159);
        // k256's takes the even root (k256/src/arithmetic/affine.rs:207-211)
        u32 yb[NW], nb_[NW], yl[NW], nl[NW];
        C::fe_store(yb, y); C::fe_store(nb_, ny);
        words_load_be<NW>(yl, yb); words_load_be<NW>(nl, nb_);
        keep = mp_geq<NW>(nl, yl);
      }
      C::fe_select(y, keep, y, ny);
      good = canon && has;
      if (!good) { x = zero; y = zero; }
    } else if (tag == 0 && any == 0 && tail_any == 0) {
      good = true;                                  // the fixed-width identity
    }
    C::fe_store(o, x);
    C::fe_store(o + NW, y);
    ok[i] = good ? 1 : 0;
  }
}

}  // namespace sec1
}  // namespace ecgpu
