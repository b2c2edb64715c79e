// GroupEncoding::{to_bytes, from_bytes} for batches: the fixed-width compressed SEC1 representation
// (1 + NB bytes: tag 0x02 / 0x03 and x; the identity is all zeros - "technically an invalid SEC1 encoding",
// k256/src/arithmetic/affine.rs:213-238, primeorder/src/affine.rs:256-278), and ToEncodedPoint / FromEncodedPoint with
// the uncompressed form 0x04 || x || y (k256 affine.rs:241-284, primeorder affine.rs:161-195, 340-358) in fixed-width
// records (the identity: tag 0x00 and zero padding).  Device code only.
#pragma once
#include "kernels.hpp"

namespace ecgpu {
namespace sec1 {

// affine x || y (zeros = identity) -> tag || x
template <class C>
__global__ void __launch_bounds__(256) to_bytes_kernel(const u32* xy, const uint8_t* inf, uint8_t* out, size_t n, int uncompressed) {
  constexpr int NB = C::NB;
  const int body = uncompressed ? 2 * NB : NB;
  ECGPU_GRID_STRIDE(i, n) {
    const uint8_t* src = (const uint8_t*)(xy + i * 2 * C::NW);
    uint8_t* o = out + i * (size_t)(body + 1);
    u32 z = 0;
#pragma unroll
    for (int j = 0; j < 2 * C::NW; j++) z |= xy[i * 2 * C::NW + j];
    const bool id = (z == 0) || (inf && inf[i]);
    o[0] = id ? 0 : (uncompressed ? (uint8_t)4 : (uint8_t)(2 + (src[2 * NB - 1] & 1)));
    for (int j = 0; j < body; j++) o[1 + j] = id ? 0 : src[j];
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
