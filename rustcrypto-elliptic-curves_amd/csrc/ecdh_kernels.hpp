// Batched ECDH around the constant-time variable-base kernel: elliptic_curve::ecdh::diffie_hellman (external
// elliptic-curve 0.13.8, re-exported at k256/src/ecdh.rs:41, p256/src/ecdh.rs, p384/src/ecdh.rs) computes
// (public_key.to_projective() * secret).to_affine() and SharedSecret takes its x (k256 ecdh.rs:51-55).  The reference's
// types make invalid inputs unrepresentable - NonZeroScalar (0 < d < n), PublicKey (on the curve, canonical, not the
// identity) -; over the byte boundary they are checked per element here and reported in ok[].  Device code only.
#pragma once
#include "ecdsa_kernels.hpp"

namespace ecgpu {
namespace ecdh {

// ok[i] = NonZeroScalar::from_repr(secret[i]).is_some() && PublicKey::from_affine(pub[i]).is_ok()
// pubs_sane[i] = pub[i] where ok, else the generator: the constant-time kernel's exception-free argument (varbase_ct.hpp) is stated for
// points of the group, so only such points reach it - a key that is off the curve never enters the multiplication (its result is
// discarded by finish_kernel anyway).  The choice depends on the PUBLIC key only.
template <class C>
__global__ void __launch_bounds__(256) prep_kernel(const u32* secrets, const u32* pubs, u32* pubs_sane, uint8_t* ok, size_t n) {
  using O = OrderOf<C>;
  ECGPU_GRID_STRIDE(i, n) {
    u32 d[O::L];
    ecdsa::load_be<O::L>(d, secrets + i * C::NW);
    const bool key_ok = ecdsa::public_key_ok<C>(pubs + i * 2 * C::NW);
    ok[i] = (ecdsa::in_range<O>(d) && key_ok) ? 1 : 0;
    if (key_ok) {
#pragma unroll
      for (int w = 0; w < 2 * C::NW; w++) pubs_sane[i * 2 * C::NW + w] = pubs[i * 2 * C::NW + w];
    } else {
      typename C::Pt g;
      C::pt_generator(g);
      C::fe_store(pubs_sane + i * 2 * C::NW, g.x);
      C::fe_store(pubs_sane + i * 2 * C::NW + C::NW, g.y);
    }
  }
}
// shared[i] = x of the product, zeros where ok[i] = 0
template <class C>
__global__ void __launch_bounds__(256) finish_kernel(const u32* prod_xy, const uint8_t* ok, u32* shared_x, size_t n) {
  ECGPU_GRID_STRIDE(i, n) {
    const u32 keep = ok[i] ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int w = 0; w < C::NW; w++) shared_x[i * C::NW + w] = prod_xy[i * 2 * C::NW + w] & keep;
  }
}

}  // namespace ecdh
}  // namespace ecgpu
