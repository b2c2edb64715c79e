// Kernel of ecgpu_map_to_curve_batch (the map itself: h2c_map.hpp).  Device code only.
#pragma once
#include "kernels.hpp"
#include "h2c_map.hpp"

namespace ecgpu {
namespace h2c {

// u: n x count field elements (canonical big-endian, < p; count = 1 or 2).  out = map(u_0) [+ map(u_1)], affine: one
// inversion per output.
template <class C>
__global__ void __launch_bounds__(256) map_kernel(const u32* u, int count, u32* out_xy, uint8_t* out_inf, size_t n) {
  using Fe = typename C::Fe;
  constexpr int NW = C::NW;
  ECGPU_GRID_STRIDE(i, n) {
    Fe uu;
    typename C::Pt p;
    C::fe_load(uu, u + i * count * NW);
    map_to_curve<C>(p, uu);
    if (count == 2) {
      typename C::Pt q, r;
      C::fe_load(uu, u + (i * count + 1) * NW);
      map_to_curve<C>(q, uu);
      C::pt_add(r, p, q);
      p = r;
    }
    store_affine_from_projective<C>(out_xy + i * 2 * NW, out_inf ? out_inf + i : nullptr, p);
  }
}

}  // namespace h2c
}  // namespace ecgpu
