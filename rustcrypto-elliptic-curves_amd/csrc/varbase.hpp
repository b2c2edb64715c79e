// Variable-base scalar multiplication kernel for P-256 / P-384: grid-stride passes over the per-lane body of
// varbase_lane.hpp (schedule and workspace described there).  Device code only.
#pragma once
#include "varbase_lane.hpp"
#include "varbase_ct.hpp"
#include "kernels.hpp"

namespace ecgpu {
namespace vb {

template <class C, int BATCH, int WAVES, int NT = 1, int WB = 4>
__global__ void __launch_bounds__(256, WAVES) mul_kernel(const u32* scalars, const u32* points, int pt_fmt, u32* out, int out_fmt,
                                                         uint8_t* out_inf, size_t n, LaneWs<C, BATCH, WB>* ws_all) {
  LaneWs<C, BATCH, WB>& ws = ws_all[(size_t)blockIdx.x * blockDim.x + threadIdx.x];
  __shared__ u32 lds_digits[NT * digit_words<C, WB>()][256];
  const DigitMem dm{&lds_digits[0][threadIdx.x], 256};
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * (BATCH / NT)) lane_pass<C, BATCH, NT, WB>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, base, T, ws, dm);
}

}  // namespace vb

namespace vbct {

// Constant-time variable base for secret scalars (varbase_ct.hpp): grid-stride passes over the per-lane body; the
// workspace is one lane-interleaved region of lane_chunks<C, BATCH>() 16-byte chunks x 256 lanes per workgroup.
template <class C, int BATCH, int WAVES>
__global__ void __launch_bounds__(256, WAVES) mul_kernel(const u32* scalars, const u32* points, int pt_fmt, u32* out, int out_fmt,
                                                         uint8_t* out_inf, size_t n, Chunk* ws_all) {
  const LaneMem ws{ws_all + (size_t)blockIdx.x * lane_chunks<C, BATCH>() * 256 + threadIdx.x, 256};
  __shared__ u32 lds_digits[C::NW][256];
  const DigitMem dm{&lds_digits[0][threadIdx.x], 256};
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * BATCH) lane_pass<C, BATCH>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, base, T, ws, dm);
}

}  // namespace vbct
}  // namespace ecgpu
