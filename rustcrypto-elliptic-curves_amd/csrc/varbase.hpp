// Variable-base scalar multiplication kernel for P-256 / P-384: grid-stride passes over the per-lane body of
// varbase_lane.hpp (schedule and workspace described there).  Device code only.
#pragma once
#include "varbase_lane.hpp"
#include "varbase_ct.hpp"
#include "kernels.hpp"
#include "sched.hpp"

namespace ecgpu {
namespace vb {

template <class C, int BATCH, int WAVES, int NT = 1, int WB = 4>
__global__ void __launch_bounds__(256, WAVES) mul_kernel(const u32* scalars, const u32* points, int pt_fmt, u32* out, int out_fmt,
                                                         uint8_t* out_inf, size_t n, LaneWs<C, BATCH, WB>* ws_all, WaveSched sched) {
  LaneWs<C, BATCH, WB>& ws = ws_all[(size_t)blockIdx.x * blockDim.x + threadIdx.x];
  __shared__ u32 lds_digits[NT * digit_words<C, WB>()][256];
  const DigitMem dm{&lds_digits[0][threadIdx.x], 256};
  // Every wave draws SMALL passes of 64 x u consecutive units from one counter (sched.hpp): lane l takes units lo + l, lo + l + 64, ..  The units of
  // a pass share the inversion of their tables; the results stay in res[] across passes and are flushed with ONE output inversion when the buffer
  // is full or the work has run out.
  constexpr int UB = BATCH / NT;
  Jac<C> res[UB];
  typename C::Fe pre[UB];
  size_t idx[UB];
  int cnt = 0, slots = 0;            // results buffered by this lane; per-lane units drawn since the last flush (wave-uniform)
  for (;;) {
    size_t lo, hi;
    const bool more = wave_next_chunk(sched, lo, hi);
    if (more) {
      lane_pass<C, BATCH, NT, WB>(scalars, points, pt_fmt, out, out_fmt, out_inf, hi, lo + (threadIdx.x & 63u), 64, ws, dm, res, idx, &cnt);
      slots += (int)((hi - lo + 63) / 64);
    }
    if (!more || slots + (int)sched.chunk_units > UB) {
      if (cnt) jac::store_batch_affine<C>(res, pre, cnt, 0, 0, out, out_fmt, out_inf, idx);
      cnt = 0;
      slots = 0;
    }
    if (!more) break;
  }
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
