// Pippenger MSM for k256: msm_kernels.hpp instantiated in its own translation unit (the library builds in parallel).
#include "msm_kernels.hpp"
using namespace ecgpu;
int ecgpu_msm_k256(ecgpu_ctx* c, const uint32_t* sc, const uint32_t* pts, int pt_fmt, size_t n, uint32_t* out, int out_fmt, ecgpu_msm_mul_fn mul) {
  return msm::msm_run<CurveK256>(c, sc, pts, pt_fmt, n, out, out_fmt, [&](const u32* s, const u32* p, int fmt, u32* prod, size_t cnt) { return mul(c, s, p, fmt, prod, cnt); });
}
