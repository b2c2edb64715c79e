// NIST P256 kernels and launchers (one translation unit per curve: the library builds in parallel).
#include "curve_ops.hpp"
using namespace ecgpu;

template <>
int CurveOps<CurveP256>::lincomb_fast(ecgpu_ctx* c, const u32* sc, const u32* pts, int, size_t terms, u32* out, int out_fmt, uint8_t* out_inf, size_t n) {
  if (terms == 1 && !pts) return mul_gen_fast(c, sc, out, out_fmt, out_inf, n);
  return 0;   // variable base: reference schedule
}
template <>
int CurveOps<CurveP256>::msm(ecgpu_ctx* c, const u32*, const u32*, int, size_t, u32*, int) {
  return ecgpu_set_err(c, ECGPU_ERR_UNSUPPORTED, "ecgpu_msm: k256 only");
}
const ecgpu_curve_ops* ecgpu_ops_p256() { return CurveOps<CurveP256>::table(); }
