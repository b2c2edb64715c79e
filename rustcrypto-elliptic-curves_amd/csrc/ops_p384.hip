// NIST P384 kernels and launchers (one translation unit per curve: the library builds in parallel).
#include "curve_ops.hpp"
#include "varbase.hpp"
using namespace ecgpu;

template <>
int CurveOps<CurveP384>::lincomb_fast(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t terms, u32* out, int out_fmt, uint8_t* out_inf, size_t n) {
  if (terms == 1 && !pts) return mul_gen_fast(c, sc, out, out_fmt, out_inf, n);
  if (terms == 1 && pts) {
    hipLaunchKernelGGL((vb::mul_kernel<CurveP384, 8, 2>), dim3(ecgpu_grid_for(c, n, 2)), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
    HIPCHK(c, hipGetLastError());
    return 1;
  }
  return 0;   // lincomb with several terms: reference schedule
}
template <>
int CurveOps<CurveP384>::msm(ecgpu_ctx* c, const u32*, const u32*, int, size_t, u32*, int) {
  return ecgpu_set_err(c, ECGPU_ERR_UNSUPPORTED, "ecgpu_msm: k256 only");
}
const ecgpu_curve_ops* ecgpu_ops_p384() { return CurveOps<CurveP384>::table(); }
