// secp256k1 kernels and launchers (one translation unit per curve: the library builds in parallel).
#include "curve_ops.hpp"
using namespace ecgpu;

template <>
int CurveOps<CurveK256>::lincomb_fast(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t terms, u32* out, int out_fmt,
                                      uint8_t* out_inf, size_t n) {
  if (terms != 1 || !pts) return 0;
  // ECGPU_K256_FAST_WAVES (2/3/4) picks the occupancy variant; default chosen from measurements (profiles/r01_kbench_variants.txt)
  static const int waves = [] { const char* e = getenv("ECGPU_K256_FAST_WAVES"); int w = e ? atoi(e) : 4; return (w < 3 || w > 4) ? 4 : w; }();
  if (waves == 3)
    hipLaunchKernelGGL((k256_mul_fast_kernel<16, 3>), dim3(ecgpu_grid_for(c, n, 3)), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
  else
    hipLaunchKernelGGL((k256_mul_fast_kernel<16, 4>), dim3(ecgpu_grid_for(c, n, 4)), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
  HIPCHK(c, hipGetLastError());
  return 1;
}
template <>
int CurveOps<CurveK256>::msm(ecgpu_ctx* c, const u32*, const u32*, int, size_t, u32*, int) {
  return ecgpu_set_err(c, ECGPU_ERR_UNSUPPORTED, "ecgpu_msm: not built yet");
}
const ecgpu_curve_ops* ecgpu_ops_k256() { return CurveOps<CurveK256>::table(); }
