// NIST P384 kernels and launchers (one translation unit per curve: the library builds in parallel).
#include "curve_ops.hpp"
#include "varbase.hpp"
#ifndef VBB
#define VBB 8          // units per lane and pass of the variable-base kernel (tables share one inversion)
#endif
using namespace ecgpu;

template <>
int CurveOps<CurveP384>::lincomb_fast(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t terms, u32* out, int out_fmt, uint8_t* out_inf, size_t n) {
  if (terms == 1 && !pts) return mul_gen_fast(c, sc, out, out_fmt, out_inf, n);
  if (terms == 1 && pts) {
    // ECGPU_VB_WAVES (2/3/4) picks the occupancy variant; the default is the measured best
    static const int waves = [] { const char* e = getenv("ECGPU_VB_WAVES"); int w = e ? atoi(e) : 4; return (w < 2 || w > 4) ? 4 : w; }();
    const dim3 grid(ecgpu_grid_for(c, n, waves));
    // per-lane workspace (8 tables of 8 points and the prefix products of their shared inversion per resident lane),
    // grow-only, shared with the other curves' kernels
    const size_t ws_need = (size_t)grid.x * 256 * sizeof(vb::LaneWs<CurveP384, VBB>);
    if (ws_need > c->tab_ws_cap) {
      if (c->tab_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->tab_ws)); c->tab_ws = nullptr; c->tab_ws_cap = 0; }
      HIPCHK(c, hipMalloc(&c->tab_ws, ws_need));
      c->tab_ws_cap = ws_need;
    }
    vb::LaneWs<CurveP384, VBB>* ws = (vb::LaneWs<CurveP384, VBB>*)c->tab_ws;
    if (waves == 2) hipLaunchKernelGGL((vb::mul_kernel<CurveP384, VBB, 2>), grid, dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n, ws);
    else if (waves == 3) hipLaunchKernelGGL((vb::mul_kernel<CurveP384, VBB, 3>), grid, dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n, ws);
    else hipLaunchKernelGGL((vb::mul_kernel<CurveP384, VBB, 4>), grid, dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n, ws);
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
