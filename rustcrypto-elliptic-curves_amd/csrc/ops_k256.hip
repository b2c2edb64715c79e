// secp256k1 kernels and launchers (one translation unit per curve: the library builds in parallel).
#include "curve_ops.hpp"
#include "varbase_ct_k256.hpp"
using namespace ecgpu;
#ifndef K256_FAST_BATCH
#define K256_FAST_BATCH 32   // results per lane that share one inversion in the variable-base kernel (16: -0.4 %)
#endif

template <>
int CurveOps<CurveK256>::lincomb_fast(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t terms, u32* out, int out_fmt,
                                      uint8_t* out_inf, size_t n) {
  if (terms == 1 && !pts) return mul_gen_fast(c, sc, out, out_fmt, out_inf, n);
  if (terms != 1 && terms != 2) return 0;
  if (!pts) return 0;
  // ECGPU_OPT_K256_WAVES (3 / 4) picks the occupancy variant; default chosen from measurements (profiles/r01_kbench_variants.txt)
  const int waves = c->opt[ECGPU_OPT_K256_WAVES] == 3 ? 3 : 4;
  const unsigned grid = ecgpu_grid_for(c, n, terms == 2 ? 4 : waves);
  // per-lane table workspace: 1 KB per resident lane (268 MB at 4 waves/SIMD on 256 CUs), grow-only
  const size_t ws_need = (size_t)grid * 256 * K256_TAB_SLOTS * sizeof(TabSlotK256) * terms;
  if (ws_need > c->tab_ws_cap) {
    if (c->tab_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->tab_ws)); c->tab_ws = nullptr; c->tab_ws_cap = 0; }
    HIPCHK(c, hipMalloc(&c->tab_ws, ws_need));
    c->tab_ws_cap = ws_need;
  }
  TabSlotK256* ws = (TabSlotK256*)c->tab_ws;
  if (terms == 2) {
    hipLaunchKernelGGL((k256_lincomb2_fast_kernel<16, 4>), dim3(grid), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n, ws);
    HIPCHK(c, hipGetLastError());
    return 1;
  }
  if (waves == 3)
    hipLaunchKernelGGL((k256_mul_fast_kernel<K256_FAST_BATCH, 3>), dim3(grid), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n, ws);
  else
    hipLaunchKernelGGL((k256_mul_fast_kernel<K256_FAST_BATCH, 4>), dim3(grid), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n, ws);
  HIPCHK(c, hipGetLastError());
  return 1;
}
// secret scalars on a variable base (ECDH): varbase_ct_k256.hpp - GLV, Jacobian formulas (exception-free on this loop's operands) over a
// per-lane common-Z table, one masked scan per window for both halves; K256_CT_BATCH results per lane share the output inversion
#ifndef K256_CT_BATCH
#define K256_CT_BATCH 16
#endif
#ifndef K256_CT_WAVES
#define K256_CT_WAVES 3        // 168 VGPRs, 26 spilled: 39.6 ms per 2^22 against 41.5 ms at 4 waves per SIMD (128 VGPRs, 82 spilled); 32 results per pass: no difference
#endif
template <int BATCH, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k256_mul_ct_kernel(const u32* scalars, const u32* points, int pt_fmt, u32* out, int out_fmt, uint8_t* out_inf,
                                                                 size_t n, vbct::Chunk* ws_all) {
  const vbct::LaneMem ws{ws_all + (size_t)blockIdx.x * vbct::k256_lane_chunks<BATCH>() * 256 + threadIdx.x, 256};
  __shared__ u32 lds_digits[8][256];
  const DigitMem dm{&lds_digits[0][threadIdx.x], 256};
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * BATCH) vbct::lane_pass_k256<BATCH>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, base, T, ws, dm);
}
template <>
int CurveOps<CurveK256>::mul_ct(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, u32* out, int out_fmt, uint8_t* out_inf, size_t n) {
  const dim3 grid(ecgpu_grid_for(c, n, K256_CT_WAVES));
  const size_t ws_need = (size_t)grid.x * 256 * vbct::k256_lane_chunks<K256_CT_BATCH>() * sizeof(vbct::Chunk);
  if (ws_need > c->tab_ws_cap) {
    if (c->tab_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->tab_ws)); c->tab_ws = nullptr; c->tab_ws_cap = 0; }
    HIPCHK(c, hipMalloc(&c->tab_ws, ws_need));
    c->tab_ws_cap = ws_need;
  }
  hipLaunchKernelGGL((k256_mul_ct_kernel<K256_CT_BATCH, K256_CT_WAVES>), grid, dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n,
                     (vbct::Chunk*)c->tab_ws);
  HIPCHK(c, hipGetLastError());
  return 1;
}
template <>
size_t CurveOps<CurveK256>::pass_units_points(const ecgpu_ctx* c, size_t terms, unsigned flags) {
  if (flags & ECGPU_EXACT_REFERENCE) return resident_lanes(c, 4);                          // lincomb_ref_kernel: one unit per lane
  if (flags & ECGPU_SECRET_SCALARS) return resident_lanes(c, K256_CT_WAVES) * K256_CT_BATCH;
  if (terms == 2) return resident_lanes(c, 4) * 16;
  if (terms > 2) return resident_lanes(c, 4);
  return resident_lanes(c, c->opt[ECGPU_OPT_K256_WAVES] == 3 ? 3 : 4) * K256_FAST_BATCH;
}
// Pippenger MSM (msm.hpp, msm_kernels.hpp; instantiated in msm_k256.hip)
static int k256_mul_for_msm(ecgpu_ctx* c, const u32* s, const u32* p, int fmt, u32* prod, size_t cnt) {
  return CurveOps<CurveK256>::lincomb(c, s, p, fmt, 1, prod, FMT_AFFINE, nullptr, cnt, 0);
}
template <>
int CurveOps<CurveK256>::msm(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t n, u32* out, int out_fmt) {
  return ecgpu_msm_k256(c, sc, pts, pt_fmt, n, out, out_fmt, k256_mul_for_msm);
}
const ecgpu_curve_ops* ecgpu_ops_k256() { return CurveOps<CurveK256>::table(); }
