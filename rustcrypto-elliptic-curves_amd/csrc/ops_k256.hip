// secp256k1 kernels and launchers (one translation unit per curve: the library builds in parallel).
#include "curve_ops.hpp"
using namespace ecgpu;
// per-lane units a wave draws at a time from the work counter (sched.hpp); the results of up to K256_FAST_BATCH units share one inversion
#ifndef K256_CHUNK_UNITS
#define K256_CHUNK_UNITS 2
#endif
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
#ifdef K256_GRID_PER_CU            // A/B switch: workgroups per CU of the single-term kernel's launch, whatever its occupancy target
  const unsigned grid = ecgpu_grid_for(c, n, terms == 2 ? 4 : K256_GRID_PER_CU);
#else
  const unsigned grid = ecgpu_grid_for(c, n, terms == 2 ? 4 : waves);
#endif
  // per-lane table workspace: 2 KB per resident lane for the single-term kernel (16 entries and their beta slots: 537 MB at 4 waves/SIMD), 2 x 1 KB for the two-term kernel
#ifdef K256_BLOCK_TIMES
  const size_t ws_need = (size_t)grid * 256 * sizeof(TabSlotK256) * (terms == 2 ? 2 * K256Win<4>::SLOTS : K256Win<K256_WB>::SLOTS) + (size_t)grid * 32 + 16;
#else
  const size_t ws_need = (size_t)grid * 256 * sizeof(TabSlotK256) * (terms == 2 ? 2 * K256Win<4>::SLOTS : K256Win<K256_WB>::SLOTS);
#endif
  if (ws_need > c->tab_ws_cap) {
    if (c->tab_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->tab_ws)); c->tab_ws = nullptr; c->tab_ws_cap = 0; }
    HIPCHK(c, hipMalloc(&c->tab_ws, ws_need));
    c->tab_ws_cap = ws_need;
  }
  TabSlotK256* ws = (TabSlotK256*)c->tab_ws;
  unsigned long long* ctr = ecgpu_sched_counter(c);
  if (!ctr) return ECGPU_ERR_RUNTIME;
  if (terms == 2) {
    hipLaunchKernelGGL((k256_lincomb2_fast_kernel<16, 4>), dim3(grid), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n, ws,
                       WaveSched{ctr, (unsigned long long)n, grid * 4u, 1u, 1u});
    HIPCHK(c, hipGetLastError());
    return 1;
  }
  const WaveSched sched{ctr, (unsigned long long)n, grid * 4u, (unsigned)K256_CHUNK_UNITS, 1u};
  if (waves == 3)
    hipLaunchKernelGGL((k256_mul_fast_kernel<K256_FAST_BATCH, 3>), dim3(grid), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n, ws, sched);
  else
    hipLaunchKernelGGL((k256_mul_fast_kernel<K256_FAST_BATCH, 4>), dim3(grid), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n, ws, sched);
  HIPCHK(c, hipGetLastError());
  return 1;
}
// secret scalars on a variable base (ECDH): varbase_ct_k256.hpp, in the branch-free translation unit (ops_k256_ct.hip)
template <>
int CurveOps<CurveK256>::mul_ct(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, u32* out, int out_fmt, uint8_t* out_inf, size_t n) {
  int rc = ecgpuint_k256_mul_ct(c, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
  return rc ? rc : 1;
}
template <>
size_t CurveOps<CurveK256>::pass_units_points(const ecgpu_ctx* c, size_t terms, unsigned flags) {
  if (flags & ECGPU_EXACT_REFERENCE) return resident_lanes(c, 4);                          // lincomb_ref_kernel: one unit per lane
  if (flags & ECGPU_SECRET_SCALARS) return ecgpuint_k256_ct_pass_units(c);
  if (terms == 2) return resident_lanes(c, 4) * 16;
  if (terms > 2) return resident_lanes(c, 3);                                              // straus::lincomb_kernel<CurveK256, 3>
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
