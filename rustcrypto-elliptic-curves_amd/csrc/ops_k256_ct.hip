// secp256k1 kernels that run on SECRET scalars - the constant-time variable-base kernel (ECDH), the constant-time fixed-base kernel
// (signing, key generation) and the reference schedules (exact X, Y, Z; the reference's own constant-time discipline) - compiled with
// ECGPU_K256_BRANCHFREE: in this translation unit the rare carry paths of the field additions, subtractions, small shifts and of the
// last fold of a multiplication (fe_k256.hpp) execute unconditionally, so these kernels contain no branch whose direction depends on
// a value derived from a secret (missing 4).  What branches remain are loop counters and the public batch size
// (profiles/r04_k256_ct_branches.txt lists every s_cbranch of these kernels with the comparison that feeds it).
// The throughput kernels (ops_k256.hip) keep the branches: they are measurably cheaper and their data is public.
#ifndef ECGPU_K256_CT_KEEP_BRANCHES        // A/B switch (make ab-ctbranch): the same kernels with the rare carry paths as branches, for the cost and the counters
#define ECGPU_K256_BRANCHFREE 1
#endif
#include "ecgpu_internal.hpp"
#include "kernels.hpp"
#include "fixedbase.hpp"
#include "varbase_ct_k256.hpp"
using namespace ecgpu;

#ifndef ECGPU_REF_GRID_MULT
#define ECGPU_REF_GRID_MULT 4
#endif
#ifndef K256_CT_BATCH
#define K256_CT_BATCH 16
#endif
#ifndef K256_CT_WAVES
#define K256_CT_WAVES 3        // 168 VGPRs: 39.6 ms per 2^22 against 41.5 ms at 4 waves per SIMD (128 VGPRs); 32 results per pass: no difference (round 3)
#endif
#ifndef K256_FBCT_WAVES
#define K256_FBCT_WAVES 4      // curve_ops.hpp FBCT_WAVES for 8-word fields
#endif
#ifndef FBCT_GRID_MULT
#define FBCT_GRID_MULT 4        // as in curve_ops.hpp
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

static int tab_reserve(ecgpu_ctx* c, size_t need) {
  if (need <= c->tab_ws_cap) return 0;
  if (c->tab_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->tab_ws)); c->tab_ws = nullptr; c->tab_ws_cap = 0; }
  HIPCHK(c, hipMalloc(&c->tab_ws, need));
  c->tab_ws_cap = need;
  return 0;
}

size_t ecgpuint_k256_ct_pass_units(const ecgpu_ctx* c) { return (size_t)c->num_cus * K256_CT_WAVES * 256 * K256_CT_BATCH; }

int ecgpuint_k256_mul_ct(ecgpu_ctx* c, const uint32_t* sc, const uint32_t* pts, int pt_fmt, uint32_t* out, int out_fmt, uint8_t* out_inf, size_t n) {
  // K256_CT_GRID_MULT (round 4): more workgroups than the chip holds, STATIC work per workgroup - the hardware hands the waiting workgroups out as the
  // favoured waves of a SIMD leave (sched.hpp explains the imbalance; the constant-time kernels do not draw their work from a counter)
#ifndef K256_CT_GRID_MULT
#define K256_CT_GRID_MULT 4      // ECDH kernel, 2^22 units: 1 / 2 / 4 / 8 = 42.6 / 39.8 / 39.0 / 38.9 ms (profiles/r04_ab_measurements.txt, set seven)
#endif
  const dim3 grid(ecgpu_grid_for(c, n, K256_CT_WAVES * K256_CT_GRID_MULT));
  int rc = tab_reserve(c, (size_t)grid.x * 256 * vbct::k256_lane_chunks<K256_CT_BATCH>() * sizeof(vbct::Chunk));
  if (rc) return rc;
  hipLaunchKernelGGL((k256_mul_ct_kernel<K256_CT_BATCH, K256_CT_WAVES>), grid, dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n,
                     (vbct::Chunk*)c->tab_ws);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int ecgpuint_k256_mul_gen_ct(ecgpu_ctx* c, const uint32_t* sc, const void* table, uint32_t* out, int out_fmt, uint8_t* out_inf, size_t n) {
  hipLaunchKernelGGL((fb::mul_ct_kernel<CurveK256, 8, K256_FBCT_WAVES>), dim3(ecgpu_grid_oversubscribed(c, n, K256_FBCT_WAVES, 8, FBCT_GRID_MULT)), dim3(256), 0, c->stream, sc,
                     (const AffEntry<CurveK256>*)table, out, out_fmt, out_inf, n);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int ecgpuint_k256_reference(ecgpu_ctx* c, const uint32_t* sc, const uint32_t* pts, int pt_fmt, size_t terms, const void* gen_table, uint32_t* out, int out_fmt,
                            uint8_t* out_inf, size_t n) {
  using C = CurveK256;
  // four times the workgroups the chip holds (the kernels walk their units with a grid stride, one at a time: no per-lane batch to shrink): the hardware
  // hands the waiting workgroups out as the favoured waves of a SIMD leave (sched.hpp)
  const unsigned g = ecgpu_grid_for(c, n, 4 * ECGPU_REF_GRID_MULT);
  if (!pts) {
    hipLaunchKernelGGL((mul_gen_ref_kernel<C>), dim3(g), dim3(256), 0, c->stream, sc, (const PtK256*)gen_table, out, out_fmt, out_inf, n);
  } else if (terms == 1) {
    hipLaunchKernelGGL((lincomb_ref_kernel<C, 1>), dim3(g), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
  } else if (terms == 2) {
    hipLaunchKernelGGL((lincomb_ref_kernel<C, 2>), dim3(g), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
  } else {
    // 3 .. 1024 terms: the reference's interleaved schedule with a run-time term count (mul_k256.hpp), tables in a per-lane global
    // scratch; the lane count is capped so that the scratch stays below 8 GB
    const size_t per_lane = terms * (16 * sizeof(PtK256) + 10 * sizeof(u32));
    size_t blocks = (n + 255) / 256, cap = (size_t)c->num_cus * ECGPU_REF_WAVES, budget = (((size_t)8 << 30) / (per_lane * 256));
    if (blocks > cap) blocks = cap;
    if (blocks > budget) blocks = budget ? budget : 1;
    const size_t lanes = blocks * 256, sz_tab = (lanes * terms * 16 * sizeof(PtK256) + 255) & ~(size_t)255;
    int rc = tab_reserve(c, sz_tab + lanes * terms * 10 * sizeof(u32));
    if (rc) return rc;
    hipLaunchKernelGGL((k256_lincomb_ref_n_kernel<C>), dim3((unsigned)blocks), dim3(256), 0, c->stream, sc, pts, pt_fmt, (int)terms, out, out_fmt, out_inf, n,
                       (PtK256*)c->tab_ws, (u32*)((char*)c->tab_ws + sz_tab));
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
