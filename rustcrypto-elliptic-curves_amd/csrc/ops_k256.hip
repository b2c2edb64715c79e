// secp256k1 kernels and launchers (one translation unit per curve: the library builds in parallel).
#include "curve_ops.hpp"
#include "msm_kernels.hpp"
using namespace ecgpu;
#ifndef K256_FAST_BATCH
#define K256_FAST_BATCH 32   // results per lane that share one inversion in the variable-base kernel (16: -0.4 %)
#endif
#ifndef MSM_BUCKET_WGS_PER_CU
#define MSM_BUCKET_WGS_PER_CU 64     // cap on bucket-sum workgroups per CU: above tasks / 256, so every lane takes one task and the dispatcher balances the CUs
#endif
#ifndef MSM_BUCKET_SPLIT
#define MSM_BUCKET_SPLIT 8           // lanes per bucket (msm_kernels.hpp: bucket_sum_kernel)
#endif

template <>
int CurveOps<CurveK256>::lincomb_fast(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t terms, u32* out, int out_fmt,
                                      uint8_t* out_inf, size_t n) {
  if (terms == 1 && !pts) return mul_gen_fast(c, sc, out, out_fmt, out_inf, n);
  if (terms != 1 && terms != 2) return 0;
  if (!pts) return 0;
  // ECGPU_K256_FAST_WAVES (2/3/4) picks the occupancy variant; default chosen from measurements (profiles/r01_kbench_variants.txt)
  static const int waves = [] { const char* e = getenv("ECGPU_K256_FAST_WAVES"); int w = e ? atoi(e) : 4; return (w < 3 || w > 4) ? 4 : w; }();
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
// Pippenger MSM (msm_k256.hpp).  All stages run on c->stream out of one grow-only workspace.
template <>
int CurveOps<CurveK256>::msm(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t n, u32* out, int out_fmt) {
  using namespace msm;
  // the sorted (term, window) list is indexed with 32-bit offsets and an entry keeps the term in 31 bits
  static_assert((size_t)NDIG * SLAB_TERMS < ((size_t)1 << 32) && SLAB_TERMS - 1 <= ENTRY_INDEX_MASK, "32-bit offsets and 24-bit term indices within a slab");
  if (((uintptr_t)pts & 15) || ((uintptr_t)sc & 3)) return ecgpu_set_err(c, ECGPU_ERR_ARG, "ecgpu_msm: device points must be 16-byte aligned");
  const size_t nb = (size_t)NWIN * NBUCKET;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  // ECGPU_MSM_SMALL = 0 forces the bucket method for every size (measurements, tests of the bucket path on small inputs)
  // (read per call so that one process can exercise both paths)
  const char* small_env = getenv("ECGPU_MSM_SMALL");
  const bool small_path = !(small_env && atoi(small_env) == 0);
  if (small_path && n > 0 && n < SMALL_MSM_TERMS) {
    // n scalar multiplications on the throughput kernel, then a two-level sum of the products
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    const size_t sz_prod = al(n * 64), sz_part = al((size_t)blocks * sizeof(JacK256)), sz_win = al(sizeof(JacK256));
    const size_t need = sz_prod + sz_part + sz_win;
    if (need > c->msm_ws_cap) {
      if (c->msm_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->msm_ws)); c->msm_ws = nullptr; c->msm_ws_cap = 0; }
      HIPCHK(c, hipMalloc(&c->msm_ws, need));
      c->msm_ws_cap = need;
    }
    char* p = (char*)c->msm_ws;
    u32* prod = (u32*)p; p += sz_prod;
    JacK256* partial = (JacK256*)p; p += sz_part;
    JacK256* win = (JacK256*)p;
    int rc = lincomb(c, sc, pts, pt_fmt, 1, prod, FMT_AFFINE, nullptr, n, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(sum_affine_kernel, dim3(blocks), dim3(256), 0, c->stream, (const u32*)prod, n, partial);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, c->stream, (const JacK256*)partial, blocks, win);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(64), 0, c->stream, (const JacK256*)win, 1, out, out_fmt);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  // Large sums run in slabs of at most SLAB_TERMS terms (a sorted entry keeps the term index in 24 bits); every slab goes
  // through the whole pipeline down to its NWIN window sums, which are added up before the final Horner pass.
  // ECGPU_MSM_SLAB overrides the slab size (tests exercise the slab loop on small inputs).
  const char* slab_env = getenv("ECGPU_MSM_SLAB");
  size_t slab = slab_env ? (size_t)atoll(slab_env) : SLAB_TERMS;
  if (slab < 1024 || slab > SLAB_TERMS) slab = SLAB_TERMS;
  const size_t m = n < slab ? n : slab;                // terms of the largest slab: sizes the workspace
  const size_t sz_aff = (pt_fmt == FMT_PROJECTIVE) ? al(m * 64) : 0, sz_endo = al(m * 64);
  const size_t sz_off = al((nb + 1) * 4), sz_coff = al((NCB + 1) * 4), sz_tot = al((size_t)NCB * 4), sz_sorted = al((size_t)NDIG * m * 4 + 32);
  // level A of the sort: one 1024-thread workgroup per CU, the chunks of a window side by side
  const int nch = (c->num_cus - 1) / NWIN > 0 ? (c->num_cus - 1) / NWIN : 1;
  const size_t sz_part = al((size_t)NWIN * nch * NCOARSE * 4);
  const size_t ms = (m + 3) & ~(size_t)3;              // row stride of the digit arrays: four terms per load
  const size_t sz_mag = al((size_t)NDIG * ms * 2), sz_sgn = al(ms * 4);
  static const int wgs_per_cu = [] { const char* e = getenv("ECGPU_MSM_WGS"); int v = e ? atoi(e) : MSM_BUCKET_WGS_PER_CU; return (v < 1 || v > 1024) ? MSM_BUCKET_WGS_PER_CU : v; }();
  static const int split = [] { const char* e = getenv("ECGPU_MSM_SPLIT"); int v = e ? atoi(e) : MSM_BUCKET_SPLIT; return (v < 1 || v > 64) ? MSM_BUCKET_SPLIT : v; }();
  const size_t sz_buckets = al(nb * sizeof(JacK256)), sz_parts = al(nb * split * sizeof(JacK256));
  const size_t n0 = (size_t)NWIN * NSEG0, n1 = (size_t)NWIN * NSEG1, nsw = (size_t)NWIN * NSUMW;
  const size_t sz_l0 = al(n0 * sizeof(JacK256)), sz_l1 = al(n1 * sizeof(JacK256)), sz_sw = al(nsw * sizeof(JacK256)), sz_win = al(NWIN * sizeof(JacK256));
  // heavy buckets (more than `cap` entries): at most L / cap of them, at most 2 L / cap + 1 chunks (msm_kernels.hpp, step 4)
  const size_t L = (size_t)NDIG * m;
  const u32 cap = (u32)((8 * (m / NBUCKET) > 2048) ? 8 * (m / NBUCKET) : 2048);
  const size_t hmax = L / cap + 1, cmax = 2 * (L / cap) + 2;
  const size_t sz_hctr = al(8), sz_heavy = al(hmax * sizeof(HeavyBucket)), sz_chunks = al(cmax * sizeof(HeavyChunk)), sz_partial = al(cmax * sizeof(JacK256));
  const size_t need = sz_aff + sz_endo + sz_mag + sz_sgn + sz_off + sz_coff + sz_tot + sz_part + 2 * sz_sorted + sz_buckets + sz_parts + 2 * sz_l0 + 2 * sz_l1 + sz_sw +
                      2 * sz_win + sz_hctr + sz_heavy + sz_chunks + sz_partial;
  if (need > c->msm_ws_cap) {
    if (c->msm_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->msm_ws)); c->msm_ws = nullptr; c->msm_ws_cap = 0; }
    HIPCHK(c, hipMalloc(&c->msm_ws, need));
    c->msm_ws_cap = need;
  }
  char* p = (char*)c->msm_ws;
  u32* aff = (u32*)p; p += sz_aff;
  u32* endo = (u32*)p; p += sz_endo;
  uint16_t* mag = (uint16_t*)p; p += sz_mag;
  u32* sgn = (u32*)p; p += sz_sgn;
  u32* offsets = (u32*)p; p += sz_off;
  u32* coarse_off = (u32*)p; p += sz_coff;
  u32* tot = (u32*)p; p += sz_tot;
  u32* part = (u32*)p; p += sz_part;
  u32* mid = (u32*)p; p += sz_sorted;
  u32* sorted = (u32*)p; p += sz_sorted;
  JacK256* buckets = (JacK256*)p; p += sz_buckets;
  JacK256* parts = (JacK256*)p; p += sz_parts;
  JacK256* t0 = (JacK256*)p; p += sz_l0;
  JacK256* w0 = (JacK256*)p; p += sz_l0;
  JacK256* t1 = (JacK256*)p; p += sz_l1;
  JacK256* w1 = (JacK256*)p; p += sz_l1;
  JacK256* sumw0 = (JacK256*)p; p += sz_sw;
  JacK256* win = (JacK256*)p; p += sz_win;
  JacK256* win_slab = (JacK256*)p; p += sz_win;
  u32* heavy_ctr = (u32*)p; p += sz_hctr;
  HeavyBucket* heavy = (HeavyBucket*)p; p += sz_heavy;
  HeavyChunk* chunks = (HeavyChunk*)p; p += sz_chunks;
  JacK256* partial = (JacK256*)p;
  const size_t pin = (pt_fmt == FMT_PROJECTIVE) ? 24 : 16;      // 32-bit words per input point
  const unsigned cb_grid = (unsigned)((NCB + 255) / 256);
  for (size_t s0 = 0; s0 < n; s0 += slab) {
    const size_t cnt = (n - s0 < slab) ? n - s0 : slab;
    const u32* ssc = sc + s0 * 8;
    const u32* xy = pts + s0 * pin;
    if (pt_fmt == FMT_PROJECTIVE) {
      hipLaunchKernelGGL(to_affine_kernel, dim3(ecgpu_grid_for(c, cnt, 8)), dim3(256), 0, c->stream, xy, aff, cnt);
      xy = aff;
    }
    JacK256* wdst = (s0 == 0) ? win : win_slab;
    hipLaunchKernelGGL(digits_kernel, dim3(ecgpu_grid_for(c, cnt, 8)), dim3(256), 0, c->stream, ssc, cnt, ms, mag, sgn);
    hipLaunchKernelGGL(endo_points_kernel, dim3(ecgpu_grid_for(c, cnt, 8)), dim3(256), 0, c->stream, xy, endo, cnt);
    hipLaunchKernelGGL(coarse_hist_kernel, dim3((unsigned)(NWIN * nch)), dim3(1024), 0, c->stream, (const uint16_t*)mag, cnt, ms, nch, part);
    hipLaunchKernelGGL(coarse_totals_kernel, dim3(cb_grid), dim3(256), 0, c->stream, (const u32*)part, nch, tot);
    hipLaunchKernelGGL(coarse_scan_kernel, dim3(1), dim3(1024), 0, c->stream, (const u32*)tot, coarse_off, offsets + nb);
    hipLaunchKernelGGL(coarse_cursors_kernel, dim3(cb_grid), dim3(256), 0, c->stream, part, nch, (const u32*)coarse_off);
    hipLaunchKernelGGL(coarse_scatter_kernel, dim3((unsigned)(NWIN * nch)), dim3(1024), 0, c->stream, (const uint16_t*)mag, (const u32*)sgn, cnt, ms, nch,
                       (const u32*)part, mid, sorted);
    hipLaunchKernelGGL(fine_sort_kernel, dim3((unsigned)NCB), dim3(256), 0, c->stream, (const u32*)mid, (const u32*)coarse_off, offsets, sorted);
    HIPCHK(c, hipMemsetAsync(heavy_ctr, 0, 8, c->stream));
    hipLaunchKernelGGL(bucket_sum_kernel, dim3(ecgpu_grid_for(c, nb * split, wgs_per_cu)), dim3(256), 0, c->stream, xy, (const u32*)endo, offsets, sorted, parts,
                       (int)nb, split, cap, heavy_ctr, heavy, chunks);
    hipLaunchKernelGGL(heavy_chunk_kernel, dim3((unsigned)c->num_cus * 8), dim3(256), 0, c->stream, xy, (const u32*)endo, (const u32*)offsets, (const u32*)sorted, cap,
                       (const u32*)heavy_ctr, (const HeavyChunk*)chunks, partial);
    hipLaunchKernelGGL(heavy_finish_kernel, dim3((unsigned)c->num_cus), dim3(256), 0, c->stream, (const u32*)heavy_ctr, (const HeavyBucket*)heavy,
                       (const JacK256*)partial, buckets);
    hipLaunchKernelGGL(bucket_combine_kernel, dim3(ecgpu_grid_for(c, nb, 8)), dim3(256), 0, c->stream, (const JacK256*)parts, buckets, (int)nb, split, (const u32*)offsets, cap);
    hipLaunchKernelGGL(segment_kernel, dim3((unsigned)((n0 + 63) / 64)), dim3(64), 0, c->stream, buckets, t0, w0, SEG0, (int)n0);
    hipLaunchKernelGGL(segment_kernel, dim3((unsigned)((n1 + 63) / 64)), dim3(64), 0, c->stream, t0, t1, w1, SEG1, (int)n1);
    hipLaunchKernelGGL(sum_kernel, dim3((unsigned)((nsw + 63) / 64)), dim3(64), 0, c->stream, w0, sumw0, SUMW_LEN, (int)nsw);
    hipLaunchKernelGGL(window_kernel, dim3(NWIN), dim3(NSEG1), 0, c->stream, t1, w1, sumw0, wdst);
    if (s0 != 0) hipLaunchKernelGGL(windows_accumulate_kernel, dim3(1), dim3(64), 0, c->stream, win, (const JacK256*)win_slab);
  }
  hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(64), 0, c->stream, (const JacK256*)win, (int)NWIN, out, out_fmt);
  HIPCHK(c, hipGetLastError());
  return 0;
}
const ecgpu_curve_ops* ecgpu_ops_k256() { return CurveOps<CurveK256>::table(); }
