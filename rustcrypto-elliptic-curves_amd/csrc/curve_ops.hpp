// Generic kernel launchers over a curve traits class; instantiated once per curve in ops_*.hip.
#pragma once
#include "ecgpu_internal.hpp"
#include "kernels.hpp"
#include "fixedbase.hpp"
#include "ecdsa_kernels.hpp"
#include "ecdh_kernels.hpp"
#include "sec1_kernels.hpp"
#include "schnorr_kernels.hpp"
#include "h2c_kernels.hpp"
#include "straus.hpp"
// Results per lane that share one inversion in the wide fixed-base kernels.  With only 12-16 additions per result
// the inversion's share is large: measured at 2^24 scalars, batch 16 / 32 / 64: p256 0.96 / 1.03 / 1.06, k256 1.19 /
// 1.24 / 1.25 x 10^9 per second; p384 (2^22) 282 / 280 / 279 x 10^6 (its results are 144 bytes each in the private
// segment), hence 16 there.
#ifndef ECGPU_REF_GRID_MULT
#define ECGPU_REF_GRID_MULT 4
#endif
#ifndef FB_BATCH
#define FB_BATCH (C::NW > 8 ? 16 : 64)
#endif
// occupancy target of the wide fixed-base kernel (A/B switch: 3 = 168 VGPRs, room for the gather prefetch ECGPU_FB_PREFETCH)
// results per lane a wave of the wide fixed-base kernel draws at a time (sched.hpp): 16 results are about one variable-base unit's work
#ifndef FB_CHUNK_UNITS
#define FB_CHUNK_UNITS 16
#endif
#ifndef FB_WIDE_WAVES
#define FB_WIDE_WAVES 4
#endif

// occupancy target (waves per SIMD) of the constant-time fixed-base kernel fb::mul_ct_kernel.  With the Jacobian addition (fixedbase_ct.hpp)
// the live set fits 128 VGPRs on the 8-word curves (k256: 10 spilled) and 168 on P-384 (15 spilled); signing per 2^20 at these against the
// round-2 targets 3 / 4 / 2: k256 4.66 vs 4.72 ms, p384 16.4 vs 17.1 ms (profiles/r03_ab_measurements.txt)
#ifndef FBCT_WAVES
#define FBCT_WAVES(C) (C::NW > 8 ? 3 : 4)
#endif
// most workgroups launched per resident one by the constant-time fixed-base kernel (ecgpu_grid_oversubscribed; profiles/r04_ab_measurements.txt, set nine)
#ifndef FBCT_GRID_MULT
#define FBCT_GRID_MULT 4
#endif

namespace ecgpu {

template <class C>
struct CurveOps {
  static int field_op(ecgpu_ctx* c, int op, const u32* a, const u32* b, u32* o, size_t n) {
    const unsigned g = ecgpu_grid_for(c, n, 8);
    switch (op) {
#define FOP(OPC) case OPC: hipLaunchKernelGGL((field_op_kernel<C, OPC>), dim3(g), dim3(256), 0, c->stream, a, b, o, n); break;
      FOP(FE_MUL) FOP(FE_SQR) FOP(FE_ADD) FOP(FE_SUB) FOP(FE_NEG) FOP(FE_INV) FOP(FE_SQRT)
#undef FOP
      default: return ecgpu_set_err(c, ECGPU_ERR_ARG, "unknown field op %d", op);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int point_op(ecgpu_ctx* c, int op, const u32* p, const u32* q, u32* o, size_t n) {
    const unsigned g = ecgpu_grid_for(c, n, 8);
    switch (op) {
      case PT_ADD: hipLaunchKernelGGL((point_op_kernel<C, PT_ADD>), dim3(g), dim3(256), 0, c->stream, p, q, o, n); break;
      case PT_ADD_MIXED: hipLaunchKernelGGL((point_op_kernel<C, PT_ADD_MIXED>), dim3(g), dim3(256), 0, c->stream, p, q, o, n); break;
      case PT_DOUBLE: hipLaunchKernelGGL((point_op_kernel<C, PT_DOUBLE>), dim3(g), dim3(256), 0, c->stream, p, q, o, n); break;
      default: return ecgpu_set_err(c, ECGPU_ERR_ARG, "unknown point op %d", op);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int normalize(ecgpu_ctx* c, const u32* p, u32* out_xy, uint8_t* out_inf, size_t n) {
    hipLaunchKernelGGL((normalize_kernel<C, 16>), dim3(ecgpu_grid_for((const ecgpu_ctx*)c, (n + 15) / 16, 8)), dim3(256), 0, c->stream, p, out_xy, out_inf, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int ensure_gen_table(ecgpu_ctx* c) {
    if (c->gen_table[C::ID]) return 0;
    void* t = nullptr;
    HIPCHK(c, hipMalloc(&t, sizeof(typename C::Pt) * C::GEN_TABLE_PTS));
    hipLaunchKernelGGL((gen_table_kernel<C>), dim3(1), dim3(64), 0, c->stream, (typename C::Pt*)t);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));     // later calls may run on another stream (ecgpu_set_stream)
    c->gen_table[C::ID] = t;
    return 0;
  }
  // fixed-base table T[j][d-1] = d 2^(8j) G (fixedbase.hpp), built once per context
  static constexpr int FB_NOMEM = 100;          // internal: a table did not fit, the caller steps down a width
  static void fb_account(ecgpu_ctx* c, size_t bytes, int window) {
    c->fb_bytes[C::ID] += bytes;
    if (window > c->fb_widest[C::ID]) c->fb_widest[C::ID] = window;
  }
  static int ensure_fb_table(ecgpu_ctx* c) {
    if (c->fb_table[C::ID]) return 0;
    const int total = fb::nwin<C>() * fb::ENTRIES;
    struct Tmp {                       // released on every exit path
      void* p = nullptr;
      ~Tmp() { if (p) (void)hipFree(p); }
    } ttmp, ttab;
    HIPCHK(c, hipMalloc(&ttmp.p, sizeof(Jac<C>) * total));
    HIPCHK(c, hipMalloc(&ttab.p, sizeof(AffEntry<C>) * total));
    void *tmp = ttmp.p, *tab = ttab.p;
    hipLaunchKernelGGL((fb::table_jac_kernel<C>), dim3((fb::nwin<C>() + 63) / 64), dim3(64), 0, c->stream, (Jac<C>*)tmp);
    hipLaunchKernelGGL((fb::table_affine_kernel<C>), dim3((total + 255) / 256), dim3(256), 0, c->stream, (const Jac<C>*)tmp, (AffEntry<C>*)tab, total);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->fb_table[C::ID] = tab;
    ttab.p = nullptr;                  // owned by the context from here on
    fb_account(c, sizeof(AffEntry<C>) * total, fb::W);
    return 0;
  }
  // wide tables TW[j][d-1] = d 2^(WB j) G, 2^(WB-1) entries per window, built by multiplying the scalars d 2^(WB j)
  // with the 8-bit-window kernel (WB <= 20) or the 20-bit-window kernel (WB > 20); used for large batches, where the
  // build cost (one pass over 0.56 M points for WB = 16, 6.8 M for WB = 20, 92 M for WB = 24, 336 M for WB = 26) is
  // amortised.  The 288 GB of HBM are what makes the last two possible: every window bit less is one mixed addition of
  // twelve saved per result, paid for with table bytes that a lane reads one 64-byte entry at a time.
  template <int WB>
  static int ensure_fb_wide_table(ecgpu_ctx* c, void** slot) {
    if (*slot) return 0;
    int rc = ensure_fb_table(c);
    if (rc) return rc;
    if constexpr (WB > 20) {
      if ((rc = ensure_fb_wide_table<20>(c, &c->fb20_table[C::ID]))) return rc;       // FB_NOMEM passes through
    }
    const size_t total = (size_t)fb::nwin_wide<C, WB>() * fb::wide_entries<WB>();
    const size_t chunk = total < ((size_t)1 << 24) ? total : ((size_t)1 << 24);     // entries per pass: at most 1.5 GB (2.3 GB for p384) of scratch
    struct Tmp {                       // released on every exit path
      void* p = nullptr;
      ~Tmp() { if (p) (void)hipFree(p); }
    } tks, txy, ttab;
    // a table that does not fit - the context's budget or the device's memory - is not an error: the caller falls back to
    // the next narrower width
    // (the budget is about the OPTIONAL wide tables: the 5-bit table of the constant-time kernel behind signing and key generation is
    // 53-80 KB and mandatory, like the 8-bit base table - neither is ever refused on the budget's account, though both count towards it)
    if (WB != fb::CT_WB && c->opt[ECGPU_OPT_FB_MEMORY_BUDGET] > 0 &&
        c->fb_bytes[C::ID] + total * sizeof(AffEntry<C>) > (size_t)c->opt[ECGPU_OPT_FB_MEMORY_BUDGET])
      return FB_NOMEM;
    {
      hipError_t e = hipMalloc(&tks.p, chunk * C::NB);
      if (e == hipSuccess) e = hipMalloc(&txy.p, chunk * 2 * C::NB);
      if (e == hipSuccess) e = hipMalloc(&ttab.p, total * sizeof(AffEntry<C>));
      if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();         // not sticky: the next launch check must not report it
        return FB_NOMEM;
      }
      if (e != hipSuccess) return ecgpu_set_err(c, ECGPU_ERR_RUNTIME, "generator table (%d-bit windows): %s", WB, hipGetErrorString(e));
    }
    void *ks = tks.p, *xy = txy.p;
    AffEntry<C>* tab = (AffEntry<C>*)ttab.p;
    for (size_t e0 = 0; e0 < total; e0 += chunk) {
      const size_t cnt = total - e0 < chunk ? total - e0 : chunk;
      hipLaunchKernelGGL((fb::table_scalars_kernel<C, WB>), dim3(ecgpu_grid_for(c, cnt, 8)), dim3(256), 0, c->stream, (u32*)ks, e0, cnt);
      if constexpr (WB > 20)
        hipLaunchKernelGGL((fb::mul_wide_kernel<C, 20, FB_BATCH, 4>), dim3(ecgpu_grid_for(c, cnt, 4)), dim3(256), 0, c->stream, (const u32*)ks,
                           (const AffEntry<C>*)c->fb20_table[C::ID], (u32*)xy, FMT_AFFINE, (uint8_t*)nullptr, cnt, WaveSched{nullptr, 0, 0, 0, 0});
      else
        hipLaunchKernelGGL((fb::mul_kernel<C, 16, 4>), dim3(ecgpu_grid_for(c, cnt, 4)), dim3(256), 0, c->stream, (const u32*)ks,
                           (const AffEntry<C>*)c->fb_table[C::ID], (u32*)xy, FMT_AFFINE, (uint8_t*)nullptr, cnt);
      hipLaunchKernelGGL((fb::table_from_bytes_kernel<C>), dim3(ecgpu_grid_for(c, cnt, 8)), dim3(256), 0, c->stream, (const u32*)xy, tab + e0, cnt);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *slot = tab;
    ttab.p = nullptr;                  // the context owns the table now; the two scratch buffers go with this scope
    fb_account(c, total * sizeof(AffEntry<C>), WB);
    return 0;
  }
  template <int WB>
  static int mul_gen_wide(ecgpu_ctx* c, void** slot, const u32* sc, u32* out, int out_fmt, uint8_t* out_inf, size_t n) {
    int rc = ensure_fb_wide_table<WB>(c, slot);
    if (rc) return rc;
    unsigned long long* ctr = ecgpu_sched_counter(c);
    if (!ctr) return ECGPU_ERR_RUNTIME;
    const unsigned grid = ecgpu_grid_for(c, n, FB_WIDE_WAVES);
    hipLaunchKernelGGL((fb::mul_wide_kernel<C, WB, FB_BATCH, FB_WIDE_WAVES>), dim3(grid), dim3(256), 0, c->stream, sc,
                       (const AffEntry<C>*)*slot, out, out_fmt, out_inf, n, WaveSched{ctr, (unsigned long long)n, grid * 4u, (unsigned)FB_CHUNK_UNITS, 1u});
    HIPCHK(c, hipGetLastError());
    return 1;
  }
  static int mul_gen_fast(ecgpu_ctx* c, const u32* sc, u32* out, int out_fmt, uint8_t* out_inf, size_t n) {
    // ECGPU_OPT_FB_WINDOW pins one table width (measurements, small-memory processes); ECGPU_OPT_FB_MAX_WINDOW caps what
    // the size rule may pick (default 26: 21.5 GB per 256-bit curve)
    const int forced = (int)c->opt[ECGPU_OPT_FB_WINDOW];
    // (P-384 stops at 24 bits: its 26-bit table would be 48 GB and measured 45.2 ms against 46.4 ms per 2^24 results)
    int wb = (n >= ((size_t)1 << 24) && C::NW <= 8) ? 26 : n >= ((size_t)1 << 23) ? 24 : n >= ((size_t)1 << 21) ? 20 : n >= ((size_t)1 << 18) ? 16 : 8;
    if (forced) wb = forced;
    // A table that cannot be allocated is not an error: step down to the next narrower one and remember the width that
    // fitted as this context's cap, so that later calls do not try (and fail) a 21 GB allocation each time.
    for (;;) {
      const int cap = (int)c->opt[ECGPU_OPT_FB_MAX_WINDOW];
      if (wb > cap) wb = cap;
      int rc;
      if (wb >= 26) rc = mul_gen_wide<26>(c, &c->fb26_table[C::ID], sc, out, out_fmt, out_inf, n);
      else if (wb >= 24) rc = mul_gen_wide<24>(c, &c->fb24_table[C::ID], sc, out, out_fmt, out_inf, n);
      else if (wb >= 20) rc = mul_gen_wide<20>(c, &c->fb20_table[C::ID], sc, out, out_fmt, out_inf, n);
      else if (wb >= 16) rc = mul_gen_wide<16>(c, &c->fb16_table[C::ID], sc, out, out_fmt, out_inf, n);
      else break;
      if (rc != FB_NOMEM) return rc;
      wb = wb >= 26 ? 24 : wb >= 24 ? 20 : wb >= 20 ? 16 : 8;
      c->opt[ECGPU_OPT_FB_MAX_WINDOW] = wb;
    }
    int rc = ensure_fb_table(c);
    if (rc) return rc;
    hipLaunchKernelGGL((fb::mul_kernel<C, 16, 4>), dim3(ecgpu_grid_for(c, n, 4)), dim3(256), 0, c->stream, sc,
                       (const AffEntry<C>*)c->fb_table[C::ID], out, out_fmt, out_inf, n);
    HIPCHK(c, hipGetLastError());
    return 1;
  }
  // k G for secret scalars: constant-time fixed-base kernel (fixedbase.hpp), one inversion per 8 results
  static int mul_gen_ct(ecgpu_ctx* c, const u32* sc, u32* out, int out_fmt, uint8_t* out_inf, size_t n) {
    int rc = ensure_fb_wide_table<fb::CT_WB>(c, &c->fbct_table[C::ID]);
    if (rc == FB_NOMEM) return ecgpu_set_err(c, ECGPU_ERR_RUNTIME, "hipMalloc failed (out of device memory) for the 5-bit generator table of the constant-time kernel");
    if (rc) return rc;
    if constexpr (C::ID == 0) {          // secp256k1: the kernel lives in the branch-free translation unit (ops_k256_ct.hip)
      return ecgpuint_k256_mul_gen_ct(c, sc, c->fbct_table[C::ID], out, out_fmt, out_inf, n);
    } else {
      constexpr int WAVES = FBCT_WAVES(C);
      hipLaunchKernelGGL((fb::mul_ct_kernel<C, 8, WAVES>), dim3(ecgpu_grid_oversubscribed(c, n, WAVES, 8, FBCT_GRID_MULT)), dim3(256), 0, c->stream, sc, (const AffEntry<C>*)c->fbct_table[C::ID], out,
                         out_fmt, out_inf, n);
      HIPCHK(c, hipGetLastError());
      return 0;
    }
  }
  // curve-specific throughput kernels hook in here (specialised in ops_*.hip); returns 1 if it launched
  static int lincomb_fast(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t terms, u32* out, int out_fmt,
                          uint8_t* out_inf, size_t n);
  // constant-time variable base for secret scalars (varbase_ct.hpp; specialised in ops_*.hip); returns 1 if it launched
  static int mul_ct(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, u32* out, int out_fmt, uint8_t* out_inf, size_t n);
  static int lincomb(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t terms, u32* out, int out_fmt, uint8_t* out_inf,
                     size_t n, unsigned flags) {
    if ((flags & ECGPU_SECRET_SCALARS) && !(flags & ECGPU_EXACT_REFERENCE)) {
      if (!pts && terms == 1) return mul_gen_ct(c, sc, out, out_fmt, out_inf, n);
      if (pts && terms == 1) {
        int rc = mul_ct(c, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
        if (rc < 0) return rc;
        if (rc == 1) return 0;
      }
      flags |= ECGPU_EXACT_REFERENCE;              // no dedicated kernel: the reference schedule is constant-time as well
    }
    if (!(flags & ECGPU_EXACT_REFERENCE)) {
      int rc = lincomb_fast(c, sc, pts, pt_fmt, terms, out, out_fmt, out_inf, n);
      if (rc < 0) return rc;
      if (rc == 1) return 0;
    }
    // (the reference-schedule kernels walk their units one at a time with a grid stride: four times the workgroups the chip holds, so that the hardware
    // hands the waiting ones out as the favoured waves of a SIMD leave - sched.hpp)
    const unsigned g = ecgpu_grid_for(c, n, 4 * ECGPU_REF_GRID_MULT);
    if (!pts && terms != 1) return ecgpu_set_err(c, ECGPU_ERR_ARG, "generator multiplication takes one term");
    if (terms > 2 && terms <= 1024 && !(flags & ECGPU_EXACT_REFERENCE)) {
      if (c->opt[ECGPU_OPT_LINCOMB_TERM_BY_TERM]) {
        hipLaunchKernelGGL((lincomb_sum_kernel<C>), dim3(g), dim3(256), 0, c->stream, sc, pts, pt_fmt, (int)terms, out, out_fmt, out_inf, n);
        HIPCHK(c, hipGetLastError());
        return 0;
      }
      return lincomb_straus(c, sc, pts, pt_fmt, terms, out, out_fmt, out_inf, n);
    }
    if (terms > 1024 || (terms > 2 && C::ID != 0))
      return ecgpu_set_err(c, ECGPU_ERR_UNSUPPORTED,
                           "lincomb_batch: at most 1024 terms per combination (use ecgpu_msm for one large sum); ECGPU_EXACT_REFERENCE with more than 2 terms "
                           "exists for secp256k1 only (the primeorder curves have no in-tree lincomb over slices to be exact to)");
    // the reference schedules (exact X, Y, Z; constant-time table scans)
    if (!pts) {
      int rc = ensure_gen_table(c);
      if (rc) return rc;
    }
    if constexpr (C::ID == 0) {            // secp256k1: in the branch-free translation unit (ops_k256_ct.hip)
      return ecgpuint_k256_reference(c, sc, pts, pt_fmt, terms, c->gen_table[C::ID], out, out_fmt, out_inf, n);
    } else {
      if (!pts) {
        hipLaunchKernelGGL((mul_gen_ref_kernel<C>), dim3(g), dim3(256), 0, c->stream, sc, (const typename C::Pt*)c->gen_table[C::ID], out,
                           out_fmt, out_inf, n);
      } else if (terms == 1) {
        hipLaunchKernelGGL((lincomb_ref_kernel<C, 1>), dim3(g), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
      } else {
        hipLaunchKernelGGL((lincomb_ref_kernel<C, 2>), dim3(g), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
      }
      HIPCHK(c, hipGetLastError());
      return 0;
    }
  }
  // the grow-only per-lane workspace of the variable-base kernels
  static int tab_reserve(ecgpu_ctx* c, size_t need) {
    if (need <= c->tab_ws_cap) return 0;
    if (c->tab_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->tab_ws)); c->tab_ws = nullptr; c->tab_ws_cap = 0; }
    HIPCHK(c, hipMalloc(&c->tab_ws, need));
    c->tab_ws_cap = need;
    return 0;
  }
  // 3 .. 1024 terms per combination, throughput schedule (straus.hpp): groups of up to 16 terms share the doublings of one window
  // loop over per-term affine tables; a second small kernel adds the groups' partial sums and writes the outputs
  static int lincomb_straus(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t terms, u32* out, int out_fmt, uint8_t* out_inf, size_t n) {
    // waves per SIMD: secp256k1 runs 3 (168 VGPRs: 22 spilled instead of 119 at 4; -3 .. -5 % at 3 .. 64 terms), P-256 measures equal and
    // P-384 was not timed at 3: both keep 4 (profiles/r04_ab_measurements.txt, set twelve)
    constexpr int WAVES = C::ID == 0 ? 3 : 4;
    int g, gpc;
    straus::plan(n, terms, resident_lanes(c, WAVES), &g, &gpc);
    const size_t items = n * (size_t)gpc, upp = (size_t)(straus::SLOTS / g);
    const unsigned grid = ecgpu_grid_for(c, (items + upp - 1) / upp, WAVES);
    int rc = tab_reserve(c, (size_t)grid * 256 * sizeof(straus::LaneWs<C>));
    if (rc) return rc;
    if ((rc = ecdsa_reserve(c, al256(items * 3 * C::NW * sizeof(u32))))) return rc;
    u32* partial = (u32*)c->ecdsa_ws;
    unsigned long long* ctr = ecgpu_sched_counter(c);
    if (!ctr) return ECGPU_ERR_RUNTIME;
    hipLaunchKernelGGL((straus::lincomb_kernel<C, WAVES>), dim3(grid), dim3(256), 0, c->stream, sc, pts, pt_fmt, (int)terms, g, gpc, items,
                       (straus::LaneWs<C>*)c->tab_ws, partial, WaveSched{ctr, (unsigned long long)items, grid * 4u, (unsigned)upp, 0u});
    hipLaunchKernelGGL((straus::fold_kernel<C>), dim3(ecgpu_grid_for(c, (n + 15) / 16, 8)), dim3(256), 0, c->stream, (const u32*)partial, gpc, out, out_fmt, out_inf, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int msm(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t n, u32* out, int out_fmt);
  // resident lanes at `waves` workgroups of 256 per CU (what ecgpu_grid_for caps a grid at)
  static size_t resident_lanes(const ecgpu_ctx* c, int waves) { return (size_t)c->num_cus * (size_t)waves * 256; }
  // units per whole pass of the variable-base kernels (specialised in ops_*.hip, where their sub-batch sizes are)
  static size_t pass_units_points(const ecgpu_ctx* c, size_t terms, unsigned flags);
  static size_t pass_units(const ecgpu_ctx* c, int has_points, size_t terms, unsigned flags) {
    if (has_points) return pass_units_points(c, terms, flags);
    if ((flags & ECGPU_SECRET_SCALARS) && !(flags & ECGPU_EXACT_REFERENCE)) return resident_lanes(c, FBCT_WAVES(C)) * 8;   // fb::mul_ct_kernel<C, 8, ..>
    if (flags & ECGPU_EXACT_REFERENCE) return resident_lanes(c, 4);
    return resident_lanes(c, 4) * (size_t)(FB_BATCH);                                                                    // fb::mul_wide_kernel<C, .., FB_BATCH, 4>
  }
  static int point_eq(ecgpu_ctx* c, const u32* p, const u32* q, uint8_t* eq, size_t n) {
    hipLaunchKernelGGL((point_eq_kernel<C>), dim3(ecgpu_grid_for(c, n, 8)), dim3(256), 0, c->stream, p, q, eq, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int validate_scalars(ecgpu_ctx* c, const u32* sc, uint8_t* ok, size_t n, size_t terms) {
    hipLaunchKernelGGL((validate_scalars_kernel<C>), dim3(ecgpu_grid_for(c, n, 8)), dim3(256), 0, c->stream, sc, ok, n, terms);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int validate_points(ecgpu_ctx* c, const u32* xy, uint8_t* ok, size_t n) {
    hipLaunchKernelGGL((validate_points_kernel<C>), dim3(ecgpu_grid_for(c, n, 8)), dim3(256), 0, c->stream, xy, ok, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int decompress(ecgpu_ctx* c, const u32* x, const uint8_t* odd, u32* out_xy, uint8_t* ok, size_t n) {
    hipLaunchKernelGGL((decompress_kernel<C>), dim3(ecgpu_grid_for(c, n, 8)), dim3(256), 0, c->stream, x, odd, out_xy, ok, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int synth_scalars(ecgpu_ctx* c, uint64_t seed, uint64_t first, u32* out, size_t n) {
    hipLaunchKernelGGL((synth_scalars_kernel<C>), dim3(ecgpu_grid_for(c, n, 8)), dim3(256), 0, c->stream, (u64)seed, (u64)first, out, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int synth_points(ecgpu_ctx* c, uint64_t seed, uint64_t first, u32* out, size_t n) {
    hipLaunchKernelGGL((synth_points_kernel<C>), dim3(ecgpu_grid_for(c, n, 8)), dim3(256), 0, c->stream, (u64)seed, (u64)first, out, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  // GroupEncoding::to_bytes / ToEncodedPoint of affine or projective points (projective input is batch-normalised first)
  static int sec1_to(ecgpu_ctx* c, const u32* pts, int pt_fmt, int uncompressed, uint8_t* out, size_t n) {
    const u32* xy = pts;
    const uint8_t* inf = nullptr;
    if (pt_fmt == FMT_PROJECTIVE) {
      const size_t sz_p = al256(n * 2 * C::NB);
      int rc = ecdsa_reserve(c, sz_p + al256(n));
      if (rc) return rc;
      u32* t = (u32*)c->ecdsa_ws;
      uint8_t* ti = (uint8_t*)c->ecdsa_ws + sz_p;
      if ((rc = normalize(c, pts, t, ti, n))) return rc;
      xy = t; inf = ti;
    }
    hipLaunchKernelGGL((sec1::to_bytes_kernel<C>), dim3(ecgpu_grid_for(c, n, 8)), dim3(256), 0, c->stream, xy, inf, out, n, uncompressed);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int to_bytes(ecgpu_ctx* c, const u32* pts, int pt_fmt, uint8_t* out, size_t n) { return sec1_to(c, pts, pt_fmt, 0, out, n); }
  static int sec1_encode(ecgpu_ctx* c, const u32* pts, int pt_fmt, int compress, uint8_t* out, size_t n) { return sec1_to(c, pts, pt_fmt, compress ? 0 : 1, out, n); }
  static int sec1_decode(ecgpu_ctx* c, const uint8_t* in, size_t record_bytes, u32* out_xy, uint8_t* ok, size_t n) {
    hipLaunchKernelGGL((sec1::from_bytes_kernel<C>), dim3(ecgpu_grid_for(c, n, 8)), dim3(256), 0, c->stream, in, out_xy, ok, n, (int)record_bytes);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int from_bytes(ecgpu_ctx* c, const uint8_t* in, u32* out_xy, uint8_t* ok, size_t n) { return sec1_decode(c, in, 1 + C::NB, out_xy, ok, n); }
  // ECDSA pipelines (ecdsa_kernels.hpp): the scalar multiplications run on the throughput kernels above
  static int ecdsa_reserve(ecgpu_ctx* c, size_t need) {
    if (need <= c->ecdsa_ws_cap) return 0;
    if (c->ecdsa_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->ecdsa_ws)); c->ecdsa_ws = nullptr; c->ecdsa_ws_cap = 0; }
    HIPCHK(c, hipMalloc(&c->ecdsa_ws, need));
    c->ecdsa_ws_cap = need;
    return 0;
  }
  static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
  static int ecdsa_verify(ecgpu_ctx* c, const u32* z, const u32* sig, const u32* q, uint8_t* ok, size_t n, unsigned flags) {
    const size_t sz_s = al256(n * C::NB), sz_p = al256(n * 2 * C::NB), sz_f = al256(n);
    int rc = ecdsa_reserve(c, 2 * sz_s + 2 * sz_p + 2 * sz_f);
    if (rc) return rc;
    char* p = (char*)c->ecdsa_ws;
    u32* u1 = (u32*)p; p += sz_s;
    u32* u2 = (u32*)p; p += sz_s;
    u32* a = (u32*)p; p += sz_p;
    u32* b = (u32*)p; p += sz_p;
    uint8_t* a_inf = (uint8_t*)p; p += sz_f;
    uint8_t* b_inf = (uint8_t*)p;
    hipLaunchKernelGGL((ecdsa::verify_prep_kernel<C, 16>), dim3(ecgpu_grid_for(c, (n + 15) / 16, 4)), dim3(256), 0, c->stream, z, sig, q, u1, u2,
                       ok, n, flags);
    HIPCHK(c, hipGetLastError());
    if ((rc = lincomb(c, u1, nullptr, FMT_AFFINE, 1, a, FMT_AFFINE, a_inf, n, 0))) return rc;
    if ((rc = lincomb(c, u2, q, FMT_AFFINE, 1, b, FMT_AFFINE, b_inf, n, 0))) return rc;
    hipLaunchKernelGGL((ecdsa::verify_check_kernel<C>), dim3(ecgpu_grid_for(c, n, 8)), dim3(256), 0, c->stream, (const u32*)a,
                       (const uint8_t*)a_inf, (const u32*)b, (const uint8_t*)b_inf, sig, ok, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int h2c_map(ecgpu_ctx* c, const u32* u, int count, u32* out_xy, uint8_t* out_inf, size_t n) {
    hipLaunchKernelGGL((h2c::map_kernel<C>), dim3(ecgpu_grid_for(c, n, 4)), dim3(256), 0, c->stream, u, count, out_xy, out_inf, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int ecdsa_recover(ecgpu_ctx* c, const u32* z, const u32* sig, const uint8_t* recid, u32* out_xy, uint8_t* ok, size_t n, unsigned flags) {
    const size_t sz_s = al256(n * C::NB), sz_p = al256(n * 2 * C::NB), sz_f = al256(n);
    int rc = ecdsa_reserve(c, 2 * sz_s + 3 * sz_p + 2 * sz_f);
    if (rc) return rc;
    char* p = (char*)c->ecdsa_ws;
    u32* u1 = (u32*)p; p += sz_s;
    u32* u2 = (u32*)p; p += sz_s;
    u32* r_xy = (u32*)p; p += sz_p;
    u32* a = (u32*)p; p += sz_p;
    u32* b = (u32*)p; p += sz_p;
    uint8_t* a_inf = (uint8_t*)p; p += sz_f;
    uint8_t* b_inf = (uint8_t*)p;
    hipLaunchKernelGGL((ecdsa::recover_prep_kernel<C, 16>), dim3(ecgpu_grid_for(c, (n + 15) / 16, 4)), dim3(256), 0, c->stream, z, sig, recid, r_xy, u1,
                       u2, ok, n, flags);
    HIPCHK(c, hipGetLastError());
    if ((rc = lincomb(c, u1, nullptr, FMT_AFFINE, 1, a, FMT_AFFINE, a_inf, n, 0))) return rc;
    if ((rc = lincomb(c, u2, r_xy, FMT_AFFINE, 1, b, FMT_AFFINE, b_inf, n, 0))) return rc;
    hipLaunchKernelGGL((ecdsa::recover_finish_kernel<C, 16>), dim3(ecgpu_grid_for(c, (n + 15) / 16, 8)), dim3(256), 0, c->stream, (const u32*)a,
                       (const uint8_t*)a_inf, (const u32*)b, (const uint8_t*)b_inf, out_xy, ok, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  // BIP340 verification (schnorr_kernels.hpp), secp256k1 only
  static int schnorr_verify(ecgpu_ctx* c, const u32* px, const u32* sig, const u32* e, uint8_t* ok, size_t n) {
    if constexpr (C::ID != 0) {
      return ecgpu_set_err(c, ECGPU_ERR_UNSUPPORTED, "ecgpu_schnorr_verify_batch: BIP340 is defined over secp256k1 only");
    } else {
      const size_t sz_s = al256(n * C::NB), sz_p = al256(n * 2 * C::NB), sz_f = al256(n);
      int rc = ecdsa_reserve(c, 2 * sz_s + 3 * sz_p + 2 * sz_f);
      if (rc) return rc;
      char* p = (char*)c->ecdsa_ws;
      u32* u1 = (u32*)p; p += sz_s;
      u32* u2 = (u32*)p; p += sz_s;
      u32* key = (u32*)p; p += sz_p;
      u32* a = (u32*)p; p += sz_p;
      u32* b = (u32*)p; p += sz_p;
      uint8_t* a_inf = (uint8_t*)p; p += sz_f;
      uint8_t* b_inf = (uint8_t*)p;
      hipLaunchKernelGGL((schnorr::verify_prep_kernel<0>), dim3(ecgpu_grid_for(c, n, 8)), dim3(256), 0, c->stream, px, sig, e, key, u1, u2, ok, n);
      HIPCHK(c, hipGetLastError());
      if ((rc = lincomb(c, u1, nullptr, FMT_AFFINE, 1, a, FMT_AFFINE, a_inf, n, 0))) return rc;
      if ((rc = lincomb(c, u2, key, FMT_AFFINE, 1, b, FMT_AFFINE, b_inf, n, 0))) return rc;
      hipLaunchKernelGGL((schnorr::verify_check_kernel<16>), dim3(ecgpu_grid_for(c, (n + 15) / 16, 8)), dim3(256), 0, c->stream, (const u32*)a,
                         (const uint8_t*)a_inf, (const u32*)b, (const uint8_t*)b_inf, sig, ok, n);
      HIPCHK(c, hipGetLastError());
      return 0;
    }
  }
  // ECDH (ecdh_kernels.hpp): input checks, the secret-scalar multiplication (constant-time kernel where the curve has
  // one, the reference schedule otherwise), x of the product
  static int ecdh(ecgpu_ctx* c, const u32* d, const u32* q, u32* shared_x, uint8_t* ok, size_t n) {
    const size_t sz_p = al256(n * 2 * C::NB);
    int rc = ecdsa_reserve(c, 2 * sz_p);
    if (rc) return rc;
    u32* prod = (u32*)c->ecdsa_ws;
    u32* q_sane = (u32*)((char*)c->ecdsa_ws + sz_p);
    // the products are secrets of the same rank as the shared values handed back: they do not stay in the workspace, whichever way
    // this function is left (the constant-time kernels clear what they park in the table workspace themselves)
    struct ProdWipe {
      ecgpu_ctx* c; void* p; size_t b;
      ~ProdWipe() { (void)hipMemsetAsync(p, 0, b, c->stream); }
    } prod_wipe{c, prod, n * 2 * C::NB};
    hipLaunchKernelGGL((ecdh::prep_kernel<C>), dim3(ecgpu_grid_for(c, n, 8)), dim3(256), 0, c->stream, d, q, q_sane, ok, n);
    HIPCHK(c, hipGetLastError());
    if ((rc = lincomb(c, d, q_sane, FMT_AFFINE, 1, prod, FMT_AFFINE, nullptr, n, (unsigned)ECGPU_SECRET_SCALARS))) return rc;
    hipLaunchKernelGGL((ecdh::finish_kernel<C>), dim3(ecgpu_grid_for(c, n, 8)), dim3(256), 0, c->stream, (const u32*)prod, (const uint8_t*)ok, shared_x, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static int ecdsa_sign(ecgpu_ctx* c, const u32* d, const u32* k, const u32* z, u32* sig, uint8_t* recid, uint8_t* ok, size_t n,
                        unsigned flags) {
    const size_t sz_p = al256(n * 2 * C::NB), sz_f = al256(n);
    int rc = ecdsa_reserve(c, sz_p + sz_f);
    if (rc) return rc;
    u32* r_xy = (u32*)c->ecdsa_ws;
    uint8_t* r_inf = (uint8_t*)c->ecdsa_ws + sz_p;
    // The nonce is secret: k G runs on the constant-time fixed-base kernel (every table entry read, one masked addition per window,
    // no digit-dependent branch or address) - or, with ECGPU_EXACT_REFERENCE, on the reference's own mul_by_generator
    // schedule, which is constant-time as well - unless the caller declares the scalars public.
    if (flags & ECGPU_PUBLIC_SCALARS) rc = lincomb(c, k, nullptr, FMT_AFFINE, 1, r_xy, FMT_AFFINE, r_inf, n, 0u);
    else if (flags & ECGPU_EXACT_REFERENCE) rc = lincomb(c, k, nullptr, FMT_AFFINE, 1, r_xy, FMT_AFFINE, r_inf, n, (unsigned)ECGPU_EXACT_REFERENCE);
    else rc = mul_gen_ct(c, k, r_xy, FMT_AFFINE, r_inf, n);
    if (rc) return rc;
    hipLaunchKernelGGL((ecdsa::sign_finish_kernel<C, 16>), dim3(ecgpu_grid_for(c, (n + 15) / 16, 4)), dim3(256), 0, c->stream, d, k, z,
                       (const u32*)r_xy, (const uint8_t*)r_inf, sig, recid, ok, n, flags);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  static const ecgpu_curve_ops* table() {
    static const ecgpu_curve_ops t = {field_op, point_op, point_eq, normalize, lincomb, msm, validate_scalars, validate_points,
                                      decompress, synth_scalars, synth_points, to_bytes, from_bytes, sec1_encode, sec1_decode, ecdsa_verify, h2c_map, ecdsa_recover, schnorr_verify, ecdsa_sign, ecdh,
                                      pass_units};
    return &t;
  }
};

}  // namespace ecgpu
