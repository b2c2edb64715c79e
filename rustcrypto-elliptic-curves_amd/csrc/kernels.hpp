// Batch kernels, written once over the curve traits (traits.hpp).
//
// HBM layout: the wire format itself (canonical big-endian bytes, array of structures).  One
// element per lane; a lane reads its NB-byte operands with dword loads and byte-swaps them in
// registers.  The hot kernels do >10^5 VALU instructions per 100-250 bytes moved, so the I/O
// pattern is irrelevant to throughput (algorithmic HBM traffic is <0.3 % of the roofline, see
// DESIGN.md); what matters is VGPR pressure and instruction count.
#pragma once
#include "traits.hpp"
#include "mulfast_k256.hpp"
#include "jacobian.hpp"
#include "sched.hpp"

namespace ecgpu {

#define ECGPU_GRID_STRIDE(i, n) \
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

enum { FE_MUL = 0, FE_SQR = 1, FE_ADD = 2, FE_SUB = 3, FE_NEG = 4, FE_INV = 5, FE_SQRT = 6 };
enum { PT_ADD = 0, PT_ADD_MIXED = 1, PT_DOUBLE = 2 };

template <class C>
__device__ __forceinline__ void load_affine(typename C::Af& a, const u32* xy) {
  C::fe_load(a.x, xy);
  C::fe_load(a.y, xy + C::NW);
  u32 z = 0;
#pragma unroll
  for (int i = 0; i < 2 * C::NW; i++) z |= xy[i];
  a.inf = (z == 0) ? 1u : 0u;
}
template <class C>
__device__ __forceinline__ void load_point(typename C::Pt& p, const u32* src, int fmt) {
  if (fmt == FMT_PROJECTIVE) {
    C::fe_load(p.x, src);
    C::fe_load(p.y, src + C::NW);
    C::fe_load(p.z, src + 2 * C::NW);
  } else {
    typename C::Af a;
    load_affine<C>(a, src);
    typename C::Pt id;
    C::pt_identity(id);
    typename C::Fe one;
    C::fe_one(one);
    C::fe_select(p.x, a.inf != 0, id.x, a.x);
    C::fe_select(p.y, a.inf != 0, id.y, a.y);
    C::fe_select(p.z, a.inf != 0, id.z, one);
  }
}
template <class C>
__device__ __forceinline__ void store_projective(u32* dst, const typename C::Pt& p) {
  C::fe_store(dst, p.x);
  C::fe_store(dst + C::NW, p.y);
  C::fe_store(dst + 2 * C::NW, p.z);
}
// projective -> affine with one inversion in this lane (to_affine, k256 projective.rs:73-84)
template <class C>
__device__ __forceinline__ void store_affine_from_projective(u32* dst_xy, uint8_t* dst_inf, const typename C::Pt& p) {
  typename C::Fe zi, x, y, zero;
  const bool inf = C::fe_is_zero(p.z);
  C::fe_inv(zi, p.z);
  C::fe_mul(x, p.x, zi);
  C::fe_mul(y, p.y, zi);
  C::fe_zero(zero);
  C::fe_select(x, inf, zero, x);
  C::fe_select(y, inf, zero, y);
  C::fe_store(dst_xy, x);
  C::fe_store(dst_xy + C::NW, y);
  if (dst_inf) *dst_inf = inf ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
template <class C, int OP>
__global__ void __launch_bounds__(256) field_op_kernel(const u32* a, const u32* b, u32* out, size_t n) {
  ECGPU_GRID_STRIDE(i, n) {
    typename C::Fe x, y, r;
    C::fe_load(x, a + i * C::NW);
    if (OP == FE_MUL || OP == FE_ADD || OP == FE_SUB) C::fe_load(y, b + i * C::NW);
    bool ok = true;
    if (OP == FE_MUL) C::fe_mul(r, x, y);
    if (OP == FE_SQR) C::fe_sqr(r, x);
    if (OP == FE_ADD) C::fe_add(r, x, y);
    if (OP == FE_SUB) C::fe_sub(r, x, y);
    if (OP == FE_NEG) C::fe_neg(r, x);
    if (OP == FE_INV) C::fe_inv(r, x);
    if (OP == FE_SQRT) ok = C::fe_sqrt(r, x);
    u32* o = out + i * C::NW;
    C::fe_store(o, r);
    if (!ok) {
#pragma unroll
      for (int j = 0; j < C::NW; j++) o[j] = 0xFFFFFFFFu;
    }
  }
}

template <class C, int OP>
__global__ void __launch_bounds__(256) point_op_kernel(const u32* p, const u32* q, u32* out, size_t n) {
  ECGPU_GRID_STRIDE(i, n) {
    typename C::Pt a, r;
    load_point<C>(a, p + i * 3 * C::NW, FMT_PROJECTIVE);
    if (OP == PT_ADD) {
      typename C::Pt b;
      load_point<C>(b, q + i * 3 * C::NW, FMT_PROJECTIVE);
      C::pt_add(r, a, b);
    }
    if (OP == PT_ADD_MIXED) {
      typename C::Af b;
      load_affine<C>(b, q + i * 2 * C::NW);
      C::pt_add_mixed(r, a, b);
    }
    if (OP == PT_DOUBLE) C::pt_double(r, a);
    store_projective<C>(out + i * 3 * C::NW, r);
  }
}

// ConstantTimeEq / PartialEq of projective points (k256 projective.rs:421-446: X1 Z2 == X2 Z1 and Y1 Z2 == Y2 Z1;
// primeorder projective.rs:191-198 compares the affine forms - the same relation on valid points).  Two identities
// are equal, an identity and a finite point are not.
template <class C>
__global__ void __launch_bounds__(256) point_eq_kernel(const u32* p, const u32* q, uint8_t* eq, size_t n) {
  ECGPU_GRID_STRIDE(i, n) {
    typename C::Pt a, b;
    load_point<C>(a, p + i * 3 * C::NW, FMT_PROJECTIVE);
    load_point<C>(b, q + i * 3 * C::NW, FMT_PROJECTIVE);
    typename C::Fe l, r, d;
    C::fe_mul(l, a.x, b.z); C::fe_mul(r, b.x, a.z); C::fe_sub(d, l, r);
    const bool ex = C::fe_is_zero(d);
    C::fe_mul(l, a.y, b.z); C::fe_mul(r, b.y, a.z); C::fe_sub(d, l, r);
    eq[i] = (ex && C::fe_is_zero(d)) ? 1 : 0;
  }
}

// BatchNormalize (k256 projective.rs:325-379, primeorder projective.rs:346-413): homogeneous (X:Y:Z) ->
// affine with Montgomery's trick.  Each lane walks BATCH elements (grid stride apart, so loads stay
// coalesced across the wave), multiplies their Z together, inverts once and unwinds: 1 inversion +
// ~5 multiplications per point instead of 1 inversion per point.  Identity inputs (Z = 0) give
// x = y = 0, infinity = 1 without disturbing the batch (same dummy-value trick as the reference, :361-364).
template <class C, int BATCH>
__global__ void __launch_bounds__(256) normalize_kernel(const u32* p, u32* out_xy, uint8_t* out_inf, size_t n) {
  using Fe = typename C::Fe;
  constexpr int NW = C::NW;
  Fe zs[BATCH], pre[BATCH];
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * BATCH) {
    int cnt = 0;
    Fe acc; C::fe_one(acc);
#pragma unroll 1
    for (int b = 0; b < BATCH; b++) {
      const size_t i = base + (size_t)b * T;
      if (i >= n) break;
      Fe z;
      C::fe_load(z, p + i * 3 * NW + 2 * NW);
      zs[b] = z;
      pre[b] = acc;
      if (C::fe_is_zero(z)) C::fe_one(z);
      C::fe_mul(acc, acc, z);
      cnt = b + 1;
    }
    Fe ai;
    C::fe_inv(ai, acc);
#pragma unroll 1
    for (int b = cnt - 1; b >= 0; b--) {
      const size_t i = base + (size_t)b * T;
      Fe z = zs[b], zi, x, y;
      const bool zr = C::fe_is_zero(z);
      if (zr) C::fe_one(z);
      C::fe_mul(zi, ai, pre[b]);
      C::fe_mul(ai, ai, z);
      C::fe_load(x, p + i * 3 * NW);
      C::fe_load(y, p + i * 3 * NW + NW);
      C::fe_mul(x, x, zi);
      C::fe_mul(y, y, zi);
      if (zr) { C::fe_zero(x); C::fe_zero(y); }
      C::fe_store(out_xy + i * 2 * NW, x);
      C::fe_store(out_xy + i * 2 * NW + NW, y);
      if (out_inf) out_inf[i] = zr ? 1 : 0;
    }
  }
}

// Reference-faithful n independent linear combinations of NT terms (NT = 1 is `&P * &k`).
#ifndef ECGPU_REF_WAVES
#define ECGPU_REF_WAVES 2          // occupancy target of the reference-schedule kernels (measured: see DESIGN.md section 4)
#endif
template <class C, int NT>
__global__ void __launch_bounds__(256, ECGPU_REF_WAVES) lincomb_ref_kernel(const u32* scalars, const u32* points, int pt_fmt,
                                                          u32* out, int out_fmt, uint8_t* out_inf, size_t n) {
  typename C::Pt tab[C::REF_TABLE_PTS * NT];
  ECGPU_GRID_STRIDE(i, n) {
    typename C::Pt pts[NT], r;
    u32 ks[NT][C::NW];
    const int pw = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * C::NW;
#pragma unroll 1
    for (int t = 0; t < NT; t++) {
      C::scalar_load(ks[t], scalars + (i * NT + t) * C::NW);
      load_point<C>(pts[t], points + (i * NT + t) * pw, pt_fmt);
    }
    C::template lincomb_ref<NT>(r, pts, ks, tab);
    if (out_fmt == FMT_PROJECTIVE) store_projective<C>(out + i * 3 * C::NW, r);
    else store_affine_from_projective<C>(out + i * 2 * C::NW, out_inf ? out_inf + i : nullptr, r);
  }
}

// n independent linear combinations of `terms` (a run-time count) terms: sum_t k_t P_t as one reference scalar
// multiplication per term folded with the complete addition.  This is the primeorder default
// (LinearCombination: x*k + y*l, primeorder/src/projective.rs:415-420) extended to any length; for k256, whose
// lincomb_ext interleaves the terms over shared doublings (mul.rs:342-393), it is the same group element.
template <class C>
__global__ void __launch_bounds__(256) lincomb_sum_kernel(const u32* scalars, const u32* points, int pt_fmt, int terms, u32* out, int out_fmt,
                                                          uint8_t* out_inf, size_t n) {
  typename C::Pt tab[C::REF_TABLE_PTS];
  ECGPU_GRID_STRIDE(i, n) {
    typename C::Pt acc, p, r;
    C::pt_identity(acc);
    const int pw = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * C::NW;
#pragma unroll 1
    for (int t = 0; t < terms; t++) {
      u32 k[C::NW];
      C::scalar_load(k, scalars + (i * terms + t) * C::NW);
      load_point<C>(p, points + (i * terms + t) * pw, pt_fmt);
      C::mul_ref(r, p, k, tab);
      C::pt_add(acc, acc, r);
    }
    if (out_fmt == FMT_PROJECTIVE) store_projective<C>(out + i * 3 * C::NW, acc);
    else store_affine_from_projective<C>(out + i * 2 * C::NW, out_inf ? out_inf + i : nullptr, acc);
  }
}

// secp256k1, exact-(X, Y, Z) contract for combinations of any length: the reference's own interleaved schedule (mul_k256.hpp:
// lincomb_ref_term / lincomb_ref_run) with the per-term tables and digits in a global per-lane scratch (`terms` x (16 points + 10 words)).
template <class C>
__global__ void __launch_bounds__(256, ECGPU_REF_WAVES) k256_lincomb_ref_n_kernel(const u32* scalars, const u32* points, int pt_fmt, int terms, u32* out, int out_fmt,
                                                                                uint8_t* out_inf, size_t n, PtK256* tab_all, u32* dig_all) {
  static_assert(C::ID == 0, "secp256k1 only");
  const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  PtK256* tab = tab_all + lane * (size_t)terms * 16;
  u32* dig = dig_all + lane * (size_t)terms * 10;
  const int pw = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * C::NW;
  ECGPU_GRID_STRIDE(i, n) {
#pragma unroll 1
    for (int t = 0; t < terms; t++) {
      u32 k[8];
      PtK256 p;
      C::scalar_load(k, scalars + (i * terms + t) * C::NW);
      load_point<C>(p, points + (i * terms + t) * pw, pt_fmt);
      k256::lincomb_ref_term(p, k, tab + 16 * t, dig + 10 * t);
    }
    PtK256 r;
    k256::lincomb_ref_run(r, terms, tab, dig);
    if (out_fmt == FMT_PROJECTIVE) store_projective<C>(out + i * 3 * C::NW, r);
    else store_affine_from_projective<C>(out + i * 2 * C::NW, out_inf ? out_inf + i : nullptr, r);
  }
}

// ---------------------------------------------------------------------------------------------
// k256 variable-base scalar multiplication, throughput schedule (mulfast_k256.hpp).
// Each lane walks its elements with a grid stride, keeps up to BATCH Jacobian results in its
// private segment and converts them to affine with one shared inversion.
//
// WAVES is the occupancy target handed to the register allocator through __launch_bounds__: left
// alone it spends 256 VGPRs (+AGPRs) on instruction-level parallelism and ends at one wave per SIMD;
// WAVES = 4 gives 128 VGPRs (callable device functions cannot carry an occupancy target, so the phases
// are inlined).  The spill and scratch figures of the shipped build are in DESIGN.md section 4 (they
// move with BATCH: the BATCH results and prefix products live in the private segment by design).
// ---------------------------------------------------------------------------------------------
struct K256FastPrep {
  u32 w1[K256_DW], w2[K256_DW];   // recoded digits of the two GLV halves (k256::recode_half)
  u32 neg1, neg2;                 // signs of the halves
  u32 p_inf;
  FeK256 zfix;                    // common table denominator times the input's own Z
};

template <int WB>
__device__ __forceinline__ void k256_fast_prep(K256FastPrep* pp, TabSlotK256* tab, const u32* sc, const u32* src, int pt_fmt) {
  u32 k[8];
  words_load_be<8>(k, sc);
  k256::scalar_reduce_once(k);
  k256::GlvSplit s;
  k256::glv_split(s, k);
  k256::recode_half<WB>(pp->w1, s.k1);
  k256::recode_half<WB>(pp->w2, s.k2);
  pp->neg1 = s.neg1; pp->neg2 = s.neg2;

  FeK256 px, py, pz;
  bool p_inf;
  k256::from_be_words(px, src);
  k256::from_be_words(py, src + 8);
  if (pt_fmt == FMT_PROJECTIVE) {
    // homogeneous (X:Y:Z) = Jacobian (XZ, YZ^2, Z): run on the curve isomorphic by u = Z
    k256::from_be_words(pz, src + 16);
    p_inf = k256::is_zero(pz);
    FeK256 zz;
    k256::mul(px, px, pz);
    k256::sqr(zz, pz);
    k256::mul(py, py, zz);
  } else {
    u32 z = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) z |= src[w];
    p_inf = (z == 0);
    k256::set_one(pz);
  }
  if (p_inf) {   // keep the arithmetic on a valid point; the result is replaced by the identity
    PtK256 g; k256::generator(g);
    px = g.x; py = g.y; k256::set_one(pz);
  }
  pp->p_inf = p_inf ? 1u : 0u;
  FeK256 zg;
  k256::table_build_globalz<WB>(tab, zg, px, py);
  k256::mul(pp->zfix, zg, pz);
}

template <int WB>
__device__ __forceinline__ void k256_fast_loop(JacK256* out, const K256FastPrep* pp, const TabSlotK256* tab) {
  constexpr int NPOS = K256Win<WB>::NPOS;
  u32 w1[K256_DW], w2[K256_DW];
#pragma unroll
  for (int i = 0; i < K256_DW; i++) { w1[i] = pp->w1[i]; w2[i] = pp->w2[i]; }
  const bool n1 = pp->neg1 != 0, n2 = pp->neg2 != 0;
  JacK256 acc;
  k256::set_zero(acc.x); k256::set_zero(acc.y); k256::set_zero(acc.z);      // infinity
  // the top position (WB = 4: the carry digits) then the others; one shared body for all positions
#pragma unroll 1
  for (int i = NPOS - 1; i >= 0; i--) {
    if (i != NPOS - 1) {
#pragma unroll 1
      for (int j = 0; j < WB; j++) k256::jac_double(acc);
    }
    const int dg1 = k256::half_digit<WB>(w1, i), dg2 = k256::half_digit<WB>(w2, i);
#pragma unroll 1
    for (int h = 0; h < 2; h++) k256::add_digit(acc, tab, h ? dg2 : dg1, h != 0, h ? n2 : n1);
  }
  k256::mul(acc.z, acc.z, pp->zfix);     // back from the isomorphic curves
  if (pp->p_inf) k256::set_zero(acc.z);
  *out = acc;
}

// Montgomery's trick over the cnt results of this lane; element j of the batch is global index base + j*T.
__device__ __forceinline__ void k256_fast_finish(const JacK256* res, FeK256* pre, int cnt, size_t base, size_t T, u32* out, int out_fmt,
                                              uint8_t* out_inf) {
  jac::store_batch_affine<CurveK256>(res, pre, cnt, base, T, out, out_fmt, out_inf);
}

// Two-term linear combination k0*P0 + k1*P1 sharing the 128 doublings (LinearCombination::lincomb,
// k256 mul.rs:313-323 with N = 2: the ECDSA-verify shape u1*G + u2*Q).  Each term gets its own common-Z
// table; the two tables live on curves isomorphic by different factors, so each is rescaled by the other's
// factor (x u^2, y u^3) to put both on the curve isomorphic by zfix0 * zfix1.
template <int WB>
__device__ __forceinline__ void k256_fast_rescale(TabSlotK256* tab, const FeK256& s) {
  FeK256 s2, s3;
  k256::sqr(s2, s);
  k256::mul(s3, s2, s);
#pragma unroll 1
  for (int j = 0; j < K256Win<WB>::NE; j++) {
    FeK256 y;
    constexpr int SS = K256_SLOT_STRIDE;
    k256::mul(tab[SS * j].x, tab[SS * j].x, s2);
    k256::mul(y, tab[SS * j].y, s3);
    tab[SS * j].y = y;
    if constexpr (SS == 2) { k256::mul(tab[2 * j + 1].x, tab[2 * j + 1].x, s2); tab[2 * j + 1].y = y; }
  }
}

template <int BATCH, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k256_lincomb2_fast_kernel(const u32* scalars, const u32* points, int pt_fmt, u32* out,
                                                                 int out_fmt, uint8_t* out_inf, size_t n, TabSlotK256* table_ws, WaveSched sched) {
  constexpr int WB = 4, SLOTS = K256Win<WB>::SLOTS, NPOS = K256Win<WB>::NPOS;       // 4-bit windows: this kernel rescales both tables entry by entry (5 bits: -6 %)
  TabSlotK256* tab = table_ws + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (2 * SLOTS);
  K256FastPrep prep[2];
  JacK256 res[BATCH];
  FeK256 pre[BATCH];
  const int pw = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * 8;
  // dynamic work distribution (sched.hpp): small chunks drawn per wave, results buffered across chunks, one shared inversion per full buffer
  size_t idx[BATCH];
  int cnt = 0, slots = 0;
  for (;;) {
    size_t lo, hi;
    const bool more = wave_next_chunk(sched, lo, hi);
    const int units = more ? (int)((hi - lo + 63) / 64) : 0;
#pragma unroll 1
    for (int j = 0; j < units; j++) {
      const size_t i = lo + (threadIdx.x & 63u) + (size_t)j * 64;
      if (i >= hi) break;
#pragma unroll 1
      for (int t = 0; t < 2; t++) k256_fast_prep<WB>(&prep[t], tab + t * SLOTS, scalars + (2 * i + t) * 8, points + (2 * i + t) * pw, pt_fmt);
#pragma unroll 1
      for (int t = 0; t < 2; t++) k256_fast_rescale<WB>(tab + t * SLOTS, prep[1 - t].zfix);
      JacK256 acc;
      k256::set_zero(acc.x); k256::set_zero(acc.y); k256::set_zero(acc.z);
#pragma unroll 1
      for (int w = NPOS - 1; w >= 0; w--) {
        if (w != NPOS - 1) {
#pragma unroll 1
          for (int d = 0; d < WB; d++) k256::jac_double(acc);
        }
#pragma unroll 1
        for (int h = 0; h < 4; h++) {
          const K256FastPrep* pp = &prep[h >> 1];
          int dg = k256::half_digit<WB>((h & 1) ? pp->w2 : pp->w1, w);
          if (pp->p_inf) dg = 0;                       // an identity input contributes nothing
          k256::add_digit(acc, tab + (h >> 1) * SLOTS, dg, (h & 1) != 0, ((h & 1) ? pp->neg2 : pp->neg1) != 0);
        }
      }
      FeK256 zf;
      k256::mul(zf, prep[0].zfix, prep[1].zfix);
      k256::mul(acc.z, acc.z, zf);
      res[cnt] = acc;
      idx[cnt] = i;
      cnt++;
    }
    slots += units;
    if (!more || slots + (int)sched.chunk_units > BATCH) {
      if (cnt) jac::store_batch_affine<CurveK256>(res, pre, cnt, 0, 0, out, out_fmt, out_inf, idx);
      cnt = 0;
      slots = 0;
    }
    if (!more) break;
  }
}

template <int BATCH, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k256_mul_fast_kernel(const u32* scalars, const u32* points, int pt_fmt, u32* out,
                                                            int out_fmt, uint8_t* out_inf, size_t n, TabSlotK256* table_ws, WaveSched sched) {
  // this lane's table: 32 slots x 64 B, contiguous, in the launch's global workspace (gridDim * 256 lanes)
  constexpr int WB = K256_WB;
  TabSlotK256* tab = table_ws + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * K256Win<WB>::SLOTS;
#ifdef K256_BLOCK_TIMES              // DIAGNOSTIC build: every workgroup records when it started and ended and where it ran, behind the table workspace
  unsigned long long* blk_times = (unsigned long long*)(table_ws + (size_t)gridDim.x * blockDim.x * K256Win<WB>::SLOTS) + 4 * (size_t)blockIdx.x;
  if (threadIdx.x == 0) {
    blk_times[0] = wall_clock64();
    blk_times[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID: wave, simd, pipe, cu, sh, se
    blk_times[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // XCC_ID
  }
#endif
  K256FastPrep prep;
  JacK256 res[BATCH];
  FeK256 pre[BATCH];
  const int pw = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * 8;
#ifdef ECGPU_STATIC_GRID_STRIDE      // A/B switch: the static assignment of rounds 1-3 (every lane the same number of units, grid stride)
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t base = tid; base < n; base += T * BATCH) {
    int cnt = 0;
#pragma unroll 1
    for (int j = 0; j < BATCH; j++) {
      const size_t i = base + (size_t)j * T;
      if (i >= n) break;
      k256_fast_prep<WB>(&prep, tab, scalars + i * 8, points + i * pw, pt_fmt);
      k256_fast_loop<WB>(&res[j], &prep, tab);
      cnt = j + 1;
    }
    k256_fast_finish(res, pre, cnt, base, T, out, out_fmt, out_inf);
  }
#else
  // Every wave draws small chunks of 64 x u consecutive units (sched.hpp): lane l takes units lo + l, lo + l + 64, ..; the results stay in res[]
  // across chunks and are flushed - ONE shared inversion - when the buffer is full or the work has run out.
  size_t idx[BATCH];                 // global index of every buffered result
  int cnt = 0, slots = 0;            // results buffered by this lane; per-lane units drawn since the last flush (wave-uniform)
  for (;;) {
    size_t lo, hi;
    const bool more = wave_next_chunk(sched, lo, hi);
    const int units = more ? (int)((hi - lo + 63) / 64) : 0;
#pragma unroll 1
    for (int j = 0; j < units; j++) {
      const size_t i = lo + (threadIdx.x & 63u) + (size_t)j * 64;
      if (i >= hi) break;
      k256_fast_prep<WB>(&prep, tab, scalars + i * 8, points + i * pw, pt_fmt);
      k256_fast_loop<WB>(&res[cnt], &prep, tab);
      idx[cnt] = i;
      cnt++;
    }
    slots += units;
    if (!more || slots + (int)sched.chunk_units > BATCH) {
      if (cnt) jac::store_batch_affine<CurveK256>(res, pre, cnt, 0, 0, out, out_fmt, out_inf, idx);
      cnt = 0;
      slots = 0;
    }
    if (!more) break;
  }
#endif
#ifdef K256_BLOCK_TIMES
  __syncthreads();
  if (threadIdx.x == 0) blk_times[1] = wall_clock64();
#endif
}

template <class C>
__global__ void gen_table_kernel(typename C::Pt* tab) {
  if (blockIdx.x == 0 && threadIdx.x == 0) C::gen_table_build(tab);
}

// (occupancy target: the 12-limb curve needs the same two-waves budget as lincomb_ref_kernel - left alone the allocator takes
// 280 VGPRs, one wave per SIMD, and P-384's constant-time k G ran 29 % slower than its variable-base twin)
template <class C>
__global__ void __launch_bounds__(256, (C::NW > 8 ? ECGPU_REF_WAVES : 1)) mul_gen_ref_kernel(const u32* scalars, const typename C::Pt* gen_tab, u32* out,
                                                          int out_fmt, uint8_t* out_inf, size_t n) {
  typename C::Pt tab[C::ID == 0 ? 1 : C::REF_TABLE_PTS];   // scratch for curves whose mul_by_generator is G * k
  ECGPU_GRID_STRIDE(i, n) {
    u32 k[C::NW];
    C::scalar_load(k, scalars + i * C::NW);
    typename C::Pt r;
    C::mul_gen_ref(r, k, gen_tab, tab);
    if (out_fmt == FMT_PROJECTIVE) store_projective<C>(out + i * 3 * C::NW, r);
    else store_affine_from_projective<C>(out + i * 2 * C::NW, out_inf ? out_inf + i : nullptr, r);
  }
}

// ---------------------------------------------------------------------------------------------
// ok[i] = 1 iff every one of the `terms` scalars of element i is below the group order (Scalar::from_repr)
template <class C>
__global__ void __launch_bounds__(256) validate_scalars_kernel(const u32* scalars, uint8_t* ok, size_t n, size_t terms) {
  ECGPU_GRID_STRIDE(i, n) {
    u32 k[C::NW], o[C::NW];
    C::order(o);
    bool good = true;
#pragma unroll 1
    for (size_t t = 0; t < terms; t++) {
      C::scalar_load(k, scalars + (i * terms + t) * C::NW);
      good = good && !mp_geq<C::NW>(k, o);
    }
    ok[i] = good ? 1 : 0;
  }
}
template <class C>
__global__ void __launch_bounds__(256) validate_points_kernel(const u32* xy, uint8_t* ok, size_t n) {
  ECGPU_GRID_STRIDE(i, n) {
    const u32* src = xy + i * 2 * C::NW;
    u32 raw[2 * C::NW];
    u32 z = 0;
#pragma unroll
    for (int j = 0; j < 2 * C::NW; j++) { raw[j] = src[j]; z |= raw[j]; }
    typename C::Fe x, y, l, r;
    // canonical-range check must look at the raw integers, before any domain conversion
    bool canon = true;
    {
      u32 lx[C::NW], ly[C::NW], p[C::NW];
      words_load_be<C::NW>(lx, raw);
      words_load_be<C::NW>(ly, raw + C::NW);
      C::modulus(p);
      canon = !mp_geq<C::NW>(lx, p) && !mp_geq<C::NW>(ly, p);
    }
    C::fe_load(x, raw);
    C::fe_load(y, raw + C::NW);
    C::fe_sqr(l, y);
    C::curve_rhs(r, x);
    typename C::Fe d;
    C::fe_sub(d, l, r);
    ok[i] = (z == 0) || (canon && C::fe_is_zero(d)) ? 1 : 0;
  }
}
template <class C>
__global__ void __launch_bounds__(256) decompress_kernel(const u32* xs, const uint8_t* y_is_odd, u32* out_xy, uint8_t* ok, size_t n) {
  ECGPU_GRID_STRIDE(i, n) {
    u32 raw[C::NW], lx[C::NW], p[C::NW];
#pragma unroll
    for (int j = 0; j < C::NW; j++) raw[j] = xs[i * C::NW + j];
    words_load_be<C::NW>(lx, raw);
    C::modulus(p);
    const bool canon = !mp_geq<C::NW>(lx, p);
    typename C::Fe x, rhs, y, ny;
    C::fe_load(x, raw);
    C::curve_rhs(rhs, x);
    const bool has = C::fe_sqrt(y, rhs);
    C::fe_neg(ny, y);
    const bool odd = C::fe_is_odd(y);
    C::fe_select(y, odd == ((y_is_odd[i] & 1) != 0), y, ny);
    const bool good = canon && has;
    u32* o = out_xy + i * 2 * C::NW;
    typename C::Fe zero;
    C::fe_zero(zero);
    C::fe_select(x, good, x, zero);
    C::fe_select(y, good, y, zero);
    C::fe_store(o, x);
    C::fe_store(o + C::NW, y);
    ok[i] = good ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// synthetic inputs (oracle/synth.py is the specification)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 synth_word(u64 seed, u64 stream, u64 index, u32 j) {
  u64 z = (seed ^ (stream * 0xD1342543DE82EF95ull)) + (8 * index + j + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// value as little-endian 32-bit limbs (word 0 of the stream is the most significant)
template <int NW>
__device__ __forceinline__ void synth_value(u32* limbs, u64 seed, u64 stream, u64 index) {
#pragma unroll
  for (int j = 0; j < NW / 2; j++) {
    const u64 w = synth_word(seed, stream, index, j);
    limbs[NW - 1 - 2 * j] = (u32)(w >> 32);
    limbs[NW - 2 - 2 * j] = (u32)w;
  }
}
template <class C>
__global__ void __launch_bounds__(256) synth_scalars_kernel(u64 seed, u64 first, u32* out, size_t n) {
  ECGPU_GRID_STRIDE(i, n) {
    u32 v[C::NW], o[C::NW];
    synth_value<C::NW>(v, seed, 0, first + i);
    C::order(o);
    reduce_once<C::NW>(v, o);
    words_store_be<C::NW>(out + i * C::NW, v);
  }
}
template <class C>
__global__ void __launch_bounds__(256) synth_points_kernel(u64 seed, u64 first, u32* out_xy, size_t n) {
  ECGPU_GRID_STRIDE(i, n) {
    typename C::Fe x, y, ny, rhs;
    bool found = false;
    for (int t = 0; t < 64 && !found; t++) {
      u32 v[C::NW], p[C::NW], be[C::NW];
      synth_value<C::NW>(v, seed, 1 + t, first + i);
      C::modulus(p);
      reduce_once<C::NW>(v, p);
      const u32 want_odd = (u32)synth_word(seed, 1 + t, first + i, 7) & 1u;
      words_store_be<C::NW>(be, v);
      C::fe_load(x, be);
      C::curve_rhs(rhs, x);
      if (C::fe_sqrt(y, rhs)) {
        C::fe_neg(ny, y);
        const bool odd = C::fe_is_odd(y);
        C::fe_select(y, odd == (want_odd != 0), y, ny);
        found = true;
      }
    }
    u32* o = out_xy + i * 2 * C::NW;
    C::fe_store(o, x);
    C::fe_store(o + C::NW, y);
  }
}

}  // namespace ecgpu
