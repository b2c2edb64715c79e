// TEST-ONLY host build of the NIST (primeorder) templates through the curve traits.
#include <string.h>
#include <stdlib.h>
#include "hosttwin_trace.hpp"
#include "traits.hpp"
using namespace ecgpu;

template <class C> static void load(typename C::Fe& f, const uint8_t* b) { u32 w[C::NW]; memcpy(w, b, C::NB); C::fe_load(f, w); }
template <class C> static void store(uint8_t* b, const typename C::Fe& f) { u32 w[C::NW]; C::fe_store(w, f); memcpy(b, w, C::NB); }
template <class C> static void load_pt(typename C::Pt& p, const uint8_t* b) { load<C>(p.x, b); load<C>(p.y, b + C::NB); load<C>(p.z, b + 2 * C::NB); }
template <class C> static void store_pt(uint8_t* b, const typename C::Pt& p) { store<C>(b, p.x); store<C>(b + C::NB, p.y); store<C>(b + 2 * C::NB, p.z); }

template <class C>
static int fe_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, int n) {
  for (int i = 0; i < n; i++) {
    typename C::Fe x, y, r; load<C>(x, a + C::NB * i); if (b) load<C>(y, b + C::NB * i);
    int ok = 1;
    switch (op) {
      case 0: C::fe_mul(r, x, y); break;
      case 1: C::fe_sqr(r, x); break;
      case 2: C::fe_add(r, x, y); break;
      case 3: C::fe_sub(r, x, y); break;
      case 4: C::fe_neg(r, x); break;
      case 5: C::fe_inv(r, x); break;
      case 6: ok = C::fe_sqrt(r, x); break;
      default: return -1;
    }
    store<C>(out + C::NB * i, r);
    if (op == 6 && !ok) memset(out + C::NB * i, 0xFF, C::NB);
  }
  return 0;
}
template <class C>
static int pt_op(int op, const uint8_t* p, const uint8_t* q, uint8_t* out, int n) {
  for (int i = 0; i < n; i++) {
    typename C::Pt a, b, r; load_pt<C>(a, p + 3 * C::NB * i);
    switch (op) {
      case 0: load_pt<C>(b, q + 3 * C::NB * i); C::pt_add(r, a, b); break;
      case 1: {
        typename C::Af m; const uint8_t* s = q + (2 * C::NB + 1) * i;
        load<C>(m.x, s); load<C>(m.y, s + C::NB); m.inf = s[2 * C::NB];
        C::pt_add_mixed(r, a, m); break;
      }
      case 2: C::pt_double(r, a); break;
      default: return -1;
    }
    store_pt<C>(out + 3 * C::NB * i, r);
  }
  return 0;
}
template <class C>
static int mul_ref(const uint8_t* pts, const uint8_t* ks, uint8_t* out, int n, int gen) {
  typename C::Pt* tab = (typename C::Pt*)malloc(sizeof(typename C::Pt) * 16);
  typename C::Pt g; C::pt_generator(g);
  for (int i = 0; i < n; i++) {
    typename C::Pt p, r;
    if (gen) p = g; else load_pt<C>(p, pts + 3 * C::NB * i);
    u32 w[C::NW], k[C::NW]; memcpy(w, ks + C::NB * i, C::NB); C::scalar_load(k, w);
    C::mul_ref(r, p, k, tab);
    store_pt<C>(out + 3 * C::NB * i, r);
  }
  free(tab);
  return 0;
}

extern "C" {
int ht_nist_fe_op(int curve, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, int n) {
  return curve == 1 ? fe_op<CurveP256>(op, a, b, out, n) : fe_op<CurveP384>(op, a, b, out, n);
}
int ht_nist_pt_op(int curve, int op, const uint8_t* p, const uint8_t* q, uint8_t* out, int n) {
  return curve == 1 ? pt_op<CurveP256>(op, p, q, out, n) : pt_op<CurveP384>(op, p, q, out, n);
}
int ht_nist_mul_ref(int curve, const uint8_t* pts, const uint8_t* ks, uint8_t* out, int n, int gen) {
  return curve == 1 ? mul_ref<CurveP256>(pts, ks, out, n, gen) : mul_ref<CurveP384>(pts, ks, out, n, gen);
}
}

// ---- generic Jacobian group law (jacobian.hpp) over any curve --------------------------------------
#include "jacobian.hpp"
template <class C> static void load_jac(Jac<C>& p, const uint8_t* b) { load<C>(p.x, b); load<C>(p.y, b + C::NB); load<C>(p.z, b + 2 * C::NB); }
template <class C> static void store_jac(uint8_t* b, const Jac<C>& p) { store<C>(b, p.x); store<C>(b + C::NB, p.y); store<C>(b + 2 * C::NB, p.z); }
// op: 0 dbl, 1 add_mixed (q = x||y), 2 add (q Jacobian), 3 add_affine (p = (x, y, 1), q = x||y)
template <class C>
static int jac_op(int op, const uint8_t* p, const uint8_t* q, uint8_t* out, int n) {
  for (int i = 0; i < n; i++) {
    Jac<C> a, b, r; load_jac<C>(a, p + 3 * C::NB * i);
    if (op == 0) { r = a; jac::dbl<C>(r); }
    else if (op == 1) { typename C::Fe x, y; load<C>(x, q + 2 * C::NB * i); load<C>(y, q + 2 * C::NB * i + C::NB); r = a; jac::add_mixed<C>(r, x, y); }
    else if (op == 3) { typename C::Fe x, y; load<C>(x, q + 2 * C::NB * i); load<C>(y, q + 2 * C::NB * i + C::NB); r = a; jac::add_affine<C>(r, x, y); }
    else { load_jac<C>(b, q + 3 * C::NB * i); jac::add<C>(r, a, b); }
    store_jac<C>(out + 3 * C::NB * i, r);
  }
  return 0;
}
extern "C" int ht_jac_op(int curve, int op, const uint8_t* p, const uint8_t* q, uint8_t* out, int n) {
  if (curve == 0) return jac_op<CurveK256>(op, p, q, out, n);
  return curve == 1 ? jac_op<CurveP256>(op, p, q, out, n) : jac_op<CurveP384>(op, p, q, out, n);
}

// ---- per-lane body of the P-256 / P-384 variable-base kernel (varbase_lane.hpp), walked as `lanes` lanes -----------------
// The kernel gives lane tid the units tid, tid + T, ... in passes of BATCH; here T = lanes is small, so a few hundred
// units exercise every slot count 1..BATCH, several passes, the shared table inversion and the batched output.
#include "varbase_lane.hpp"
template <class C, int NT, int BATCH = 8>
static int vb_walk(const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n, size_t lanes) {
  vb::LaneWs<C, BATCH>* ws = (vb::LaneWs<C, BATCH>*)malloc(sizeof(vb::LaneWs<C, BATCH>));
  u32 digits[NT * C::NW];
  const DigitMem dm{digits, 1};
  for (size_t tid = 0; tid < lanes; tid++)
    for (size_t base = tid; base < n; base += lanes * (BATCH / NT))
      vb::lane_pass<C, BATCH, NT>((const u32*)scalars, (const u32*)points, pt_fmt, (u32*)out, out_fmt, out_inf, n, base, lanes, *ws, dm);
  free(ws);
  return 0;
}
// the 5-bit-window form (16 table entries, ceil((32 NW + 1) / 5) positions), 16 slots per pass
template <class C>
static int vb_walk5(const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n, size_t lanes) {
  using Ws = vb::LaneWs<C, 16, 5>;
  Ws* ws = (Ws*)malloc(sizeof(Ws));
  u32 digits[vb::digit_words<C, 5>()];
  const DigitMem dm{digits, 1};
  for (size_t tid = 0; tid < lanes; tid++)
    for (size_t base = tid; base < n; base += lanes * 16)
      vb::lane_pass<C, 16, 1, 5>((const u32*)scalars, (const u32*)points, pt_fmt, (u32*)out, out_fmt, out_inf, n, base, lanes, *ws, dm);
  free(ws);
  return 0;
}
extern "C" int ht_vb_mul_w5(int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n,
                            size_t lanes) {
  return curve == 1 ? vb_walk5<CurveP256>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes)
                    : vb_walk5<CurveP384>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes);
}
// n units of `terms` (1 or 2) terms each
extern "C" int ht_vb_lincomb(int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, int terms, uint8_t* out, int out_fmt, uint8_t* out_inf,
                             size_t n, size_t lanes) {
  if (terms == 2)
    return curve == 1 ? vb_walk<CurveP256, 2>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes)
                      : vb_walk<CurveP384, 2>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes);
  return curve == 1 ? vb_walk<CurveP256, 1>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes)
                    : vb_walk<CurveP384, 1>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes);
}
// the product's own pass size (VBB = 16 table slots per lane and pass, ops_nist.inc)
extern "C" int ht_vb_mul16(int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n,
                           size_t lanes) {
  return curve == 1 ? vb_walk<CurveP256, 1, 16>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes)
                    : vb_walk<CurveP384, 1, 16>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes);
}
extern "C" int ht_vb_mul(int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n,
                         size_t lanes) {
  return ht_vb_lincomb(curve, scalars, points, pt_fmt, 1, out, out_fmt, out_inf, n, lanes);
}

// ---- constant-time variable base (varbase_ct.hpp): the per-lane body walked as `lanes` lanes over a lane-interleaved
//      workspace of stride `lanes`, exactly as the kernel lays it out with stride 256.  The trace hook records the table
//      entries the window loop reads.
#include "varbase_ct.hpp"
template <class C, int BATCH = 8>
static int vbct_walk(const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n, size_t lanes) {
  vbct::Chunk* mem = (vbct::Chunk*)aligned_alloc(16, sizeof(vbct::Chunk) * vbct::lane_chunks<C, BATCH>() * lanes);
  memset(mem, 0xA5, sizeof(vbct::Chunk) * vbct::lane_chunks<C, BATCH>() * lanes);
  u32 digits[C::NW];
  const DigitMem dm{digits, 1};
  for (size_t tid = 0; tid < lanes; tid++) {
    const vbct::LaneMem ws{mem + tid, lanes};
    for (size_t base = tid; base < n; base += lanes * BATCH)
      vbct::lane_pass<C, BATCH>((const u32*)scalars, (const u32*)points, pt_fmt, (u32*)out, out_fmt, out_inf, n, base, lanes, ws, dm);
  }
  free(mem);
  return 0;
}
#include "varbase_ct_k256.hpp"
// secp256k1: the constant-time body (GLV, Jacobian formulas over a common-Z table: varbase_ct_k256.hpp), same walk
template <int BATCH>
static int vbct_walk_k256(const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n, size_t lanes) {
  vbct::Chunk* mem = (vbct::Chunk*)aligned_alloc(16, sizeof(vbct::Chunk) * vbct::k256_lane_chunks<BATCH>() * lanes);
  memset(mem, 0xA5, sizeof(vbct::Chunk) * vbct::k256_lane_chunks<BATCH>() * lanes);
  u32 digits[8];
  const DigitMem dm{digits, 1};
  for (size_t tid = 0; tid < lanes; tid++) {
    const vbct::LaneMem ws{mem + tid, lanes};
    for (size_t base = tid; base < n; base += lanes * BATCH)
      vbct::lane_pass_k256<BATCH>((const u32*)scalars, (const u32*)points, pt_fmt, (u32*)out, out_fmt, out_inf, n, base, lanes, ws, dm);
  }
  free(mem);
  return 0;
}
extern "C" int ht_vbct_mul16(int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n,
                             size_t lanes) {
  if (curve == 0) return vbct_walk_k256<16>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes);
  return curve == 1 ? vbct_walk<CurveP256, 16>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes)
                    : vbct_walk<CurveP384, 16>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes);
}
extern "C" int ht_vbct_mul(int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n,
                           size_t lanes) {
  if (curve == 0) return vbct_walk_k256<8>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes);
  return curve == 1 ? vbct_walk<CurveP256>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes)
                    : vbct_walk<CurveP384>(scalars, points, pt_fmt, out, out_fmt, out_inf, n, lanes);
}

// ---- XYZZ bucket accumulator of the MSM (msm.hpp) over any curve: p (X||Y||ZZ||ZZZ) += q (affine x||y), converted to the
//      Jacobian triple the reduction tree works on (odd entries: and back to XYZZ and forth once more); out = X||Y||Z
#include "msm.hpp"
template <class C>
static int xyzz_op(const uint8_t* p, const uint8_t* q, uint8_t* out, int n) {
  for (int i = 0; i < n; i++) {
    msm::Xyzz<C> a;
    load<C>(a.x, p + 4 * C::NB * i); load<C>(a.y, p + 4 * C::NB * i + C::NB); load<C>(a.zz, p + 4 * C::NB * i + 2 * C::NB); load<C>(a.zzz, p + 4 * C::NB * i + 3 * C::NB);
    typename C::Fe x, y; load<C>(x, q + 2 * C::NB * i); load<C>(y, q + 2 * C::NB * i + C::NB);
    msm::xyzz_add_mixed<C>(a, x, y);
    Jac<C> r;
    msm::xyzz_to_jacobian<C>(r, a);
    if ((i & 3) == 2) {                           // every fourth entry: the same sum by the general XYZZ addition instead
      load<C>(a.x, p + 4 * C::NB * i); load<C>(a.y, p + 4 * C::NB * i + C::NB); load<C>(a.zz, p + 4 * C::NB * i + 2 * C::NB); load<C>(a.zzz, p + 4 * C::NB * i + 3 * C::NB);
      // the addend in XYZZ form with a denominator that is not one: (x u^2, y u^3, u^2, u^3) for u = x + 1
      typename C::Fe one, u, u2, u3; C::fe_one(one); C::fe_add(u, x, one); C::fe_sqr(u2, u); C::fe_mul(u3, u2, u);
      msm::Xyzz<C> qq; C::fe_mul(qq.x, x, u2); C::fe_mul(qq.y, y, u3); qq.zz = u2; qq.zzz = u3;
      msm::xyzz_add<C>(a, qq);
      msm::xyzz_to_jacobian<C>(r, a);
    }
    if (i & 1) {                                  // odd entries also go through the way back, as the folded bucket pieces do
      msm::Xyzz<C> b;
      msm::jacobian_to_xyzz<C>(b, r);
      msm::xyzz_to_jacobian<C>(r, b);
    }
    store_jac<C>(out + 3 * C::NB * i, r);
  }
  return 0;
}
extern "C" int ht_xyzz_add_mixed(int curve, const uint8_t* p, const uint8_t* q, uint8_t* out, int n) {
  if (curve == 0) return xyzz_op<CurveK256>(p, q, out, n);
  return curve == 1 ? xyzz_op<CurveP256>(p, q, out, n) : xyzz_op<CurveP384>(p, q, out, n);
}

// ---- hash to curve: map_to_curve (count = 1) or the sum of two maps (count = 2), as the kernel does it (h2c_map.hpp) ----
#include "h2c_map.hpp"
template <class C>
static int h2c_map_host(const uint8_t* u, int count, uint8_t* out_xy, uint8_t* out_inf, int n) {
  for (int i = 0; i < n; i++) {
    typename C::Fe uu;
    typename C::Pt p;
    load<C>(uu, u + (size_t)C::NB * count * i);
    h2c::map_to_curve<C>(p, uu);
    if (count == 2) {
      typename C::Pt q, r;
      load<C>(uu, u + (size_t)C::NB * (count * i + 1));
      h2c::map_to_curve<C>(q, uu);
      C::pt_add(r, p, q);
      p = r;
    }
    typename C::Fe zi, x, y;
    const bool inf = C::fe_is_zero(p.z);
    C::fe_inv(zi, p.z);
    C::fe_mul(x, p.x, zi); C::fe_mul(y, p.y, zi);
    if (inf) { C::fe_zero(x); C::fe_zero(y); }
    store<C>(out_xy + 2 * C::NB * i, x); store<C>(out_xy + 2 * C::NB * i + C::NB, y);
    out_inf[i] = inf ? 1 : 0;
  }
  return 0;
}
extern "C" int ht_h2c_map(int curve, const uint8_t* u, int count, uint8_t* out_xy, uint8_t* out_inf, int n) {
  if (curve == 0) return h2c_map_host<CurveK256>(u, count, out_xy, out_inf, n);
  return curve == 1 ? h2c_map_host<CurveP256>(u, count, out_xy, out_inf, n) : h2c_map_host<CurveP384>(u, count, out_xy, out_inf, n);
}

// ---- constant-time fixed base (fixedbase_ct.hpp): k G from a table T[j][d-1] = d 2^(5j) G given as canonical x||y rows;
//      out = X||Y||Z (the reference's homogeneous projective coordinates).  The trace hook records every table entry read.
#include "fixedbase_ct.hpp"
template <class C>
static int mul_ct(const uint8_t* table_xy, const uint8_t* ks, uint8_t* out, int n) {
  const int total = fb::ct_nwin<C>() * fb::CT_ENTRIES;
  AffEntry<C>* tab = (AffEntry<C>*)malloc(sizeof(AffEntry<C>) * total);
  for (int e = 0; e < total; e++) { load<C>(tab[e].x, table_xy + 2 * C::NB * e); load<C>(tab[e].y, table_xy + 2 * C::NB * e + C::NB); }
  for (int i = 0; i < n; i++) {
    u32 w[C::NW], k[C::NW], ord[C::NW];
    memcpy(w, ks + C::NB * i, C::NB);
    C::scalar_load(k, w);
    C::order(ord);
    reduce_once<C::NW>(k, ord);
    typename C::Pt r;
    fb::mul_ct_one<C>(r, k, tab);
    store_pt<C>(out + 3 * C::NB * i, r);
  }
  free(tab);
  return 0;
}
extern "C" int ht_mul_ct(int curve, const uint8_t* table_xy, const uint8_t* ks, uint8_t* out, int n) {
  if (curve == 0) return mul_ct<CurveK256>(table_xy, ks, out, n);
  return curve == 1 ? mul_ct<CurveP256>(table_xy, ks, out, n) : mul_ct<CurveP384>(table_xy, ks, out, n);
}

// ---- linear combinations of 3 .. 1024 terms (straus.hpp): both stages walked as `lanes` lanes with the product's own plan for
//      `plan_lanes` resident lanes (g_force > 0 pins the group size instead), and the exact-(X, Y, Z) run-time-N form for secp256k1
#include "straus.hpp"
template <class C>
static int straus_walk(const uint8_t* scalars, const uint8_t* points, int pt_fmt, int terms, uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n, size_t lanes,
                       size_t plan_lanes, int g_force, int* g_used) {
  int g, gpc;
  straus::plan(n, (size_t)terms, plan_lanes, &g, &gpc);
  if (g_force > 0) { g = g_force; gpc = (terms + g - 1) / g; }
  if (g_used) { g_used[0] = g; g_used[1] = gpc; }
  const size_t items = n * (size_t)gpc, upp = (size_t)(straus::SLOTS / g);
  straus::LaneWs<C>* ws = (straus::LaneWs<C>*)malloc(sizeof(straus::LaneWs<C>));
  u32* partial = (u32*)malloc(items * 3 * C::NW * sizeof(u32));
  memset(ws, 0xA5, sizeof(*ws));
  for (size_t tid = 0; tid < lanes; tid++)
    for (size_t base = tid; base < items; base += lanes * upp)
      straus::lane_pass<C>((const u32*)scalars, (const u32*)points, pt_fmt, terms, g, gpc, items, base, lanes, *ws, partial);
  for (size_t tid = 0; tid < lanes; tid++)
    for (size_t base = tid; base < n; base += lanes * 16)
      straus::fold_pass<C, 16>(partial, gpc, (u32*)out, out_fmt, out_inf, n, base, lanes);
  free(partial);
  free(ws);
  return 0;
}
extern "C" int ht_straus(int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, int terms, uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n,
                         size_t lanes, size_t plan_lanes, int g_force, int* g_used) {
  if (curve == 0) return straus_walk<CurveK256>(scalars, points, pt_fmt, terms, out, out_fmt, out_inf, n, lanes, plan_lanes, g_force, g_used);
  return curve == 1 ? straus_walk<CurveP256>(scalars, points, pt_fmt, terms, out, out_fmt, out_inf, n, lanes, plan_lanes, g_force, g_used)
                    : straus_walk<CurveP384>(scalars, points, pt_fmt, terms, out, out_fmt, out_inf, n, lanes, plan_lanes, g_force, g_used);
}
// secp256k1 lincomb_ext over a slice, exact (X, Y, Z): points X || Y || Z in, X || Y || Z out
extern "C" int ht_k256_lincomb_ref_n(const uint8_t* scalars, const uint8_t* points_xyz, int terms, uint8_t* out_xyz, size_t n) {
  using C = CurveK256;
  PtK256* tab = (PtK256*)malloc(sizeof(PtK256) * 16 * (size_t)terms);
  u32* dig = (u32*)malloc(sizeof(u32) * 10 * (size_t)terms);
  for (size_t i = 0; i < n; i++) {
    for (int t = 0; t < terms; t++) {
      PtK256 p;
      load_pt<C>(p, points_xyz + 96 * (i * terms + t));
      u32 w[8], k[8];
      memcpy(w, scalars + 32 * (i * terms + t), 32);
      C::scalar_load(k, w);
      k256::lincomb_ref_term(p, k, tab + 16 * t, dig + 10 * t);
    }
    PtK256 r;
    k256::lincomb_ref_run(r, terms, tab, dig);
    store_pt<C>(out_xyz + 96 * i, r);
  }
  free(tab);
  free(dig);
  return 0;
}
