// TEST-ONLY host build of the k256 point / scalar-multiplication templates.
#include <string.h>
#include "hosttwin_trace.hpp"
#include <stdlib.h>
#include "mul_k256.hpp"
using namespace ecgpu;

static void load(FeK256& f, const uint8_t* b) { u32 w[8]; memcpy(w, b, 32); k256::from_be_words(f, w); }
static void store(uint8_t* b, const FeK256& f0) { FeK256 f; k256::normalize(f, f0); u32 w[8]; k256::to_be_words(w, f); memcpy(b, w, 32); }
static void load_pt(PtK256& p, const uint8_t* b) { load(p.x, b); load(p.y, b + 32); load(p.z, b + 64); }
static void store_pt(uint8_t* b, const PtK256& p) { store(b, p.x); store(b + 32, p.y); store(b + 64, p.z); }
static void load_scalar(u32* k, const uint8_t* b) { u32 w[8]; memcpy(w, b, 32); for (int i = 0; i < 8; i++) k[i] = bswap32(w[7 - i]); }

extern "C" {
// op: 0 add, 1 add_mixed (q = x||y||inf, 65 B), 2 double, 3 neg, 4 endomorphism
int ht_k256_pt_op(int op, const uint8_t* p, const uint8_t* q, uint8_t* out, int n) {
  for (int i = 0; i < n; i++) {
    PtK256 a, b, r; load_pt(a, p + 96 * i);
    switch (op) {
      case 0: load_pt(b, q + 96 * i); k256::pt_add(r, a, b); break;
      case 1: { AfK256 m; load(m.x, q + 65 * i); load(m.y, q + 65 * i + 32); m.inf = q[65 * i + 64]; k256::pt_add_mixed(r, a, m); break; }
      case 2: k256::pt_double(r, a); break;
      case 3: k256::pt_neg(r, a); break;
      case 4: k256::pt_endomorphism(r, a); break;
      default: return -1;
    }
    store_pt(out + 96 * i, r);
  }
  return 0;
}
// out per element: k1 (16 B BE) || k2 (16 B BE) || neg1 || neg2
int ht_k256_glv(const uint8_t* ks, uint8_t* out, int n) {
  for (int i = 0; i < n; i++) {
    u32 k[8]; load_scalar(k, ks + 32 * i);
    k256::GlvSplit s; k256::glv_split(s, k);
    uint8_t* o = out + 34 * i;
    for (int j = 0; j < 4; j++) { u32 w = bswap32(s.k1[3 - j]); memcpy(o + 4 * j, &w, 4); w = bswap32(s.k2[3 - j]); memcpy(o + 16 + 4 * j, &w, 4); }
    o[32] = s.neg1; o[33] = s.neg2;
  }
  return 0;
}
int ht_k256_mul_ref(const uint8_t* pts, const uint8_t* ks, uint8_t* out, int n) {
  PtK256* tab = (PtK256*)malloc(sizeof(PtK256) * 16);
  for (int i = 0; i < n; i++) {
    PtK256 p, r; load_pt(p, pts + 96 * i);
    u32 k[8]; load_scalar(k, ks + 32 * i);
    k256::mul_ref(r, p, k, tab);
    store_pt(out + 96 * i, r);
  }
  free(tab);
  return 0;
}
// out[i] = k[2i]*P[2i] + k[2i+1]*P[2i+1]   (lincomb, mul.rs:313-323 with N = 2)
int ht_k256_lincomb2_ref(const uint8_t* pts, const uint8_t* ks, uint8_t* out, int n) {
  PtK256* tab = (PtK256*)malloc(sizeof(PtK256) * 32);
  for (int i = 0; i < n; i++) {
    PtK256 p[2], r; load_pt(p[0], pts + 192 * i); load_pt(p[1], pts + 192 * i + 96);
    u32 k[2][8]; load_scalar(k[0], ks + 64 * i); load_scalar(k[1], ks + 64 * i + 32);
    k256::lincomb_ref<2>(r, p, k, tab);
    store_pt(out + 96 * i, r);
  }
  free(tab);
  return 0;
}
int ht_k256_mul_gen_ref(const uint8_t* ks, uint8_t* out, int n) {
  static PtK256* tab = nullptr;
  if (!tab) { tab = (PtK256*)malloc(sizeof(PtK256) * 33 * 8); PtK256 g; k256::generator(g); k256::gen_table_build(tab, g); }
  for (int i = 0; i < n; i++) {
    u32 k[8]; load_scalar(k, ks + 32 * i);
    PtK256 r; k256::mul_gen_ref(r, k, tab);
    store_pt(out + 96 * i, r);
  }
  return 0;
}
int ht_k256_to_affine(const uint8_t* pts, uint8_t* out, int n) {
  for (int i = 0; i < n; i++) {
    PtK256 p; load_pt(p, pts + 96 * i);
    AfK256 a; k256::pt_to_affine(a, p);
    store(out + 65 * i, a.x); store(out + 65 * i + 32, a.y); out[65 * i + 64] = (uint8_t)a.inf;
  }
  return 0;
}
}

// ---- throughput schedule (mulfast_k256.hpp) -----------------------------------------------------
#include "mulfast_k256.hpp"
extern "C" {
}  // extern "C"
// points: affine x||y (zeros = identity) or, with proj != 0, homogeneous X||Y||Z; out: x||y||inf (65 B); WB = window width (4 | 5)
template <int WB>
static int mul_fast_walk(const uint8_t* pts, int proj, const uint8_t* ks, uint8_t* out, int n, int batch) {
  TabSlotK256 tab[K256Win<WB>::SLOTS];
  JacK256* res = (JacK256*)malloc(sizeof(JacK256) * batch);
  FeK256* pre = (FeK256*)malloc(sizeof(FeK256) * batch * 3);
  u32* inf = (u32*)malloc(sizeof(u32) * batch);
  const int pw = proj ? 96 : 64;
  for (int base = 0; base < n; base += batch) {
    int cnt = (n - base < batch) ? n - base : batch;
    for (int j = 0; j < cnt; j++) {
      const uint8_t* src = pts + (size_t)pw * (base + j);
      FeK256 px, py, pz; load(px, src); load(py, src + 32);
      bool p_inf;
      if (proj) {
        load(pz, src + 64); p_inf = k256::is_zero(pz);
        FeK256 zz; k256::mul(px, px, pz); k256::sqr(zz, pz); k256::mul(py, py, zz);
      } else {
        int z = 1; for (int b = 0; b < 64; b++) z &= (src[b] == 0);
        p_inf = z; k256::set_one(pz);
      }
      if (p_inf) { PtK256 g; k256::generator(g); px = g.x; py = g.y; k256::set_one(pz); }
      u32 k[8]; load_scalar(k, ks + 32 * (base + j));
      k256::scalar_reduce_once(k);
      k256::mul_fast_jac<WB>(res[j], px, py, k, tab);
      k256::mul(res[j].z, res[j].z, pz);
      FeK256 zero; k256::set_zero(zero);
      k256::select(res[j].z, p_inf, zero, res[j].z);
    }
    k256::jac_batch_to_affine<0>(pre + batch, pre + 2 * batch, inf, res, cnt, pre);
    for (int j = 0; j < cnt; j++) {
      uint8_t* o = out + 65 * (size_t)(base + j);
      store(o, pre[batch + j]); store(o + 32, pre[2 * batch + j]); o[64] = (uint8_t)inf[j];
    }
  }
  free(res); free(pre); free(inf);
  return 0;
}
extern "C" {
// the product's window width (K256_WB) and, explicitly, both widths (the two-term kernel stays on 4 bits)
int ht_k256_mul_fast(const uint8_t* pts, int proj, const uint8_t* ks, uint8_t* out, int n, int batch) { return mul_fast_walk<K256_WB>(pts, proj, ks, out, n, batch); }
int ht_k256_mul_fast_w(int wb, const uint8_t* pts, int proj, const uint8_t* ks, uint8_t* out, int n, int batch) {
  return wb == 4 ? mul_fast_walk<4>(pts, proj, ks, out, n, batch) : mul_fast_walk<5>(pts, proj, ks, out, n, batch);
}
// r = P + Q with P Jacobian (X||Y||Z, x = X/Z^2) and Q affine: out Jacobian X||Y||Z (canonical bytes)
int ht_k256_jac_add_mixed(const uint8_t* p, const uint8_t* q, uint8_t* out, int n) {
  for (int i = 0; i < n; i++) {
    JacK256 a, r; load(a.x, p + 96 * i); load(a.y, p + 96 * i + 32); load(a.z, p + 96 * i + 64);
    FeK256 x, y; load(x, q + 64 * i); load(y, q + 64 * i + 32);
    k256::jac_add_mixed(r, a, x, y, nullptr);
    store(out + 96 * i, r.x); store(out + 96 * i + 32, r.y); store(out + 96 * i + 64, r.z);
  }
  return 0;
}
int ht_k256_jac_double(const uint8_t* p, uint8_t* out, int n) {
  for (int i = 0; i < n; i++) {
    JacK256 a, r; load(a.x, p + 96 * i); load(a.y, p + 96 * i + 32); load(a.z, p + 96 * i + 64);
    k256::jac_double(r, a);
    store(out + 96 * i, r.x); store(out + 96 * i + 32, r.y); store(out + 96 * i + 64, r.z);
  }
  return 0;
}
}

// ---- MSM primitive: general Jacobian addition with exceptional cases ------------------------------
#include "msm.hpp"
extern "C" int ht_k256_jac_add(const uint8_t* p, const uint8_t* q, uint8_t* out, int n) {
  for (int i = 0; i < n; i++) {
    Jac<CurveK256> a, b, r;
    load(a.x, p + 96 * i); load(a.y, p + 96 * i + 32); load(a.z, p + 96 * i + 64);
    load(b.x, q + 96 * i); load(b.y, q + 96 * i + 32); load(b.z, q + 96 * i + 64);
    msm::pt_add<CurveK256>(r, a, b);
    store(out + 96 * i, r.x); store(out + 96 * i + 32, r.y); store(out + 96 * i + 64, r.z);
  }
  return 0;
}
