/* ecoracle - CPU restatement in C of the reference's algorithms for the hot path.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ (large-batch parity on the GPU box), by
 * __graft_entry__.smoke() and by the cpu_baseline leg of bench.py.  Never linked into or called
 * by the product (rustcrypto-elliptic-curves_amd/).  The reference is Rust and cannot be built in
 * this image (no rustc, un-vendored crates), so this is a port ("kind": "port"), pinned by
 * tests/test_oracle_c.py against oracle/ecmodel.py, which is itself pinned against the
 * reference's own known-answer vectors (tests/golden/).
 *
 * What follows the reference limb for limb (so that timing it is a fair CPU baseline):
 *   k256 field      5x52 lazily reduced limbs, magnitude discipline    k256/src/arithmetic/field/field_5x52.rs
 *   k256 group law  RCB complete formulas incl. weak normalisations     k256/src/arithmetic/projective.rs:96-274
 *   k256 mul        GLV + signed radix-16 + constant-time table scan    k256/src/arithmetic/mul.rs:59-445
 *   k256 scalars    4x64 words, mul_wide + reduce                       k256/src/arithmetic/scalar/wide64.rs
 *   p256 field      4x64 Montgomery, HAC 14.32 word-by-word reduction   p256/src/arithmetic/field.rs:118-319
 *   p384 field      6x64 word-by-word Montgomery (what fiat-crypto's    p384/src/arithmetic/field/p384_64.rs:146
 *                   generated code computes; not a transliteration)
 *   primeorder      RCB a=-3 formulas, 4-bit window mul                 primeorder/src/point_arithmetic.rs:199-317,
 *                                                                       primeorder/src/projective.rs:106-150
 * Build: make -C oracle   (gcc -O3 -march=native, unsigned __int128)
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;
typedef uint8_t u8;

/* ======================================================================================== */
/* k256 base field, 5x52 (field_5x52.rs)                                                      */
/* ======================================================================================== */
typedef struct { u64 n[5]; } fe5;
#define M52 0xFFFFFFFFFFFFFULL

static void fe5_from_bytes(fe5* r, const u8* b) { /* field_5x52.rs:27-68 */
  u64 w[4];
  for (int i = 0; i < 4; i++) {
    u64 v = 0;
    for (int j = 0; j < 8; j++) v = (v << 8) | b[8 * (3 - i) + j];
    w[i] = v;
  }
  r->n[0] = w[0] & M52;
  r->n[1] = ((w[0] >> 52) | (w[1] << 12)) & M52;
  r->n[2] = ((w[1] >> 40) | (w[2] << 24)) & M52;
  r->n[3] = ((w[2] >> 28) | (w[3] << 36)) & M52;
  r->n[4] = w[3] >> 16;
}
static void fe5_to_bytes(u8* b, const fe5* a) { /* field_5x52.rs:96-131 (input normalised) */
  u64 w[4];
  w[0] = a->n[0] | (a->n[1] << 52);
  w[1] = (a->n[1] >> 12) | (a->n[2] << 40);
  w[2] = (a->n[2] >> 24) | (a->n[3] << 28);
  w[3] = (a->n[3] >> 36) | (a->n[4] << 16);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 8; j++) b[8 * (3 - i) + j] = (u8)(w[i] >> (56 - 8 * j));
}
static fe5 fe5_add_modulus_correction(const fe5* a, u64 x) { /* :134-152 */
  fe5 r;
  u64 t0 = a->n[0] + x * 0x1000003D1ULL;
  u64 t1 = a->n[1] + (t0 >> 52); t0 &= M52;
  u64 t2 = a->n[2] + (t1 >> 52); t1 &= M52;
  u64 t3 = a->n[3] + (t2 >> 52); t2 &= M52;
  u64 t4 = a->n[4] + (t3 >> 52); t3 &= M52;
  r.n[0] = t0; r.n[1] = t1; r.n[2] = t2; r.n[3] = t3; r.n[4] = t4;
  return r;
}
static int fe5_get_overflow(const fe5* a) { /* :163-170 */
  u64 m = a->n[1] & a->n[2] & a->n[3];
  return ((a->n[4] >> 48) != 0) | ((a->n[4] == 0x0FFFFFFFFFFFFULL) & (m == M52) & (a->n[0] >= 0xFFFFEFFFFFC2FULL));
}
static fe5 fe5_normalize_weak(const fe5* a) { /* :173-184 */
  fe5 t = *a;
  u64 x = t.n[4] >> 48;
  t.n[4] &= 0x0FFFFFFFFFFFFULL;
  return fe5_add_modulus_correction(&t, x);
}
static fe5 fe5_normalize(const fe5* a) { /* :189-206 */
  fe5 res = fe5_normalize_weak(a);
  int overflow = fe5_get_overflow(&res);
  fe5 c = fe5_add_modulus_correction(&res, 1);
  c.n[4] &= 0x0FFFFFFFFFFFFULL;
  return overflow ? c : res;
}
static int fe5_normalizes_to_zero(const fe5* a) { /* :209-223 */
  fe5 r = fe5_normalize_weak(a);
  u64 z0 = r.n[0] | r.n[1] | r.n[2] | r.n[3] | r.n[4];
  u64 z1 = (r.n[0] ^ 0x1000003D0ULL) & r.n[1] & r.n[2] & r.n[3] & (r.n[4] ^ 0xF000000000000ULL);
  return (z0 == 0) | (z1 == M52);
}
static fe5 fe5_negate(const fe5* a, unsigned magnitude) { /* :252-260 */
  u64 m = magnitude + 1;
  fe5 r;
  r.n[0] = 0xFFFFEFFFFFC2FULL * 2 * m - a->n[0];
  r.n[1] = M52 * 2 * m - a->n[1];
  r.n[2] = M52 * 2 * m - a->n[2];
  r.n[3] = M52 * 2 * m - a->n[3];
  r.n[4] = 0x0FFFFFFFFFFFFULL * 2 * m - a->n[4];
  return r;
}
static fe5 fe5_add(const fe5* a, const fe5* b) { /* :264-272 */
  fe5 r;
  for (int i = 0; i < 5; i++) r.n[i] = a->n[i] + b->n[i];
  return r;
}
static fe5 fe5_mul_single(const fe5* a, u64 k) { /* :276-285 */
  fe5 r;
  for (int i = 0; i < 5; i++) r.n[i] = a->n[i] * k;
  return r;
}
static fe5 fe5_double(const fe5* a) { return fe5_add(a, a); }
/* mul_inner, field_5x52.rs:288-449: p0..p8 column sums, p5..p8 folded with R = 0x1000003D10 */
static fe5 fe5_mul(const fe5* x, const fe5* y) {
  const u64 a0 = x->n[0], a1 = x->n[1], a2 = x->n[2], a3 = x->n[3], a4 = x->n[4];
  const u64 b0 = y->n[0], b1 = y->n[1], b2 = y->n[2], b3 = y->n[3], b4 = y->n[4];
  const u64 R = 0x1000003D10ULL;
  u128 c, d;
  u64 t3, t4, tx, u0, r0, r1, r2, r3, r4;
  d = (u128)a0 * b3 + (u128)a1 * b2 + (u128)a2 * b1 + (u128)a3 * b0;
  c = (u128)a4 * b4;
  d += (u128)((u64)c & M52) * R; c >>= 52;
  t3 = (u64)d & M52; d >>= 52;
  d += (u128)a0 * b4 + (u128)a1 * b3 + (u128)a2 * b2 + (u128)a3 * b1 + (u128)a4 * b0;
  d += (u128)(u64)c * R;
  t4 = (u64)d & M52; d >>= 52;
  tx = t4 >> 48; t4 &= (M52 >> 4);
  c = (u128)a0 * b0;
  d += (u128)a1 * b4 + (u128)a2 * b3 + (u128)a3 * b2 + (u128)a4 * b1;
  u0 = (u64)d & M52; d >>= 52;
  u0 = (u0 << 4) | tx;
  c += (u128)u0 * (R >> 4);
  r0 = (u64)c & M52; c >>= 52;
  c += (u128)a0 * b1 + (u128)a1 * b0;
  d += (u128)a2 * b4 + (u128)a3 * b3 + (u128)a4 * b2;
  c += (u128)((u64)d & M52) * R; d >>= 52;
  r1 = (u64)c & M52; c >>= 52;
  c += (u128)a0 * b2 + (u128)a1 * b1 + (u128)a2 * b0;
  d += (u128)a3 * b4 + (u128)a4 * b3;
  c += (u128)((u64)d & M52) * R; d >>= 52;
  r2 = (u64)c & M52; c >>= 52;
  c += (u128)(u64)d * R + t3;
  r3 = (u64)c & M52; c >>= 52;
  r4 = (u64)c + t4;
  fe5 r = {{r0, r1, r2, r3, r4}};
  return r;
}
static fe5 fe5_sqr(const fe5* a) { return fe5_mul(a, a); } /* :462-464 */
static fe5 fe5_sqn(fe5 a, int n) { while (n--) a = fe5_sqr(&a); return a; }
/* invert, k256/src/arithmetic/field.rs:187-216 */
static fe5 fe5_pow_prefix(const fe5* a, fe5* x22, fe5* x2o) {
  fe5 x2, x3, x6, x9, x11, x44, x88, x176, x220, x223, t;
  t = fe5_sqr(a); x2 = fe5_mul(&t, a);
  t = fe5_sqr(&x2); x3 = fe5_mul(&t, a);
  t = fe5_sqn(x3, 3); x6 = fe5_mul(&t, &x3);
  t = fe5_sqn(x6, 3); x9 = fe5_mul(&t, &x3);
  t = fe5_sqn(x9, 2); x11 = fe5_mul(&t, &x2);
  t = fe5_sqn(x11, 11); *x22 = fe5_mul(&t, &x11);
  t = fe5_sqn(*x22, 22); x44 = fe5_mul(&t, x22);
  t = fe5_sqn(x44, 44); x88 = fe5_mul(&t, &x44);
  t = fe5_sqn(x88, 88); x176 = fe5_mul(&t, &x88);
  t = fe5_sqn(x176, 44); x220 = fe5_mul(&t, &x44);
  t = fe5_sqn(x220, 3); x223 = fe5_mul(&t, &x3);
  *x2o = x2;
  return x223;
}
static fe5 fe5_invert(const fe5* a) {
  fe5 x22, x2, t;
  fe5 x223 = fe5_pow_prefix(a, &x22, &x2);
  t = fe5_sqn(x223, 23); t = fe5_mul(&t, &x22);
  t = fe5_sqn(t, 5); t = fe5_mul(&t, a);
  t = fe5_sqn(t, 3); t = fe5_mul(&t, &x2);
  t = fe5_sqn(t, 2); return fe5_mul(&t, a);
}
static int fe5_sqrt(fe5* r, const fe5* a) { /* field.rs:220-255 */
  fe5 x22, x2, t;
  fe5 x223 = fe5_pow_prefix(a, &x22, &x2);
  t = fe5_sqn(x223, 23); t = fe5_mul(&t, &x22);
  t = fe5_sqn(t, 6); t = fe5_mul(&t, &x2);
  *r = fe5_sqn(t, 2);
  fe5 chk = fe5_sqr(r), na = fe5_negate(a, 1);
  chk = fe5_add(&chk, &na);
  return fe5_normalizes_to_zero(&chk);
}

/* ---------------- k256 points (projective.rs) --------------------------------------------- */
typedef struct { fe5 x, y, z; } k256_pt;
static const fe5 FE5_ZERO = {{0, 0, 0, 0, 0}}, FE5_ONE = {{1, 0, 0, 0, 0}};
static k256_pt k256_identity(void) { k256_pt p = {FE5_ZERO, FE5_ONE, FE5_ZERO}; return p; }
static k256_pt k256_neg(const k256_pt* p) { /* :87-93 */
  k256_pt r = *p;
  fe5 n = fe5_negate(&p->y, 1);
  r.y = fe5_normalize_weak(&n);
  return r;
}
static k256_pt k256_add(const k256_pt* p, const k256_pt* q) { /* :96-161 */
  fe5 xx = fe5_mul(&p->x, &q->x), yy = fe5_mul(&p->y, &q->y), zz = fe5_mul(&p->z, &q->z);
  fe5 t, u, v;
  t = fe5_add(&xx, &yy); fe5 n_xx_yy = fe5_negate(&t, 2);
  t = fe5_add(&yy, &zz); fe5 n_yy_zz = fe5_negate(&t, 2);
  t = fe5_add(&xx, &zz); fe5 n_xx_zz = fe5_negate(&t, 2);
  u = fe5_add(&p->x, &p->y); v = fe5_add(&q->x, &q->y); t = fe5_mul(&u, &v); fe5 xy_pairs = fe5_add(&t, &n_xx_yy);
  u = fe5_add(&p->y, &p->z); v = fe5_add(&q->y, &q->z); t = fe5_mul(&u, &v); fe5 yz_pairs = fe5_add(&t, &n_yy_zz);
  u = fe5_add(&p->x, &p->z); v = fe5_add(&q->x, &q->z); t = fe5_mul(&u, &v); fe5 xz_pairs = fe5_add(&t, &n_xx_zz);
  fe5 bzz = fe5_mul_single(&zz, 7);
  t = fe5_double(&bzz); t = fe5_add(&t, &bzz); fe5 bzz3 = fe5_normalize_weak(&t);
  t = fe5_negate(&bzz3, 1); fe5 yy_m_bzz3 = fe5_add(&yy, &t);
  fe5 yy_p_bzz3 = fe5_add(&yy, &bzz3);
  t = fe5_mul_single(&yz_pairs, 7); fe5 byz = fe5_normalize_weak(&t);
  t = fe5_double(&byz); t = fe5_add(&t, &byz); fe5 byz3 = fe5_normalize_weak(&t);
  t = fe5_double(&xx); fe5 xx3 = fe5_add(&t, &xx);
  t = fe5_double(&xx3); t = fe5_add(&t, &xx3); t = fe5_normalize_weak(&t); t = fe5_mul_single(&t, 7); fe5 bxx9 = fe5_normalize_weak(&t);
  k256_pt r;
  u = fe5_mul(&xy_pairs, &yy_m_bzz3); v = fe5_mul(&byz3, &xz_pairs); v = fe5_negate(&v, 1); t = fe5_add(&u, &v); r.x = fe5_normalize_weak(&t);
  u = fe5_mul(&yy_p_bzz3, &yy_m_bzz3); v = fe5_mul(&bxx9, &xz_pairs); t = fe5_add(&u, &v); r.y = fe5_normalize_weak(&t);
  u = fe5_mul(&yz_pairs, &yy_p_bzz3); v = fe5_mul(&xx3, &xy_pairs); t = fe5_add(&u, &v); r.z = fe5_normalize_weak(&t);
  return r;
}
/* add_mixed, projective.rs:164-221 (RCB Algorithm 8); the affine operand is (x, y, infinity) */
static k256_pt k256_add_mixed(const k256_pt* p, const fe5* qx, const fe5* qy, int q_inf) {
  fe5 xx = fe5_mul(&p->x, qx), yy = fe5_mul(&p->y, qy);
  fe5 t, u, v;
  t = fe5_add(&xx, &yy); fe5 n_xx_yy = fe5_negate(&t, 2);
  u = fe5_add(&p->x, &p->y); v = fe5_add(qx, qy); t = fe5_mul(&u, &v); fe5 xy_pairs = fe5_add(&t, &n_xx_yy);
  t = fe5_mul(qy, &p->z); fe5 yz_pairs = fe5_add(&t, &p->y);
  t = fe5_mul(qx, &p->z); fe5 xz_pairs = fe5_add(&t, &p->x);
  fe5 bzz = fe5_mul_single(&p->z, 7);
  t = fe5_double(&bzz); t = fe5_add(&t, &bzz); fe5 bzz3 = fe5_normalize_weak(&t);
  t = fe5_negate(&bzz3, 1); fe5 yy_m_bzz3 = fe5_add(&yy, &t);
  fe5 yy_p_bzz3 = fe5_add(&yy, &bzz3);
  t = fe5_mul_single(&yz_pairs, 7); fe5 byz = fe5_normalize_weak(&t);
  t = fe5_double(&byz); t = fe5_add(&t, &byz); fe5 byz3 = fe5_normalize_weak(&t);
  t = fe5_double(&xx); fe5 xx3 = fe5_add(&t, &xx);
  t = fe5_double(&xx3); t = fe5_add(&t, &xx3); t = fe5_normalize_weak(&t); t = fe5_mul_single(&t, 7); fe5 bxx9 = fe5_normalize_weak(&t);
  k256_pt r;
  u = fe5_mul(&xy_pairs, &yy_m_bzz3); v = fe5_mul(&byz3, &xz_pairs); v = fe5_negate(&v, 1); t = fe5_add(&u, &v); r.x = fe5_normalize_weak(&t);
  u = fe5_mul(&yy_p_bzz3, &yy_m_bzz3); v = fe5_mul(&bxx9, &xz_pairs); t = fe5_add(&u, &v); r.y = fe5_normalize_weak(&t);
  u = fe5_mul(&yz_pairs, &yy_p_bzz3); v = fe5_mul(&xx3, &xy_pairs); t = fe5_add(&u, &v); r.z = fe5_normalize_weak(&t);
  return q_inf ? *p : r;                       /* :219 conditional_assign(self, other.is_identity()) */
}
static k256_pt k256_double(const k256_pt* p) { /* :225-274 */
  fe5 yy = fe5_sqr(&p->y), zz = fe5_sqr(&p->z);
  fe5 t = fe5_mul(&p->x, &p->y); fe5 xy2 = fe5_double(&t);
  fe5 bzz = fe5_mul_single(&zz, 7);
  t = fe5_double(&bzz); t = fe5_add(&t, &bzz); fe5 bzz3 = fe5_normalize_weak(&t);
  t = fe5_double(&bzz3); t = fe5_add(&t, &bzz3); fe5 bzz9 = fe5_normalize_weak(&t);
  t = fe5_negate(&bzz9, 1); fe5 yy_m_bzz9 = fe5_add(&yy, &t);
  fe5 yy_p_bzz3 = fe5_add(&yy, &bzz3);
  fe5 yy_zz = fe5_mul(&yy, &zz);
  t = fe5_double(&yy_zz); t = fe5_double(&t); fe5 yy_zz8 = fe5_double(&t);
  t = fe5_double(&yy_zz8); t = fe5_add(&t, &yy_zz8); t = fe5_normalize_weak(&t); fe5 tt = fe5_mul_single(&t, 7);
  k256_pt r;
  r.x = fe5_mul(&xy2, &yy_m_bzz9);
  t = fe5_mul(&yy_m_bzz9, &yy_p_bzz3); t = fe5_add(&t, &tt); r.y = fe5_normalize_weak(&t);
  t = fe5_mul(&yy, &p->y); t = fe5_mul(&t, &p->z); t = fe5_double(&t); t = fe5_double(&t); t = fe5_double(&t); r.z = fe5_normalize_weak(&t);
  return r;
}
static const u8 K256_BETA[32] = {0x7a, 0xe9, 0x6a, 0x2b, 0x65, 0x7c, 0x07, 0x10, 0x6e, 0x64, 0x47, 0x9e, 0xac, 0x34, 0x34, 0xe9,
                                 0x9c, 0xf0, 0x49, 0x75, 0x12, 0xf5, 0x89, 0x95, 0xc1, 0x39, 0x6c, 0x28, 0x71, 0x95, 0x01, 0xee};
static k256_pt k256_endomorphism(const k256_pt* p) { /* :287-293 */
  fe5 b; fe5_from_bytes(&b, K256_BETA);
  k256_pt r = *p; r.x = fe5_mul(&p->x, &b);
  return r;
}
static void k256_ct_select(k256_pt* t, const k256_pt* a, u64 mask) { /* conditional_assign */
  for (int i = 0; i < 5; i++) {
    t->x.n[i] ^= mask & (t->x.n[i] ^ a->x.n[i]);
    t->y.n[i] ^= mask & (t->y.n[i] ^ a->y.n[i]);
    t->z.n[i] ^= mask & (t->z.n[i] ^ a->z.n[i]);
  }
}

/* ---------------- k256 scalars: 4x64 canonical (scalar.rs, scalar/wide64.rs) ----------------- */
typedef struct { u64 w[4]; } sc4;
static const sc4 K256_N = {{0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}};
static void sc4_from_bytes(sc4* r, const u8* b) {
  for (int i = 0; i < 4; i++) { u64 v = 0; for (int j = 0; j < 8; j++) v = (v << 8) | b[8 * (3 - i) + j]; r->w[i] = v; }
}
static int sc4_geq(const sc4* a, const sc4* b) {
  for (int i = 3; i >= 0; i--) { if (a->w[i] > b->w[i]) return 1; if (a->w[i] < b->w[i]) return 0; }
  return 1;
}
static sc4 sc4_sub_raw(const sc4* a, const sc4* b) {
  sc4 r; u64 bw = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a->w[i] - b->w[i] - bw; r.w[i] = (u64)t; bw = (u64)(t >> 64) & 1; }
  return r;
}
static sc4 sc4_add_mod(const sc4* a, const sc4* b) { /* Scalar::add = add_mod */
  sc4 r; u64 c = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a->w[i] + b->w[i] + c; r.w[i] = (u64)t; c = (u64)(t >> 64); }
  if (c || sc4_geq(&r, &K256_N)) r = sc4_sub_raw(&r, &K256_N);
  return r;
}
static sc4 sc4_neg(const sc4* a) {
  sc4 z = {{0, 0, 0, 0}};
  if ((a->w[0] | a->w[1] | a->w[2] | a->w[3]) == 0) return z;
  return sc4_sub_raw(&K256_N, a);
}
static void sc4_mul_wide(u64* l, const sc4* a, const sc4* b) { /* wide64.rs:23-59 */
  u64 c0 = 0, c1 = 0, c2 = 0;
  for (int k = 0; k < 7; k++) {
    for (int i = 0; i < 4; i++) {
      int j = k - i;
      if (j < 0 || j > 3) continue;
      u128 t = (u128)a->w[i] * b->w[j];
      u64 tl = (u64)t, th = (u64)(t >> 64);
      c0 += tl; th += (c0 < tl); c1 += th; c2 += (c1 < th);
    }
    l[k] = c0; c0 = c1; c1 = c2; c2 = 0;
  }
  l[7] = c0;
}
/* 512 -> 256 bit reduction mod n: same fold by NEG_MODULUS = 2^256 - n as wide64.rs:121-212 */
static sc4 sc4_reduce512(const u64* l) {
  const u64 nm[3] = {~K256_N.w[0] + 1, ~K256_N.w[1], 1}; /* 2^256 - n, 129 bits */
  u64 t[9];
  for (int i = 0; i < 8; i++) t[i] = l[i];
  t[8] = 0;
  /* fold the high words three times: 512 -> 385 -> 258 -> 256(+1) bits */
  for (int round = 0; round < 3; round++) {
    u64 hi[5] = {t[4], t[5], t[6], t[7], t[8]};
    u64 acc[9] = {t[0], t[1], t[2], t[3], 0, 0, 0, 0, 0};
    for (int i = 0; i < 5; i++) {
      u64 carry = 0;
      for (int j = 0; j < 3; j++) {
        u128 p = (u128)hi[i] * nm[j] + acc[i + j] + carry;
        acc[i + j] = (u64)p; carry = (u64)(p >> 64);
      }
      for (int k2 = i + 3; carry && k2 < 9; k2++) { u128 p = (u128)acc[k2] + carry; acc[k2] = (u64)p; carry = (u64)(p >> 64); }
    }
    memcpy(t, acc, sizeof(acc));
  }
  sc4 r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || sc4_geq(&r, &K256_N)) r = sc4_sub_raw(&r, &K256_N);
  return r;
}
static sc4 sc4_mul(const sc4* a, const sc4* b) { u64 l[8]; sc4_mul_wide(l, a, b); return sc4_reduce512(l); } /* scalar.rs:114-124 */
static sc4 sc4_mul_shift_384(const sc4* a, const sc4* b) { /* wide64.rs:64-119 */
  u64 l[8];
  sc4_mul_wide(l, a, b);
  sc4 r = {{l[6], l[7], 0, 0}};
  if ((l[5] >> 63) & 1) { sc4 one = {{1, 0, 0, 0}}; r = sc4_add_mod(&r, &one); }
  return r;
}
static int sc4_is_high(const sc4* a) { /* scalar.rs:519-523: a > (n-1)/2 */
  static const sc4 half = {{0xDFE92F46681B20A0ULL, 0x5D576E7357A4501DULL, 0xFFFFFFFFFFFFFFFFULL, 0x7FFFFFFFFFFFFFFFULL}};
  return sc4_geq(a, &half) && memcmp(a, &half, sizeof(sc4)) != 0;
}
static const sc4 K256_MINUS_LAMBDA = {{0xE0CFC810B51283CFULL, 0xA880B9FC8EC739C2ULL, 0x5AD9E3FD77ED9BA4ULL, 0xAC9C52B33FA3CF1FULL}};
static const sc4 K256_MINUS_B1 = {{0x6F547FA90ABFE4C3ULL, 0xE4437ED6010E8828ULL, 0, 0}};
static const sc4 K256_MINUS_B2 = {{0xD765CDA83DB1562CULL, 0x8A280AC50774346DULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}};
static const sc4 K256_G1 = {{0xE893209A45DBB031ULL, 0x3DAA8A1471E8CA7FULL, 0xE86C90E49284EB15ULL, 0x3086D221A7D46BCDULL}};
static const sc4 K256_G2 = {{0x1571B4AE8AC47F71ULL, 0x221208AC9DF506C6ULL, 0x6F547FA90ABFE4C4ULL, 0xE4437ED6010E8828ULL}};
static void k256_decompose(const sc4* k, sc4* r1, sc4* r2) { /* mul.rs:260-268 */
  sc4 c1 = sc4_mul_shift_384(k, &K256_G1); c1 = sc4_mul(&c1, &K256_MINUS_B1);
  sc4 c2 = sc4_mul_shift_384(k, &K256_G2); c2 = sc4_mul(&c2, &K256_MINUS_B2);
  *r2 = sc4_add_mod(&c1, &c2);
  sc4 t = sc4_mul(r2, &K256_MINUS_LAMBDA);
  *r1 = sc4_add_mod(k, &t);
}
static void radix16(int8_t* out, int D, const sc4* x) { /* mul.rs:274-305 */
  memset(out, 0, D);
  for (int i = 0; i < (D - 1) / 2; i++) {
    u8 byte = (u8)(x->w[i / 8] >> (8 * (i % 8)));
    out[2 * i] = byte & 0xf;
    out[2 * i + 1] = (byte >> 4) & 0xf;
  }
  for (int i = 0; i < D - 1; i++) {
    int8_t carry = (out[i] + 8) >> 4;
    out[i] -= carry << 4;
    out[i + 1] += carry;
  }
}
typedef struct { k256_pt p[8]; } k256_table;
static void k256_table_from(k256_table* t, const k256_pt* p) { /* mul.rs:65-73 */
  t->p[0] = *p;
  for (int j = 0; j < 7; j++) t->p[j + 1] = k256_add(p, &t->p[j]);
}
static k256_pt k256_table_select(const k256_table* t, int8_t x) { /* mul.rs:92-127 */
  int8_t xmask = x >> 7;
  u8 xabs = (u8)((x + xmask) ^ xmask);
  k256_pt r = k256_identity();
  for (int j = 1; j < 9; j++) k256_ct_select(&r, &t->p[j - 1], (u64)0 - (u64)(xabs == j));
  k256_pt n = k256_neg(&r);
  k256_ct_select(&r, &n, (u64)0 - (u64)(xmask & 1));
  return r;
}
static k256_pt k256_lincomb(const k256_pt* xs, const sc4* ks, int nterms) { /* mul.rs:342-393 */
  k256_table* tables = (k256_table*)malloc(sizeof(k256_table) * 2 * (size_t)nterms);
  int8_t* digits = (int8_t*)malloc((size_t)66 * (size_t)nterms);
  for (int i = 0; i < nterms; i++) {
    sc4 r1, r2;
    k256_decompose(&ks[i], &r1, &r2);
    k256_pt xb = k256_endomorphism(&xs[i]);
    int s1 = sc4_is_high(&r1), s2 = sc4_is_high(&r2);
    sc4 r1c = s1 ? sc4_neg(&r1) : r1, r2c = s2 ? sc4_neg(&r2) : r2;
    k256_pt a = s1 ? k256_neg(&xs[i]) : xs[i], b = s2 ? k256_neg(&xb) : xb;
    k256_table_from(&tables[2 * i], &a);
    k256_table_from(&tables[2 * i + 1], &b);
    radix16(digits + 66 * i, 33, &r1c);
    radix16(digits + 66 * i + 33, 33, &r2c);
  }
  k256_pt acc = k256_identity(), t;
  for (int c = 0; c < nterms; c++) {
    t = k256_table_select(&tables[2 * c], digits[66 * c + 32]); acc = k256_add(&acc, &t);
    t = k256_table_select(&tables[2 * c + 1], digits[66 * c + 33 + 32]); acc = k256_add(&acc, &t);
  }
  for (int i = 31; i >= 0; i--) {
    for (int j = 0; j < 4; j++) acc = k256_double(&acc);
    for (int c = 0; c < nterms; c++) {
      t = k256_table_select(&tables[2 * c], digits[66 * c + i]); acc = k256_add(&acc, &t);
      t = k256_table_select(&tables[2 * c + 1], digits[66 * c + 33 + i]); acc = k256_add(&acc, &t);
    }
  }
  free(tables); free(digits);
  return acc;
}
static const u8 K256_GX[32] = {0x79, 0xBE, 0x66, 0x7E, 0xF9, 0xDC, 0xBB, 0xAC, 0x55, 0xA0, 0x62, 0x95, 0xCE, 0x87, 0x0B, 0x07,
                               0x02, 0x9B, 0xFC, 0xDB, 0x2D, 0xCE, 0x28, 0xD9, 0x59, 0xF2, 0x81, 0x5B, 0x16, 0xF8, 0x17, 0x98};
static const u8 K256_GY[32] = {0x48, 0x3A, 0xDA, 0x77, 0x26, 0xA3, 0xC4, 0x65, 0x5D, 0xA4, 0xFB, 0xFC, 0x0E, 0x11, 0x08, 0xA8,
                               0xFD, 0x17, 0xB4, 0x48, 0xA6, 0x85, 0x54, 0x19, 0x9C, 0x47, 0xD0, 0x8F, 0xFB, 0x10, 0xD4, 0xB8};
static k256_table* k256_gen_table = NULL; /* GEN_LOOKUP_TABLE, mul.rs:396-413 */
static pthread_once_t k256_gen_once = PTHREAD_ONCE_INIT;
static void k256_gen_init(void) {
  k256_gen_table = (k256_table*)malloc(sizeof(k256_table) * 33);
  k256_pt g; fe5_from_bytes(&g.x, K256_GX); fe5_from_bytes(&g.y, K256_GY); g.z = FE5_ONE;
  for (int i = 0; i < 33; i++) {
    k256_table_from(&k256_gen_table[i], &g);
    for (int j = 0; j < 8; j++) g = k256_double(&g);
  }
}
static k256_pt k256_mul_by_generator(const sc4* k) { /* mul.rs:424-439 */
  pthread_once(&k256_gen_once, k256_gen_init);
  int8_t d[65];
  radix16(d, 65, k);
  k256_pt acc = k256_table_select(&k256_gen_table[32], d[64]), acc2 = k256_identity(), t;
  for (int i = 31; i >= 0; i--) {
    t = k256_table_select(&k256_gen_table[i], d[2 * i + 1]); acc2 = k256_add(&acc2, &t);
    t = k256_table_select(&k256_gen_table[i], d[2 * i]); acc = k256_add(&acc, &t);
  }
  for (int j = 0; j < 4; j++) acc2 = k256_double(&acc2);
  return k256_add(&acc, &acc2);
}
/* to_affine (projective.rs:73-84): out = x || y || inf */
static void k256_to_affine_bytes(u8* out, const k256_pt* p) {
  if (fe5_normalizes_to_zero(&p->z)) { memset(out, 0, 64); out[64] = 1; return; }
  fe5 zi = fe5_invert(&p->z);
  fe5 x = fe5_mul(&p->x, &zi), y = fe5_mul(&p->y, &zi);
  x = fe5_normalize(&x); y = fe5_normalize(&y);
  fe5_to_bytes(out, &x); fe5_to_bytes(out + 32, &y); out[64] = 0;
}
static void k256_pt_to_bytes(u8* out, const k256_pt* p) {
  fe5 t = fe5_normalize(&p->x); fe5_to_bytes(out, &t);
  t = fe5_normalize(&p->y); fe5_to_bytes(out + 32, &t);
  t = fe5_normalize(&p->z); fe5_to_bytes(out + 64, &t);
}
static k256_pt k256_pt_from_affine_bytes(const u8* xy) {
  int zero = 1;
  for (int i = 0; i < 64; i++) zero &= (xy[i] == 0);
  if (zero) return k256_identity();
  k256_pt p; fe5_from_bytes(&p.x, xy); fe5_from_bytes(&p.y, xy + 32); p.z = FE5_ONE;
  return p;
}

/* ======================================================================================== */
/* Montgomery fields on NL 64-bit limbs (p256: 4, p384: 6), values always fully reduced        */
/* ======================================================================================== */
#define MAXL 6
typedef struct { u64 w[MAXL]; } mfe;
typedef struct {
  int nl;            /* limbs */
  int nbytes;
  u64 p[MAXL], r2[MAXL], one[MAXL], b[MAXL] /* curve b, Montgomery form */, gx[MAXL], gy[MAXL];
  u64 n[MAXL];       /* group order */
  u64 minv;          /* -p^-1 mod 2^64 */
} mcurve;
static mcurve P256C, P384C;

static void mfe_add(const mcurve* c, mfe* r, const mfe* a, const mfe* b) { /* p256 field.rs:118-134 */
  u64 t[MAXL + 1], carry = 0;
  for (int i = 0; i < c->nl; i++) { u128 s = (u128)a->w[i] + b->w[i] + carry; t[i] = (u64)s; carry = (u64)(s >> 64); }
  t[c->nl] = carry;
  u64 d[MAXL], bw = 0;
  for (int i = 0; i < c->nl; i++) { u128 s = (u128)t[i] - c->p[i] - bw; d[i] = (u64)s; bw = (u64)(s >> 64) & 1; }
  int ge = (t[c->nl] >= bw);   /* no borrow out of the extra word */
  u64 m = ge ? ~(u64)0 : 0;
  if (t[c->nl] == 0 && bw) m = 0;
  for (int i = 0; i < c->nl; i++) r->w[i] = (d[i] & m) | (t[i] & ~m);
}
static void mfe_sub(const mcurve* c, mfe* r, const mfe* a, const mfe* b) { /* p256 field.rs:142-197 */
  u64 d[MAXL], bw = 0;
  for (int i = 0; i < c->nl; i++) { u128 s = (u128)a->w[i] - b->w[i] - bw; d[i] = (u64)s; bw = (u64)(s >> 64) & 1; }
  u64 m = bw ? ~(u64)0 : 0, carry = 0;
  for (int i = 0; i < c->nl; i++) { u128 s = (u128)d[i] + (c->p[i] & m) + carry; r->w[i] = (u64)s; carry = (u64)(s >> 64); }
}
/* a*b*R^-1 mod p: schoolbook product (p256 field.rs:293-319) then word-by-word Montgomery reduction.
 * P-256 takes the reference's shortcuts (field.rs:240-277): the Montgomery constant is 1, so the quotient digit of round
 * i is the word r_i itself; p[0] = 2^64 - 1 makes k*p[0] + r_i = k*2^64 (word i becomes zero and the carry into word
 * i + 1 is k); p[2] = 0 needs no product at all.  Two multiplications per round instead of five.
 * P-384 does the plain word-by-word reduction with m' = 0x100000001 - the work fiat_p384_mul does (p384_64.rs:146-868:
 * 36 products for a*b and 36 for the multiples of p). */
static void mfe_mul(const mcurve* c, mfe* r, const mfe* a, const mfe* b) {
  const int nl = c->nl;
  u64 t[2 * MAXL + 1];
  memset(t, 0, sizeof(t));
  for (int i = 0; i < nl; i++) {
    u64 carry = 0;
    for (int j = 0; j < nl; j++) { u128 s = (u128)a->w[i] * b->w[j] + t[i + j] + carry; t[i + j] = (u64)s; carry = (u64)(s >> 64); }
    t[i + nl] = carry;
  }
  u64 top = 0;
  if (c->minv == 1 && c->p[0] == ~(u64)0 && c->p[2] == 0) {   /* the P-256 base field: p' = 1, p[0] = 2^64 - 1, p[2] = 0 */
    u64 carry2 = 0;
    for (int i = 0; i < 4; i++) {
      const u64 k = t[i];
      u128 s = (u128)k * c->p[1] + t[i + 1] + k;                       /* mac(r1, r0, modulus[1], r0) */
      t[i + 1] = (u64)s; u64 carry = (u64)(s >> 64);
      s = (u128)t[i + 2] + carry;                                        /* adc(r2, 0, carry): p[2] = 0 */
      t[i + 2] = (u64)s; carry = (u64)(s >> 64);
      s = (u128)k * c->p[3] + t[i + 3] + carry;                          /* mac(r3, r0, modulus[3], carry) */
      t[i + 3] = (u64)s; carry = (u64)(s >> 64);
      s = (u128)t[i + 4] + carry2 + carry;                               /* adc(r4, carry2, carry) */
      t[i + 4] = (u64)s; carry2 = (u64)(s >> 64);
    }
    top = carry2;
  } else {
    for (int i = 0; i < nl; i++) {
      u64 m = t[i] * c->minv, carry = 0;
      for (int j = 0; j < nl; j++) { u128 s = (u128)m * c->p[j] + t[i + j] + carry; t[i + j] = (u64)s; carry = (u64)(s >> 64); }
      for (int k = i + nl; k < 2 * nl; k++) { u128 s = (u128)t[k] + carry; t[k] = (u64)s; carry = (u64)(s >> 64); }
      top += carry;
    }
  }
  u64 d[MAXL], bw = 0;
  for (int i = 0; i < nl; i++) { u128 s = (u128)t[nl + i] - c->p[i] - bw; d[i] = (u64)s; bw = (u64)(s >> 64) & 1; }
  int use_d = top || !bw;
  for (int i = 0; i < nl; i++) r->w[i] = use_d ? d[i] : t[nl + i];
}
static void mfe_from_bytes(const mcurve* c, mfe* r, const u8* b) { /* to_montgomery: * R^2 */
  mfe t; memset(&t, 0, sizeof(t));
  for (int i = 0; i < c->nl; i++) { u64 v = 0; for (int j = 0; j < 8; j++) v = (v << 8) | b[8 * (c->nl - 1 - i) + j]; t.w[i] = v; }
  mfe r2; memset(&r2, 0, sizeof(r2)); memcpy(r2.w, c->r2, sizeof(u64) * c->nl);
  mfe_mul(c, r, &t, &r2);
}
static void mfe_to_bytes(const mcurve* c, u8* b, const mfe* a) { /* to_canonical: * 1 */
  mfe one; memset(&one, 0, sizeof(one)); one.w[0] = 1;
  mfe t; mfe_mul(c, &t, a, &one);
  for (int i = 0; i < c->nl; i++) for (int j = 0; j < 8; j++) b[8 * (c->nl - 1 - i) + j] = (u8)(t.w[i] >> (56 - 8 * j));
}
static int mfe_is_zero(const mcurve* c, const mfe* a) { u64 z = 0; for (int i = 0; i < c->nl; i++) z |= a->w[i]; return z == 0; }
static void mfe_pow(const mcurve* c, mfe* r, const mfe* a, const u64* e) { /* left-to-right square-and-multiply (input synthesis only) */
  mfe acc; memset(&acc, 0, sizeof(acc)); memcpy(acc.w, c->one, sizeof(u64) * c->nl);
  for (int i = c->nl * 64 - 1; i >= 0; i--) {
    mfe_mul(c, &acc, &acc, &acc);
    if ((e[i / 64] >> (i % 64)) & 1) mfe_mul(c, &acc, &acc, a);
  }
  *r = acc;
}
/* acc = acc^(2^k) * m */
static void mfe_sqn_mul(const mcurve* c, mfe* acc, int k, const mfe* m) {
  while (k--) mfe_mul(c, acc, acc, acc);
  if (m) mfe_mul(c, acc, acc, m);
}
/* P-384 inversion as the reference does it: Bernstein-Yang divsteps (p384 field.rs:67-91 -> impl_bernstein_yang_invert,
 * primeorder/src/field.rs:505-559).  (49 * 384 + 57) / 17 = 1 110 iterations of
 *   divstep(delta, f, g, v, r):  if delta > 0 and g odd: (1 - delta, g, (g - f) / 2, 2 r, r - v)
 *                                else:                    (1 + delta, f, (g + (g mod 2) f) / 2, 2 v, r + (g mod 2) v)
 * (the postconditions of fiat_p384_divstep, p384_64.rs:3286-3313; f, g are 7-word two's-complement integers, v, r field
 * elements), starting from f = m, g = from_montgomery(a), v = 0, r = 1; then v := -v if f < 0 and the result is
 * v * precomp with precomp = ((m - 1) / 2)^1110 (fiat_p384_divstep_precomp), here computed once instead of tabulated.
 * Branch-free like fiat's: every step runs the same additions under masks. */
static mfe P384_BY_PRECOMP;
static pthread_once_t p384_by_once = PTHREAD_ONCE_INIT;
static void mfe_pow(const mcurve* c, mfe* r, const mfe* a, const u64* e);
static void p384_by_init(void) {
  const mcurve* c = &P384C;
  u64 h[MAXL];                                           /* (m - 1) / 2 as an integer */
  for (int i = 0; i < c->nl; i++) h[i] = (c->p[i] >> 1) | (i + 1 < c->nl ? c->p[i + 1] << 63 : 0);
  mfe hm, t; memset(&hm, 0, sizeof(hm)); memset(&t, 0, sizeof(t));
  memcpy(t.w, h, sizeof(u64) * c->nl);
  mfe r2; memset(&r2, 0, sizeof(r2)); memcpy(r2.w, c->r2, sizeof(u64) * c->nl);
  mfe_mul(c, &hm, &t, &r2);                              /* to Montgomery form */
  u64 e[MAXL] = {1110, 0, 0, 0, 0, 0};
  mfe_pow(c, &P384_BY_PRECOMP, &hm, e);
}
/* fixed-size helpers of the divstep loop (six-word field elements, seven-word integers) */
static inline void by_addmod6(u64* r, const u64* a, const u64* b, const u64* p) {      /* (a + b) mod p for a < p, b <= p */
  u64 t[7], d[6], carry = 0, bw = 0;
  for (int i = 0; i < 6; i++) { u128 s = (u128)a[i] + b[i] + carry; t[i] = (u64)s; carry = (u64)(s >> 64); }
  t[6] = carry;
  for (int i = 0; i < 6; i++) { u128 s = (u128)t[i] - p[i] - bw; d[i] = (u64)s; bw = (u64)(s >> 64) & 1; }
  const u64 m = 0 - (u64)((t[6] != 0) | (bw == 0));       /* t >= p */
  for (int i = 0; i < 6; i++) r[i] = (d[i] & m) | (t[i] & ~m);
}
static void mfe_invert_by(const mcurve* c, mfe* out, const mfe* a) {
  pthread_once(&p384_by_once, p384_by_init);
  u64 f[7], g[7], v[6], r[6], P[6];
  memcpy(P, c->p, sizeof(P));
  memcpy(f, c->p, 48); f[6] = 0;                         /* msat: the modulus as a 7-word integer */
  mfe one_plain; memset(&one_plain, 0, sizeof(one_plain)); one_plain.w[0] = 1;
  mfe ac; mfe_mul(c, &ac, a, &one_plain);                /* from_montgomery */
  memcpy(g, ac.w, 48); g[6] = 0;
  memset(v, 0, sizeof(v)); memcpy(r, c->one, 48);
  int64_t delta = 1;
  for (int it = 0; it < 1110; it++) {
    const u64 godd = 0 - (g[0] & 1);                      /* mask: g odd */
    const u64 swap = godd & (0 - (u64)(delta > 0));       /* mask: delta > 0 and g odd */
    /* t = g - f (swap) or g + (g odd ? f : 0); f' = swap ? g : f; g' = t >> 1 (arithmetic) */
    u64 t[7], cy = swap & 1;
    for (int i = 0; i < 7; i++) {
      const u64 addend = (swap & ~f[i]) | (~swap & godd & f[i]);
      u128 s2 = (u128)g[i] + addend + cy;
      t[i] = (u64)s2; cy = (u64)(s2 >> 64);
      f[i] = (swap & g[i]) | (~swap & f[i]);
    }
    for (int i = 0; i < 6; i++) g[i] = (t[i] >> 1) | (t[i + 1] << 63);
    g[6] = (u64)((int64_t)t[6] >> 1);
    /* v' = 2 (swap ? r : v);  r' = r + (g odd ? (swap ? -v : v) : 0)   (swap implies g odd) */
    u64 sel[6], w[6], bw = 0;
    for (int i = 0; i < 6; i++) { u128 d = (u128)P[i] - v[i] - bw; w[i] = (u64)d; bw = (u64)(d >> 64) & 1; }      /* p - v (= p for v = 0) */
    for (int i = 0; i < 6; i++) {
      sel[i] = (swap & r[i]) | (~swap & v[i]);
      w[i] = godd & ((swap & w[i]) | (~swap & v[i]));
    }
    by_addmod6(r, r, w, P);
    by_addmod6(v, sel, sel, P);
    delta = (swap ? 1 - delta : 1 + delta);
  }
  const u64 neg = 0 - (f[6] >> 63);                       /* f < 0: the inverse is -v */
  mfe vv, zero, nvv; memset(&zero, 0, sizeof(zero)); memset(&vv, 0, sizeof(vv)); memcpy(vv.w, v, 48);
  mfe_sub(c, &nvv, &zero, &vv);
  for (int i = 0; i < 6; i++) vv.w[i] = (neg & nvv.w[i]) | (~neg & vv.w[i]);
  mfe_mul(c, out, &vv, &P384_BY_PRECOMP);
}
/* a^(p-2), the unique inverse.  P-256: the addition chain of p256 field.rs:357-382 (255 squarings + 12 multiplications).
 * P-384: the Bernstein-Yang inversion above, as in the reference; the Fermat chain for p - 2 = [255 ones][0][32 ones]
 * [64 zeros][30 ones][0][1] (385 squarings + 14 multiplications) stays for cross-checking (eco_p384_invert with which = 1). */
static void mfe_invert_fermat(const mcurve* c, mfe* r, const mfe* a);
static void mfe_invert(const mcurve* c, mfe* r, const mfe* a) {
  if (c == &P384C) { mfe_invert_by(c, r, a); return; }
  mfe_invert_fermat(c, r, a);
}
static void mfe_invert_fermat(const mcurve* c, mfe* r, const mfe* a) {
  if (c != &P256C && c != &P384C) {                 /* the scalar fields (ECDSA): plain a^(n-2) */
    u64 e[MAXL]; memcpy(e, c->p, sizeof(e)); e[0] -= 2;
    mfe_pow(c, r, a, e);
    return;
  }
  mfe x2 = *a, x3, x6, x12, x15, t;
  mfe_sqn_mul(c, &x2, 1, a);
  x3 = x2; mfe_sqn_mul(c, &x3, 1, a);
  x6 = x3; mfe_sqn_mul(c, &x6, 3, &x3);
  x12 = x6; mfe_sqn_mul(c, &x12, 6, &x6);
  x15 = x12; mfe_sqn_mul(c, &x15, 3, &x3);
  if (c == &P256C) {
    mfe x16 = x15, x32, x47;
    mfe_sqn_mul(c, &x16, 1, a);
    x32 = x16; mfe_sqn_mul(c, &x32, 16, &x16);
    t = x32; mfe_sqn_mul(c, &t, 15, NULL);          /* i53 = x32 << 15 */
    mfe_mul(c, &x47, &t, &x15);
    mfe_sqn_mul(c, &t, 17, a);
    mfe_sqn_mul(c, &t, 143, &x47);
    mfe_sqn_mul(c, &t, 47, &x47);
    mfe_sqn_mul(c, &t, 2, a);
  } else {
    mfe x30 = x15, x32, x60, x120;
    mfe_sqn_mul(c, &x30, 15, &x15);
    x32 = x30; mfe_sqn_mul(c, &x32, 2, &x2);
    x60 = x30; mfe_sqn_mul(c, &x60, 30, &x30);
    x120 = x60; mfe_sqn_mul(c, &x120, 60, &x60);
    t = x120; mfe_sqn_mul(c, &t, 120, &x120);       /* x240 */
    mfe_sqn_mul(c, &t, 15, &x15);                   /* x255 */
    mfe_sqn_mul(c, &t, 33, &x32);
    mfe_sqn_mul(c, &t, 94, &x30);
    mfe_sqn_mul(c, &t, 2, a);
  }
  *r = t;
}

typedef struct { mfe x, y, z; } mpt;
static mpt mpt_identity(const mcurve* c) { mpt p; memset(&p, 0, sizeof(p)); memcpy(p.y.w, c->one, sizeof(u64) * c->nl); return p; }
#define MUL(r, a, b) mfe_mul(c, &(r), &(a), &(b))
#define ADD(r, a, b) mfe_add(c, &(r), &(a), &(b))
#define SUB(r, a, b) mfe_sub(c, &(r), &(a), &(b))
static mpt mpt_add(const mcurve* c, const mpt* p, const mpt* q) { /* point_arithmetic.rs:209-238 */
  mfe B; memset(&B, 0, sizeof(B)); memcpy(B.w, c->b, sizeof(u64) * c->nl);
  mfe xx, yy, zz, t0, t1, xy_pairs, yz_pairs, xz_pairs;
  MUL(xx, p->x, q->x); MUL(yy, p->y, q->y); MUL(zz, p->z, q->z);
  ADD(t0, p->x, p->y); ADD(t1, q->x, q->y); MUL(xy_pairs, t0, t1); ADD(t0, xx, yy); SUB(xy_pairs, xy_pairs, t0);
  ADD(t0, p->y, p->z); ADD(t1, q->y, q->z); MUL(yz_pairs, t0, t1); ADD(t0, yy, zz); SUB(yz_pairs, yz_pairs, t0);
  ADD(t0, p->x, p->z); ADD(t1, q->x, q->z); MUL(xz_pairs, t0, t1); ADD(t0, xx, zz); SUB(xz_pairs, xz_pairs, t0);
  mfe bzz_part, bzz3_part, yy_m_bzz3, yy_p_bzz3, zz3, bxz_part, bxz3_part, xx3_m_zz3;
  MUL(t0, B, zz); SUB(bzz_part, xz_pairs, t0);
  ADD(t0, bzz_part, bzz_part); ADD(bzz3_part, t0, bzz_part);
  SUB(yy_m_bzz3, yy, bzz3_part); ADD(yy_p_bzz3, yy, bzz3_part);
  ADD(t0, zz, zz); ADD(zz3, t0, zz);
  MUL(t0, B, xz_pairs); ADD(t1, zz3, xx); SUB(bxz_part, t0, t1);
  ADD(t0, bxz_part, bxz_part); ADD(bxz3_part, t0, bxz_part);
  ADD(t0, xx, xx); ADD(t0, t0, xx); SUB(xx3_m_zz3, t0, zz3);
  mpt r;
  MUL(t0, yy_p_bzz3, xy_pairs); MUL(t1, yz_pairs, bxz3_part); SUB(r.x, t0, t1);
  MUL(t0, yy_p_bzz3, yy_m_bzz3); MUL(t1, xx3_m_zz3, bxz3_part); ADD(r.y, t0, t1);
  MUL(t0, yy_m_bzz3, yz_pairs); MUL(t1, xy_pairs, xx3_m_zz3); ADD(r.z, t0, t1);
  return r;
}
static mpt mpt_add_mixed(const mcurve* c, const mpt* p, const mfe* qx, const mfe* qy, int q_inf) { /* point_arithmetic.rs:247-277 */
  mfe B; memset(&B, 0, sizeof(B)); memcpy(B.w, c->b, sizeof(u64) * c->nl);
  mfe xx, yy, t0, t1, xy_pairs, yz_pairs, xz_pairs;
  MUL(xx, p->x, *qx); MUL(yy, p->y, *qy);
  ADD(t0, p->x, p->y); ADD(t1, *qx, *qy); MUL(xy_pairs, t0, t1); ADD(t0, xx, yy); SUB(xy_pairs, xy_pairs, t0);
  MUL(t0, *qy, p->z); ADD(yz_pairs, t0, p->y);
  MUL(t0, *qx, p->z); ADD(xz_pairs, t0, p->x);
  mfe bz_part, bz3_part, yy_m_bzz3, yy_p_bzz3, z3, bxz_part, bxz3_part, xx3_m_zz3;
  MUL(t0, B, p->z); SUB(bz_part, xz_pairs, t0);
  ADD(t0, bz_part, bz_part); ADD(bz3_part, t0, bz_part);
  SUB(yy_m_bzz3, yy, bz3_part); ADD(yy_p_bzz3, yy, bz3_part);
  ADD(t0, p->z, p->z); ADD(z3, t0, p->z);
  MUL(t0, B, xz_pairs); ADD(t1, z3, xx); SUB(bxz_part, t0, t1);
  ADD(t0, bxz_part, bxz_part); ADD(bxz3_part, t0, bxz_part);
  ADD(t0, xx, xx); ADD(t0, t0, xx); SUB(xx3_m_zz3, t0, z3);
  mpt r;
  MUL(t0, yy_p_bzz3, xy_pairs); MUL(t1, yz_pairs, bxz3_part); SUB(r.x, t0, t1);
  MUL(t0, yy_p_bzz3, yy_m_bzz3); MUL(t1, xx3_m_zz3, bxz3_part); ADD(r.y, t0, t1);
  MUL(t0, yy_m_bzz3, yz_pairs); MUL(t1, xy_pairs, xx3_m_zz3); ADD(r.z, t0, t1);
  return q_inf ? *p : r;                       /* :275 conditional_assign(lhs, rhs.is_identity()) */
}
static mpt mpt_double(const mcurve* c, const mpt* p) { /* point_arithmetic.rs:286-317 */
  mfe B; memset(&B, 0, sizeof(B)); memcpy(B.w, c->b, sizeof(u64) * c->nl);
  mfe xx, yy, zz, xy2, xz2, t0, t1;
  MUL(xx, p->x, p->x); MUL(yy, p->y, p->y); MUL(zz, p->z, p->z);
  MUL(t0, p->x, p->y); ADD(xy2, t0, t0);
  MUL(t0, p->x, p->z); ADD(xz2, t0, t0);
  mfe bzz_part, bzz3_part, yy_m_bzz3, yy_p_bzz3, y_frag, x_frag, zz3, bxz2_part, bxz6_part, xx3_m_zz3, yz2;
  MUL(t0, B, zz); SUB(bzz_part, t0, xz2);
  ADD(t0, bzz_part, bzz_part); ADD(bzz3_part, t0, bzz_part);
  SUB(yy_m_bzz3, yy, bzz3_part); ADD(yy_p_bzz3, yy, bzz3_part);
  MUL(y_frag, yy_p_bzz3, yy_m_bzz3); MUL(x_frag, yy_m_bzz3, xy2);
  ADD(t0, zz, zz); ADD(zz3, t0, zz);
  MUL(t0, B, xz2); ADD(t1, zz3, xx); SUB(bxz2_part, t0, t1);
  ADD(t0, bxz2_part, bxz2_part); ADD(bxz6_part, t0, bxz2_part);
  ADD(t0, xx, xx); ADD(t0, t0, xx); SUB(xx3_m_zz3, t0, zz3);
  mpt r;
  MUL(t0, xx3_m_zz3, bxz6_part); ADD(r.y, y_frag, t0);
  MUL(t0, p->y, p->z); ADD(yz2, t0, t0);
  MUL(t0, bxz6_part, yz2); SUB(r.x, x_frag, t0);
  MUL(t0, yz2, yy); ADD(t0, t0, t0); ADD(r.z, t0, t0);
  return r;
}
static mpt mpt_mul(const mcurve* c, const mpt* p, const u8* k_be) { /* primeorder/src/projective.rs:106-150 */
  mpt pc[16];
  pc[0] = mpt_identity(c); pc[1] = *p;
  for (int i = 2; i < 16; i++) pc[i] = (i % 2 == 0) ? mpt_double(c, &pc[i / 2]) : mpt_add(c, &pc[i - 1], p);
  mpt q = mpt_identity(c);
  int pos = c->nbytes * 8 - 4;
  for (;;) {
    int byte = k_be[c->nbytes - 1 - (pos >> 3)];
    unsigned slot = (byte >> (pos & 7)) & 0xf;
    mpt t = mpt_identity(c);
    for (unsigned i = 1; i < 16; i++) { /* constant-time scan, :132-137 */
      u64 m = (u64)0 - (u64)(((slot ^ i) - 1) >> 8 & 1);
      for (int w = 0; w < c->nl; w++) {
        t.x.w[w] ^= m & (t.x.w[w] ^ pc[i].x.w[w]); t.y.w[w] ^= m & (t.y.w[w] ^ pc[i].y.w[w]); t.z.w[w] ^= m & (t.z.w[w] ^ pc[i].z.w[w]);
      }
    }
    q = mpt_add(c, &q, &t);
    if (pos == 0) break;
    q = mpt_double(c, &q); q = mpt_double(c, &q); q = mpt_double(c, &q); q = mpt_double(c, &q);
    pos -= 4;
  }
  return q;
}
static void mpt_to_affine_bytes(const mcurve* c, u8* out, const mpt* p) { /* projective.rs:62-74 */
  int nb = c->nbytes;
  if (mfe_is_zero(c, &p->z)) { memset(out, 0, 2 * nb); out[2 * nb] = 1; return; }
  mfe zi, x, y;
  mfe_invert(c, &zi, &p->z);
  MUL(x, p->x, zi); MUL(y, p->y, zi);
  mfe_to_bytes(c, out, &x); mfe_to_bytes(c, out + nb, &y); out[2 * nb] = 0;
}
static mpt mpt_from_affine_bytes(const mcurve* c, const u8* xy) {
  int zero = 1;
  for (int i = 0; i < 2 * c->nbytes; i++) zero &= (xy[i] == 0);
  if (zero) return mpt_identity(c);
  mpt p; memset(&p, 0, sizeof(p));
  mfe_from_bytes(c, &p.x, xy); mfe_from_bytes(c, &p.y, xy + c->nbytes); memcpy(p.z.w, c->one, sizeof(u64) * c->nl);
  return p;
}

static void hex_to_words(u64* w, int nl, const char* hex) {
  memset(w, 0, sizeof(u64) * MAXL);
  int len = (int)strlen(hex);
  for (int i = 0; i < len; i++) {
    char ch = hex[len - 1 - i];
    u64 v = (ch >= '0' && ch <= '9') ? ch - '0' : (ch >= 'a' && ch <= 'f') ? ch - 'a' + 10 : ch - 'A' + 10;
    w[i / 16] |= v << (4 * (i % 16));
  }
  (void)nl;
}
static void mcurve_init(mcurve* c, int nl, const char* p, const char* n, const char* b, const char* gx, const char* gy) {
  memset(c, 0, sizeof(*c));
  c->nl = nl; c->nbytes = nl * 8;
  hex_to_words(c->p, nl, p); hex_to_words(c->n, nl, n);
  u64 inv = 1; /* Newton: inv = p^-1 mod 2^64 */
  for (int i = 0; i < 6; i++) inv *= 2 - c->p[0] * inv;
  c->minv = (u64)0 - inv;
  /* R mod p and R^2 mod p by repeated doubling */
  mfe x; memset(&x, 0, sizeof(x)); x.w[0] = 1;
  for (int i = 0; i < 64 * nl; i++) mfe_add(c, &x, &x, &x);
  memcpy(c->one, x.w, sizeof(u64) * nl);
  for (int i = 0; i < 64 * nl; i++) mfe_add(c, &x, &x, &x);
  memcpy(c->r2, x.w, sizeof(u64) * nl);
  u64 raw[MAXL]; mfe t, r2; memset(&r2, 0, sizeof(r2)); memcpy(r2.w, c->r2, sizeof(u64) * nl);
  hex_to_words(raw, nl, b); memset(&t, 0, sizeof(t)); memcpy(t.w, raw, sizeof(u64) * nl); mfe_mul(c, &t, &t, &r2); memcpy(c->b, t.w, sizeof(u64) * nl);
  hex_to_words(raw, nl, gx); memset(&t, 0, sizeof(t)); memcpy(t.w, raw, sizeof(u64) * nl); mfe_mul(c, &t, &t, &r2); memcpy(c->gx, t.w, sizeof(u64) * nl);
  hex_to_words(raw, nl, gy); memset(&t, 0, sizeof(t)); memcpy(t.w, raw, sizeof(u64) * nl); mfe_mul(c, &t, &t, &r2); memcpy(c->gy, t.w, sizeof(u64) * nl);
}
static pthread_once_t curves_once = PTHREAD_ONCE_INIT;
static void curves_init(void) {
  mcurve_init(&P256C, 4, "ffffffff00000001000000000000000000000000ffffffffffffffffffffffff",
              "ffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551",
              "5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b",
              "6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296",
              "4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5");
  mcurve_init(&P384C, 6, "fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffeffffffff0000000000000000ffffffff",
              "ffffffffffffffffffffffffffffffffffffffffffffffffc7634d81f4372ddf581a0db248b0a77aecec196accc52973",
              "b3312fa7e23ee7e4988e056be3f82d19181d9c6efe8141120314088f5013875ac656398d8a2ed19d2a85c8edd3ec2aef",
              "aa87ca22be8b05378eb1c71ef320ad746e1d3b628ba79b9859f741e082542a385502f25dbf55296c3a545e3872760ab7",
              "3617de4a96262c6f5d9e98bf9292dc29f8f41dbd289a147ce9da3113b5f0b8c00a60b1ce1d7e819d7a431d7c90ea0e5f");
}
static const mcurve* get_mcurve(int curve) {
  pthread_once(&curves_once, curves_init);
  return curve == 1 ? &P256C : curve == 2 ? &P384C : NULL;
}

/* ======================================================================================== */
/* synthetic inputs (oracle/synth.py is the specification)                                     */
/* ======================================================================================== */
static u64 synth_word(u64 seed, u64 stream, u64 index, unsigned j) {
  u64 z = (seed ^ (stream * 0xD1342543DE82EF95ULL)) + (8 * index + j + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static void synth_value_be(u8* out, int nbytes, u64 seed, u64 stream, u64 index) {
  for (int j = 0; j < nbytes / 8; j++) {
    u64 w = synth_word(seed, stream, index, j);
    for (int b = 0; b < 8; b++) out[8 * j + b] = (u8)(w >> (56 - 8 * b));
  }
}
static int be_geq(const u8* a, const u8* b, int n) { return memcmp(a, b, n) >= 0; }
static void be_sub(u8* a, const u8* b, int n) {
  int bw = 0;
  for (int i = n - 1; i >= 0; i--) { int d = a[i] - b[i] - bw; bw = d < 0; a[i] = (u8)(d + (bw << 8)); }
}
static void curve_moduli(int curve, u8* p, u8* n, int* nbytes) {
  static const char* P[3] = {"fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2f",
                             "ffffffff00000001000000000000000000000000ffffffffffffffffffffffff",
                             "fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffeffffffff0000000000000000ffffffff"};
  static const char* N[3] = {"fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141",
                             "ffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551",
                             "ffffffffffffffffffffffffffffffffffffffffffffffffc7634d81f4372ddf581a0db248b0a77aecec196accc52973"};
  int nb = curve == 2 ? 48 : 32;
  *nbytes = nb;
  for (int i = 0; i < nb; i++) {
    unsigned v; char h[3] = {P[curve][2 * i], P[curve][2 * i + 1], 0}; v = (unsigned)strtoul(h, NULL, 16); p[i] = (u8)v;
    char g[3] = {N[curve][2 * i], N[curve][2 * i + 1], 0}; v = (unsigned)strtoul(g, NULL, 16); n[i] = (u8)v;
  }
}

/* ======================================================================================== */
/* exported API                                                                               */
/* ======================================================================================== */
typedef struct {
  int curve, op, terms;
  const u8 *scalars, *points;
  u8* out;
  size_t begin, end;
  int out_proj;
} job_t;

/* one variable-base (points != NULL) or fixed-base multiplication, reference algorithm */
static void do_mul_range(job_t* j) {
  const int nb = j->curve == 2 ? 48 : 32;
  const size_t ostride = j->out_proj ? 3 * nb : 2 * nb + 1;
  for (size_t i = j->begin; i < j->end; i++) {
    u8* o = j->out + i * ostride;
    if (j->curve == 0) {
      sc4 ks[2]; k256_pt ps[2], r;
      for (int t = 0; t < j->terms; t++) {
        sc4_from_bytes(&ks[t], j->scalars + (i * j->terms + t) * 32);
        if (sc4_geq(&ks[t], &K256_N)) ks[t] = sc4_sub_raw(&ks[t], &K256_N);
        if (j->points) ps[t] = k256_pt_from_affine_bytes(j->points + (i * j->terms + t) * 64);
      }
      r = j->points ? k256_lincomb(ps, ks, j->terms) : k256_mul_by_generator(&ks[0]);
      if (j->out_proj) k256_pt_to_bytes(o, &r); else k256_to_affine_bytes(o, &r);
    } else {
      const mcurve* c = get_mcurve(j->curve);
      mpt acc = mpt_identity(c);
      for (int t = 0; t < j->terms; t++) {
        mpt p;
        if (j->points) p = mpt_from_affine_bytes(c, j->points + (i * j->terms + t) * 2 * nb);
        else { memset(&p, 0, sizeof(p)); memcpy(p.x.w, c->gx, 8 * c->nl); memcpy(p.y.w, c->gy, 8 * c->nl); memcpy(p.z.w, c->one, 8 * c->nl); }
        mpt r = mpt_mul(c, &p, j->scalars + (i * j->terms + t) * nb);
        acc = (t == 0) ? r : mpt_add(c, &acc, &r);   /* default lincomb: x*k + y*l, primeorder projective.rs:415-420 */
      }
      if (j->out_proj) { mfe_to_bytes(c, o, &acc.x); mfe_to_bytes(c, o + nb, &acc.y); mfe_to_bytes(c, o + 2 * nb, &acc.z); }
      else mpt_to_affine_bytes(c, o, &acc);
    }
  }
}
static void* worker(void* a) { do_mul_range((job_t*)a); return NULL; }

/* out: affine x||y||inf (2*nb+1 bytes per element) or projective X||Y||Z when out_proj.
 * points: affine x||y (identity = zeros) or NULL for the generator.  Scalars must be < n
 * (k256 reduces once like Reduce<U256>::reduce). */
int eco_lincomb_batch(int curve, const u8* scalars, const u8* points, int terms, u8* out, size_t n, int out_proj, int threads) {
  if (curve < 0 || curve > 2 || terms < 1 || terms > 2) return -1;
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = (int)(n ? n : 1);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  job_t* jobs = (job_t*)malloc(sizeof(job_t) * threads);
  if (curve == 0 && !points) pthread_once(&k256_gen_once, k256_gen_init);
  for (int t = 0; t < threads; t++) {
    job_t j = {curve, 0, terms, scalars, points, out, n * t / threads, n * (t + 1) / threads, out_proj};
    jobs[t] = j;
    if (threads == 1) do_mul_range(&jobs[t]); else pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(th); free(jobs);
  return 0;
}

/* sum_i scalars[i] * points[i] by the plain definition (one reference mul per term + complete adds).
 * out = affine x||y||inf. */
int eco_msm_naive(int curve, const u8* scalars, const u8* points, size_t n, u8* out) {
  const int nb = curve == 2 ? 48 : 32;
  if (curve == 0) {
    k256_pt acc = k256_identity();
    for (size_t i = 0; i < n; i++) {
      sc4 k; sc4_from_bytes(&k, scalars + 32 * i);
      if (sc4_geq(&k, &K256_N)) k = sc4_sub_raw(&k, &K256_N);
      k256_pt p = k256_pt_from_affine_bytes(points + 64 * i);
      k256_pt r = k256_lincomb(&p, &k, 1);
      acc = k256_add(&acc, &r);
    }
    k256_to_affine_bytes(out, &acc);
    return 0;
  }
  const mcurve* c = get_mcurve(curve);
  if (!c) return -1;
  mpt acc = mpt_identity(c);
  for (size_t i = 0; i < n; i++) {
    mpt p = mpt_from_affine_bytes(c, points + 2 * nb * i);
    mpt r = mpt_mul(c, &p, scalars + nb * i);
    acc = mpt_add(c, &acc, &r);
  }
  mpt_to_affine_bytes(c, out, &acc);
  return 0;
}

int eco_synth_scalars(int curve, u64 seed, u64 first, u8* out, size_t n) {
  u8 p[48], ord[48]; int nb;
  if (curve < 0 || curve > 2) return -1;
  curve_moduli(curve, p, ord, &nb);
  for (size_t i = 0; i < n; i++) {
    u8* o = out + i * nb;
    synth_value_be(o, nb, seed, 0, first + i);
    if (be_geq(o, ord, nb)) be_sub(o, ord, nb);
  }
  return 0;
}
/* try-and-increment decompress (k256 affine.rs:184-202, primeorder affine.rs:129-150) */
int eco_synth_points(int curve, u64 seed, u64 first, u8* out_xy, size_t n) {
  u8 p[48], ord[48]; int nb;
  if (curve < 0 || curve > 2) return -1;
  curve_moduli(curve, p, ord, &nb);
  const mcurve* c = curve ? get_mcurve(curve) : NULL;
  for (size_t i = 0; i < n; i++) {
    u8* o = out_xy + i * 2 * nb;
    for (int t = 0; t < 64; t++) {
      u8 xb[48];
      synth_value_be(xb, nb, seed, 1 + t, first + i);
      if (be_geq(xb, p, nb)) be_sub(xb, p, nb);
      int want_odd = (int)(synth_word(seed, 1 + t, first + i, 7) & 1);
      if (curve == 0) {
        fe5 x, y, rhs, seven = {{7, 0, 0, 0, 0}};
        fe5_from_bytes(&x, xb);
        rhs = fe5_sqr(&x); rhs = fe5_mul(&rhs, &x); rhs = fe5_add(&rhs, &seven);
        if (!fe5_sqrt(&y, &rhs)) continue;
        y = fe5_normalize(&y);
        if ((int)(y.n[0] & 1) != want_odd) { y = fe5_negate(&y, 1); y = fe5_normalize(&y); }
        memcpy(o, xb, 32); fe5_to_bytes(o + 32, &y);
        break;
      } else {
        mfe x, y, rhs, t3, three; u64 e[MAXL];
        mfe_from_bytes(c, &x, xb);
        mfe_mul(c, &rhs, &x, &x); mfe_mul(c, &rhs, &rhs, &x);
        mfe_add(c, &t3, &x, &x); mfe_add(c, &three, &t3, &x);  /* 3x */
        mfe_sub(c, &rhs, &rhs, &three);
        mfe B; memset(&B, 0, sizeof(B)); memcpy(B.w, c->b, sizeof(u64) * (size_t)c->nl);
        mfe_add(c, &rhs, &rhs, &B);
        /* (p+1)/4 */
        memcpy(e, c->p, sizeof(e));
        { u64 carry = 1; for (int w = 0; w < c->nl; w++) { u128 s = (u128)e[w] + carry; e[w] = (u64)s; carry = (u64)(s >> 64); }
          for (int w = 0; w < c->nl; w++) e[w] = (e[w] >> 2) | (w + 1 < c->nl ? e[w + 1] << 62 : carry << 62); }
        mfe_pow(c, &y, &rhs, e);
        mfe chk; mfe_mul(c, &chk, &y, &y);
        if (memcmp(chk.w, rhs.w, 8 * c->nl) != 0) continue;
        u8 yb[48];
        mfe_to_bytes(c, yb, &y);
        if ((yb[nb - 1] & 1) != want_odd) { mfe z; memset(&z, 0, sizeof(z)); mfe_sub(c, &y, &z, &y); mfe_to_bytes(c, yb, &y); }
        memcpy(o, xb, nb); memcpy(o + nb, yb, nb);
        break;
      }
    }
  }
  return 0;
}

/* ======================================================================================== */
/* ECDSA (external ecdsa 0.16.9 hazmat::{verify_prehashed, sign_prehashed}, entered from        */
/* k256/src/ecdsa.rs:182-209, p256/src/ecdsa.rs:72-75, p384/src/ecdsa.rs:69-72)                */
/* ======================================================================================== */
/* scalar fields: the generic Montgomery code above with the group order as modulus */
static mcurve ORD[3];
static pthread_once_t ord_once = PTHREAD_ONCE_INIT;
static void ord_init_one(mcurve* c, int nl, const char* n) {
  memset(c, 0, sizeof(*c));
  c->nl = nl; c->nbytes = nl * 8;
  hex_to_words(c->p, nl, n);
  u64 inv = 1;
  for (int i = 0; i < 6; i++) inv *= 2 - c->p[0] * inv;
  c->minv = (u64)0 - inv;
  mfe x; memset(&x, 0, sizeof(x)); x.w[0] = 1;
  for (int i = 0; i < 64 * nl; i++) mfe_add(c, &x, &x, &x);
  memcpy(c->one, x.w, sizeof(u64) * nl);
  for (int i = 0; i < 64 * nl; i++) mfe_add(c, &x, &x, &x);
  memcpy(c->r2, x.w, sizeof(u64) * nl);
}
static void ord_init(void) {
  ord_init_one(&ORD[0], 4, "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141");
  ord_init_one(&ORD[1], 4, "ffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551");
  ord_init_one(&ORD[2], 6, "ffffffffffffffffffffffffffffffffffffffffffffffffc7634d81f4372ddf581a0db248b0a77aecec196accc52973");
}
static int be_is_zero(const u8* a, int n) { int z = 1; for (int i = 0; i < n; i++) z &= (a[i] == 0); return z; }
/* Reduce::reduce_bytes for a value < 2n */
static void be_reduce_once(u8* a, const u8* n, int nb) { if (be_geq(a, n, nb)) be_sub(a, n, nb); }
/* s > (n - 1) / 2  <=>  2 s > n */
static int be_is_high(const u8* s, const u8* n, int nb) {
  u8 t[49]; int carry = 0;
  for (int i = nb - 1; i >= 0; i--) { int v = 2 * s[i] + carry; t[i + 1] = (u8)v; carry = v >> 8; }
  t[0] = (u8)carry;
  if (t[0]) return 1;
  return memcmp(t + 1, n, nb) > 0;
}
static int pubkey_ok(int curve, const u8* xy, const u8* pmod, int nb) {
  if (be_geq(xy, pmod, nb) || be_geq(xy + nb, pmod, nb)) return 0;
  if (be_is_zero(xy, 2 * nb)) return 0;
  if (curve == 0) {
    fe5 x, y; fe5_from_bytes(&x, xy); fe5_from_bytes(&y, xy + 32);
    fe5 l = fe5_sqr(&y), r = fe5_sqr(&x); r = fe5_mul(&r, &x);
    fe5 seven = {{7, 0, 0, 0, 0}}; r = fe5_add(&r, &seven);
    fe5 nr = fe5_negate(&r, 1); l = fe5_add(&l, &nr);
    return fe5_normalizes_to_zero(&l);
  }
  const mcurve* c = get_mcurve(curve);
  mfe x, y, l, r, t, three; mfe_from_bytes(c, &x, xy); mfe_from_bytes(c, &y, xy + nb);
  MUL(l, y, y); MUL(r, x, x); MUL(r, r, x);
  ADD(three, x, x); ADD(three, three, x); SUB(r, r, three);
  memcpy(t.w, c->b, sizeof(t.w)); ADD(r, r, t); SUB(l, l, r);
  return mfe_is_zero(c, &l);
}
/* R = u1 G + u2 Q -> affine x || y || inf, by the reference's lincomb */
static void ecdsa_lincomb(int curve, const u8* u1, const u8* u2, const u8* q, u8* out) {
  const int nb = curve == 2 ? 48 : 32;
  if (curve == 0) {
    sc4 ks[2]; k256_pt ps[2];
    sc4_from_bytes(&ks[0], u1); sc4_from_bytes(&ks[1], u2);
    u8 g[64]; memcpy(g, K256_GX, 32); memcpy(g + 32, K256_GY, 32);
    ps[0] = k256_pt_from_affine_bytes(g); ps[1] = k256_pt_from_affine_bytes(q);
    k256_pt r = k256_lincomb(ps, ks, 2);
    k256_to_affine_bytes(out, &r);
  } else {
    const mcurve* c = get_mcurve(curve);
    mpt g; memset(&g, 0, sizeof(g)); memcpy(g.x.w, c->gx, 8 * c->nl); memcpy(g.y.w, c->gy, 8 * c->nl); memcpy(g.z.w, c->one, 8 * c->nl);
    mpt qq = mpt_from_affine_bytes(c, q);
    mpt a = mpt_mul(c, &g, u1), b = mpt_mul(c, &qq, u2);
    a = mpt_add(c, &a, &b);
    mpt_to_affine_bytes(c, out, &a);
  }
  (void)nb;
}
/* z: bits2field output, sig: r || s, q: x || y; ok[i] in {0, 1}.  low_s: k256's VerifyPrimitive rejects s > n/2. */
int eco_ecdsa_verify_batch(int curve, const u8* z, const u8* sig, const u8* q, u8* ok, size_t n, int low_s) {
  if (curve < 0 || curve > 2) return -1;
  pthread_once(&ord_once, ord_init);
  const mcurve* f = &ORD[curve];
  u8 pm[48], nm[48]; int nb;
  curve_moduli(curve, pm, nm, &nb);
  for (size_t i = 0; i < n; i++) {
    const u8 *r = sig + 2 * nb * i, *s = r + nb;
    ok[i] = 0;
    if (be_is_zero(r, nb) || be_is_zero(s, nb) || be_geq(r, nm, nb) || be_geq(s, nm, nb)) continue;
    if (low_s && be_is_high(s, nm, nb)) continue;
    if (!pubkey_ok(curve, q + 2 * nb * i, pm, nb)) continue;
    u8 e[48]; memcpy(e, z + nb * i, nb); be_reduce_once(e, nm, nb);
    mfe sm, w, em, rm, u1m, u2m;
    mfe_from_bytes(f, &sm, s); mfe_invert(f, &w, &sm);
    mfe_from_bytes(f, &em, e); mfe_from_bytes(f, &rm, r);
    mfe_mul(f, &u1m, &em, &w); mfe_mul(f, &u2m, &rm, &w);
    u8 u1[48], u2[48], R[97];
    mfe_to_bytes(f, u1, &u1m); mfe_to_bytes(f, u2, &u2m);
    ecdsa_lincomb(curve, u1, u2, q + 2 * nb * i, R);
    if (R[2 * nb]) continue;
    be_reduce_once(R, nm, nb);
    ok[i] = memcmp(R, r, nb) == 0;
  }
  return 0;
}
/* sig_out: r || s (zeros when ok = 0); recid: y_is_odd | x_reduced << 1 */
int eco_ecdsa_sign_batch(int curve, const u8* d, const u8* k, const u8* z, u8* sig_out, u8* recid, u8* ok, size_t n, int low_s) {
  if (curve < 0 || curve > 2) return -1;
  pthread_once(&ord_once, ord_init);
  if (curve == 0) pthread_once(&k256_gen_once, k256_gen_init);
  const mcurve* f = &ORD[curve];
  u8 pm[48], nm[48]; int nb;
  curve_moduli(curve, pm, nm, &nb);
  for (size_t i = 0; i < n; i++) {
    const u8 *di = d + nb * i, *ki = k + nb * i;
    u8* o = sig_out + 2 * nb * i;
    memset(o, 0, 2 * nb); ok[i] = 0; if (recid) recid[i] = 0;
    if (be_is_zero(di, nb) || be_is_zero(ki, nb) || be_geq(di, nm, nb) || be_geq(ki, nm, nb)) continue;
    u8 R[97];
    job_t j = {curve, 0, 1, ki, NULL, R, 0, 1, 0};
    do_mul_range(&j);
    int y_odd = R[2 * nb - 1] & 1, x_red = be_geq(R, nm, nb);
    be_reduce_once(R, nm, nb);
    u8 e[48]; memcpy(e, z + nb * i, nb); be_reduce_once(e, nm, nb);
    mfe km, kinv, em, rm, dm, t, sm;
    mfe_from_bytes(f, &km, ki); mfe_invert(f, &kinv, &km);
    mfe_from_bytes(f, &em, e); mfe_from_bytes(f, &rm, R); mfe_from_bytes(f, &dm, di);
    mfe_mul(f, &t, &rm, &dm); mfe_add(f, &t, &t, &em); mfe_mul(f, &sm, &t, &kinv);
    u8 sb[48]; mfe_to_bytes(f, sb, &sm);
    if (be_is_zero(R, nb) || be_is_zero(sb, nb)) continue;
    if (low_s && be_is_high(sb, nm, nb)) { u8 t2[48]; memcpy(t2, nm, nb); be_sub(t2, sb, nb); memcpy(sb, t2, nb); y_odd ^= 1; }
    memcpy(o, R, nb); memcpy(o + nb, sb, nb);
    if (recid) recid[i] = (u8)(y_odd | (x_red << 1));
    ok[i] = 1;
  }
  return 0;
}

/* P-384 field inversion both ways (canonical big-endian in / out): which = 0 Bernstein-Yang (the reference's), 1 the Fermat chain */
int eco_p384_invert(int which, const u8* a, u8* out, size_t n) {
  const mcurve* c = get_mcurve(2);
  if (!c) return -1;
  for (size_t i = 0; i < n; i++) {
    mfe x, r; memset(&x, 0, sizeof(x));
    mfe_from_bytes(c, &x, a + 48 * i);
    if (which == 0) mfe_invert_by(c, &r, &x); else mfe_invert_fermat(c, &r, &x);
    mfe_to_bytes(c, out + 48 * i, &r);
  }
  return 0;
}

/* point add (op 0, q = X || Y || Z), double (op 1), add_mixed (op 2, q = affine x || y, all zero = AffinePoint::IDENTITY)
 * with exact projective outputs */
int eco_point_op(int curve, int op, const u8* p_xyz, const u8* q, u8* out_xyz, size_t n) {
  const int nb = curve == 2 ? 48 : 32;
  for (size_t i = 0; i < n; i++) {
    int q_inf = 1;
    if (op == 2) for (int j = 0; j < 2 * nb; j++) if (q[2 * nb * i + j]) { q_inf = 0; break; }
    if (curve == 0) {
      k256_pt a, b, r;
      fe5_from_bytes(&a.x, p_xyz + 96 * i); fe5_from_bytes(&a.y, p_xyz + 96 * i + 32); fe5_from_bytes(&a.z, p_xyz + 96 * i + 64);
      if (op == 0) { fe5_from_bytes(&b.x, q + 96 * i); fe5_from_bytes(&b.y, q + 96 * i + 32); fe5_from_bytes(&b.z, q + 96 * i + 64); r = k256_add(&a, &b); }
      else if (op == 2) { fe5_from_bytes(&b.x, q + 64 * i); fe5_from_bytes(&b.y, q + 64 * i + 32); r = k256_add_mixed(&a, &b.x, &b.y, q_inf); }
      else r = k256_double(&a);
      k256_pt_to_bytes(out_xyz + 96 * i, &r);
    } else {
      const mcurve* c = get_mcurve(curve);
      if (!c) return -1;
      mpt a, b, r; memset(&a, 0, sizeof(a)); memset(&b, 0, sizeof(b));
      const u8* s = p_xyz + 3 * nb * i;
      mfe_from_bytes(c, &a.x, s); mfe_from_bytes(c, &a.y, s + nb); mfe_from_bytes(c, &a.z, s + 2 * nb);
      if (op == 0) { s = q + 3 * nb * i; mfe_from_bytes(c, &b.x, s); mfe_from_bytes(c, &b.y, s + nb); mfe_from_bytes(c, &b.z, s + 2 * nb); r = mpt_add(c, &a, &b); }
      else if (op == 2) { s = q + 2 * nb * i; mfe_from_bytes(c, &b.x, s); mfe_from_bytes(c, &b.y, s + nb); r = mpt_add_mixed(c, &a, &b.x, &b.y, q_inf); }
      else r = mpt_double(c, &a);
      u8* o = out_xyz + 3 * nb * i;
      mfe_to_bytes(c, o, &r.x); mfe_to_bytes(c, o + nb, &r.y); mfe_to_bytes(c, o + 2 * nb, &r.z);
    }
  }
  return 0;
}
