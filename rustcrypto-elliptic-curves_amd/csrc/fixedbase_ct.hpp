// Constant-time fixed-base multiplication, the per-result body (host + device: the host twin walks it with a trace of the
// table entries it reads).  See fixedbase.hpp (mul_ct_kernel) for the schedule and its place in the library.
#pragma once
#include "jacobian.hpp"

namespace ecgpu {
namespace fb {

constexpr int CT_WB = 5;                                        // signed 5-bit digits
constexpr int CT_ENTRIES = 1 << (CT_WB - 1);                    // T[j][d-1] = d 2^(5j) G, d = 1..16
template <class C> constexpr int ct_nwin() { return (8 * C::NB + CT_WB - 1) / CT_WB; }

// acc = k G for a scalar k < n (reduced by the caller), in the reference's homogeneous projective coordinates.
// Nothing but data depends on k: the digits come from branch-free recoding, EVERY entry of window j is read (the
// address depends on j and the entry number only) and the digit's one is kept by AND / OR masks - arithmetic masks, not
// selects: a select of a loaded value lets the compiler load only under the condition, and it did (the VMEM instruction
// count followed the digits until this was an AND / OR) -, the sign is a masked negation, a zero digit the addend's
// infinity flag, the addition the reference's complete mixed addition.
template <class C>
ECGPU_HD void mul_ct_one(typename C::Pt& acc, const u32* k, const AffEntry<C>* table) {
  constexpr int NW = C::NW, NWIN = ct_nwin<C>();
  static_assert((8 * C::NB) % CT_WB != 0, "the top window must have room for the last carry");
  using Fe = typename C::Fe;
  C::pt_identity(acc);
  u32 carry = 0;
#pragma unroll 1
  for (int j = 0; j < NWIN; j++) {
    const int wi = (CT_WB * j) >> 5, sh = (CT_WB * j) & 31;       // public: the window number
    u32 w0 = 0, w1 = 0;
#pragma unroll
    for (int q = 0; q < NW; q++) { w0 = (wi == q) ? k[q] : w0; w1 = (wi + 1 == q) ? k[q] : w1; }
    const u64 pair = ((u64)w1 << 32) | w0;
    const u32 v = ((u32)(pair >> sh) & ((1u << CT_WB) - 1u)) + carry;            // 0 .. 32
    carry = (j == NWIN - 1) ? 0u : ((v + (1u << (CT_WB - 1))) >> CT_WB);         // v >= 16 -> v - 32 and a carry; the top window keeps v
    const int d = (int)v - (int)(carry << CT_WB);                                // -16 .. 16
    const u32 sgn = (u32)(d >> 31), mag = ((u32)d ^ sgn) - sgn;                  // |d| without a branch
    typename C::Af q;
    C::fe_zero(q.x);
    C::fe_zero(q.y);
    const AffEntry<C>* row = table + (size_t)j * CT_ENTRIES;
#pragma unroll 4
    for (int e = 0; e < CT_ENTRIES; e++) {
      ECGPU_TABLE_TOUCH(j * CT_ENTRIES + e);
      const u32 mk = 0u - (((mag ^ (u32)(e + 1)) - 1u) >> 31);                   // all ones iff mag == e + 1
      const AffEntry<C> t = row[e];
#pragma unroll
      for (int w = 0; w < NW; w++) { q.x.v[w] |= t.x.v[w] & mk; q.y.v[w] |= t.y.v[w] & mk; }
    }
    Fe ny;
    C::fe_neg(ny, q.y);
    C::fe_select(q.y, sgn != 0, ny, q.y);
    q.inf = (mag == 0) ? 1u : 0u;
    typename C::Pt t;
    C::pt_add_mixed(t, acc, q);
    acc = t;
  }
}

}  // namespace fb
}  // namespace ecgpu
