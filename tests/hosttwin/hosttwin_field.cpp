// TEST-ONLY host build of the device arithmetic templates (csrc/*.hpp).
// Lets the CPU test-suite check the exact code that the HIP kernels instantiate against the
// oracle in a container without a GPU.  Never linked into libecgpu.so.
#include <string.h>
#include "hosttwin_trace.hpp"
#include "fe_k256.hpp"
using namespace ecgpu;

static void load(FeK256& f, const uint8_t* b) { u32 w[8]; memcpy(w, b, 32); k256::from_be_words(f, w); }
static void store(uint8_t* b, const FeK256& f) { u32 w[8]; k256::to_be_words(w, f); memcpy(b, w, 32); }

extern "C" {
// op: 0 mul 1 sqr 2 add 3 sub 4 neg 5 inv 6 sqrt 7 mul_small(b[0..3] LE) 8 normalize-only 9/10/11 shl<1/2/3>
//     12 mul_add2(x, y, y, x) = 2xy   13 mul_add2(x, y, ~x, ~y) (word-wise complements: raw operands up to 2^256 - 1)
// raw=1: inputs are taken as raw 256-bit integers (possibly >= p) and the output is NOT normalised
int ht_k256_fe_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, int n, int raw) {
  for (int i = 0; i < n; i++) {
    FeK256 x, y, r; load(x, a + 32 * i); if (b) load(y, b + 32 * i);
    int ok = 1;
    switch (op) {
      case 0: k256::mul(r, x, y); break;
      case 1: k256::sqr(r, x); break;
      case 2: k256::add(r, x, y); break;
      case 3: k256::sub(r, x, y); break;
      case 4: k256::neg(r, x); break;
      case 5: k256::inv(r, x); break;
      case 6: ok = k256::sqrt(r, x); break;
      case 7: { u32 k; memcpy(&k, b + 32 * i + 28, 4); k256::mul_small(r, x, bswap32(k)); break; }
      case 8: r = x; break;
      case 9: k256::shl<1>(r, x); break;
      case 10: k256::shl<2>(r, x); break;
      case 11: { r = x; k256::shl<3>(r, r); break; }          // aliased
      case 12: k256::mul_add2(r, x, y, y, x); break;
      case 13: { FeK256 cx, cy; for (int w = 0; w < 8; w++) { cx.v[w] = ~x.v[w]; cy.v[w] = ~y.v[w]; } k256::mul_add2(r, x, y, cx, cy); break; }
      default: return -1;
    }
    if (!raw) k256::normalize(r, r);
    store(out + 32 * i, r);
    if (op == 6 && !ok) memset(out + 32 * i, 0xFF, 32);
  }
  return 0;
}
}

// ---- table-access trace (hosttwin_trace.hpp): every ECGPU_TABLE_TOUCH of this library lands here ---------------------
#include <vector>
static thread_local std::vector<int> g_trace;
static thread_local bool g_trace_on = false;
extern "C" void ht_trace_push(int idx) { if (g_trace_on) g_trace.push_back(idx); }
extern "C" void ht_trace_start(void) { g_trace.clear(); g_trace_on = true; }
// stops recording; copies up to cap entries and returns the number recorded
extern "C" size_t ht_trace_stop(int* out, size_t cap) {
  g_trace_on = false;
  for (size_t i = 0; i < g_trace.size() && i < cap; i++) out[i] = g_trace[i];
  return g_trace.size();
}
