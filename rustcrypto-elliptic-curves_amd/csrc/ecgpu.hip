// C ABI of libecgpu.so (include/ecgpu.h): context, staging, kernel dispatch.  Host side of the
// boundary; all arithmetic happens in the gfx950 kernels of kernels.hpp / mulfast*.hpp.
// There is deliberately no CPU code path in this library.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>
#include <vector>

#include "../../include/ecgpu.h"
#include "kernels.hpp"

using namespace ecgpu;

struct ecgpu_ctx {
  int device = -1;
  int num_cus = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  char err[512] = {0};
  std::mutex mu;
  // grow-only device staging buffers for ECGPU_MEM_HOST calls
  void* stage[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t stage_cap[6] = {0, 0, 0, 0, 0, 0};
  // precomputed generator tables, one per curve, built on first use
  void* gen_table[3] = {nullptr, nullptr, nullptr};
};

static int set_err(ecgpu_ctx* c, int code, const char* fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
  }
  return code;
}
#define HIPCHK(c, call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) return set_err(c, ECGPU_ERR_RUNTIME, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

static int stage_reserve(ecgpu_ctx* c, int slot, size_t bytes) {
  if (bytes <= c->stage_cap[slot]) return 0;
  if (c->stage[slot]) HIPCHK(c, hipFree(c->stage[slot]));
  c->stage[slot] = nullptr;
  c->stage_cap[slot] = 0;
  size_t cap = bytes + bytes / 4 + 256;
  HIPCHK(c, hipMalloc(&c->stage[slot], cap));
  c->stage_cap[slot] = cap;
  return 0;
}

// A device view of one caller buffer: either the pointer itself or a staged copy.
struct Buf {
  ecgpu_ctx* c;
  int slot;
  const void* host_in = nullptr;
  void* host_out = nullptr;
  void* dev = nullptr;
  size_t bytes = 0;
};
static int buf_in(ecgpu_ctx* c, Buf& b, int slot, const void* p, size_t bytes, int mem) {
  b.c = c; b.slot = slot; b.bytes = bytes;
  if (!p || bytes == 0) { b.dev = nullptr; return 0; }
  if (mem == ECGPU_MEM_DEVICE) { b.dev = const_cast<void*>(p); return 0; }
  int rc = stage_reserve(c, slot, bytes);
  if (rc) return rc;
  b.dev = c->stage[slot];
  HIPCHK(c, hipMemcpyAsync(b.dev, p, bytes, hipMemcpyHostToDevice, c->stream));
  return 0;
}
static int buf_out(ecgpu_ctx* c, Buf& b, int slot, void* p, size_t bytes, int mem) {
  b.c = c; b.slot = slot; b.bytes = bytes;
  if (!p || bytes == 0) { b.dev = nullptr; return 0; }
  if (mem == ECGPU_MEM_DEVICE) { b.dev = p; return 0; }
  int rc = stage_reserve(c, slot, bytes);
  if (rc) return rc;
  b.dev = c->stage[slot];
  b.host_out = p;
  return 0;
}
static int buf_finish(ecgpu_ctx* c, Buf& b) {
  if (b.host_out && b.dev) HIPCHK(c, hipMemcpyAsync(b.host_out, b.dev, b.bytes, hipMemcpyDeviceToHost, c->stream));
  return 0;
}
static int finish_host(ecgpu_ctx* c, int mem) {
  if (mem == ECGPU_MEM_HOST) HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

static inline unsigned grid_for(const ecgpu_ctx* c, size_t n, int per_cu) {
  size_t blocks = (n + 255) / 256;
  size_t cap = (size_t)c->num_cus * per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks == 0) blocks = 1;
  return (unsigned)blocks;
}

static bool curve_ok(int curve) { return curve == ECGPU_K256; }   // widened as curves land
#define CURVE_DISPATCH(c, curve, CALL)                                            \
  switch (curve) {                                                                \
    case ECGPU_K256: { using C = CurveK256; CALL; } break;                        \
    default: return set_err(c, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve); \
  }

template <class C>
static int field_op_launch(ecgpu_ctx* c, int op, const u32* a, const u32* b, u32* o, size_t n) {
  const unsigned g = grid_for(c, n, 8);
  switch (op) {
#define FOP(OPC) case OPC: hipLaunchKernelGGL((field_op_kernel<C, OPC>), dim3(g), dim3(256), 0, c->stream, a, b, o, n); break;
    FOP(FE_MUL) FOP(FE_SQR) FOP(FE_ADD) FOP(FE_SUB) FOP(FE_NEG) FOP(FE_INV) FOP(FE_SQRT)
#undef FOP
    default: return set_err(c, ECGPU_ERR_ARG, "unknown field op %d", op);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

template <class C, int OP>
static int point_op_launch(ecgpu_ctx* c, const u32* p, const u32* q, u32* o, size_t n) {
  hipLaunchKernelGGL((point_op_kernel<C, OP>), dim3(grid_for(c, n, 8)), dim3(256), 0, c->stream, p, q, o, n);
  HIPCHK(c, hipGetLastError());
  return 0;
}
static int point_op(ecgpu_ctx* c, int curve, int op, const uint8_t* p, const uint8_t* q, size_t qbytes, uint8_t* out, size_t n, int mem) {
  if (!c || !p || !out || (op != PT_DOUBLE && !q)) return set_err(c, ECGPU_ERR_ARG, "null argument");
  if (!curve_ok(curve)) return set_err(c, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  if (n == 0) return ECGPU_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nb = ecgpu_field_bytes(curve);
  Buf bp, bq, bo;
  int rc;
  if ((rc = buf_in(c, bp, 0, p, n * 3 * nb, mem))) return rc;
  if ((rc = buf_in(c, bq, 1, q, n * qbytes, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out, n * 3 * nb, mem))) return rc;
  const u32* dp = (const u32*)bp.dev; const u32* dq = (const u32*)bq.dev; u32* dout = (u32*)bo.dev;
  switch (op) {
    case PT_ADD: CURVE_DISPATCH(c, curve, rc = (point_op_launch<C, PT_ADD>(c, dp, dq, dout, n))); break;
    case PT_ADD_MIXED: CURVE_DISPATCH(c, curve, rc = (point_op_launch<C, PT_ADD_MIXED>(c, dp, dq, dout, n))); break;
    default: CURVE_DISPATCH(c, curve, rc = (point_op_launch<C, PT_DOUBLE>(c, dp, dq, dout, n))); break;
  }
  if (rc) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}
template <class C>
static int ensure_gen_table(ecgpu_ctx* c) {
  if (c->gen_table[C::ID]) return 0;
  void* t = nullptr;
  HIPCHK(c, hipMalloc(&t, sizeof(typename C::Pt) * C::GEN_TABLE_PTS));
  hipLaunchKernelGGL((gen_table_kernel<C>), dim3(1), dim3(64), 0, c->stream, (typename C::Pt*)t);
  HIPCHK(c, hipGetLastError());
  c->gen_table[C::ID] = t;
  return 0;
}

template <class C>
static int lincomb_launch(ecgpu_ctx* c, const u32* sc, const u32* pts, int pt_fmt, size_t terms, u32* out, int out_fmt,
                          uint8_t* out_inf, size_t n, unsigned flags) {
  const unsigned g = grid_for(c, n, 4);
  if (!pts) {
    if (terms != 1) return set_err(c, ECGPU_ERR_ARG, "generator multiplication takes one term");
    int rc = ensure_gen_table<C>(c);
    if (rc) return rc;
    hipLaunchKernelGGL((mul_gen_ref_kernel<C>), dim3(g), dim3(256), 0, c->stream, sc, (const typename C::Pt*)c->gen_table[C::ID],
                       out, out_fmt, out_inf, n);
  } else if (terms == 1 && C::ID == 0 && !(flags & ECGPU_EXACT_REFERENCE)) {
    // throughput schedule: grid sized so that every lane owns a batch worth of elements when n allows
    // ECGPU_K256_FAST_WAVES (2/3/4) picks the occupancy variant; default chosen from measurements
    static const int waves = [] { const char* e = getenv("ECGPU_K256_FAST_WAVES"); int w = e ? atoi(e) : 4; return (w < 2 || w > 4) ? 4 : w; }();
    if (waves == 2)
      hipLaunchKernelGGL((k256_mul_fast_kernel<16, 2>), dim3(grid_for(c, n, 2)), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
    else if (waves == 4)
      hipLaunchKernelGGL((k256_mul_fast_kernel<16, 4>), dim3(grid_for(c, n, 4)), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
    else
      hipLaunchKernelGGL((k256_mul_fast_kernel<16, 3>), dim3(grid_for(c, n, 3)), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
  } else if (terms == 1) {
    hipLaunchKernelGGL((lincomb_ref_kernel<C, 1>), dim3(g), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
  } else if (terms == 2) {
    hipLaunchKernelGGL((lincomb_ref_kernel<C, 2>), dim3(g), dim3(256), 0, c->stream, sc, pts, pt_fmt, out, out_fmt, out_inf, n);
  } else {
    return set_err(c, ECGPU_ERR_UNSUPPORTED, "lincomb_batch supports 1 or 2 terms per combination (use ecgpu_msm for large sums)");
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}


extern "C" {

const char* ecgpu_version(void) { return "ecgpu 0.1 (gfx950)"; }

size_t ecgpu_field_bytes(int curve) {
  switch (curve) {
    case ECGPU_K256: return 32;
    case ECGPU_P256: return 32;
    case ECGPU_P384: return 48;
    default: return 0;
  }
}

int ecgpu_create(ecgpu_ctx** out, int device_index) {
  if (!out) return ECGPU_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ECGPU_ERR_NO_DEVICE;
  if (device_index < 0 || device_index >= ndev) return ECGPU_ERR_ARG;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_index) != hipSuccess) return ECGPU_ERR_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ECGPU_ERR_NO_DEVICE;   // kernels exist for gfx950 only
  ecgpu_ctx* c = new ecgpu_ctx();
  c->device = device_index;
  c->num_cus = prop.multiProcessorCount;
  if (hipSetDevice(device_index) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    delete c;
    return ECGPU_ERR_RUNTIME;
  }
  c->stream = c->own_stream;
  *out = c;
  return ECGPU_OK;
}

void ecgpu_destroy(ecgpu_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (int i = 0; i < 6; i++) if (c->stage[i]) (void)hipFree(c->stage[i]);
  for (int i = 0; i < 3; i++) if (c->gen_table[i]) (void)hipFree(c->gen_table[i]);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int ecgpu_set_stream(ecgpu_ctx* c, void* s) {
  if (!c) return ECGPU_ERR_ARG;
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return ECGPU_OK;
}
int ecgpu_synchronize(ecgpu_ctx* c) {
  if (!c) return ECGPU_ERR_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ECGPU_OK;
}
const char* ecgpu_last_error(const ecgpu_ctx* c) { return c ? c->err : "null context"; }

int ecgpu_timer_start(ecgpu_ctx* c) {
  if (!c) return ECGPU_ERR_ARG;
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  return ECGPU_OK;
}
int ecgpu_timer_stop(ecgpu_ctx* c, float* ms) {
  if (!c || !ms) return ECGPU_ERR_ARG;
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev1));
  HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
  return ECGPU_OK;
}

// ---------------------------------------------------------------------------------------------
int ecgpu_field_op_batch(ecgpu_ctx* c, int curve, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int mem) {
  if (!c || !a || !out) return set_err(c, ECGPU_ERR_ARG, "null argument");
  if (!curve_ok(curve)) return set_err(c, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  const bool binary = (op == FE_MUL || op == FE_ADD || op == FE_SUB);
  if (binary && !b) return set_err(c, ECGPU_ERR_ARG, "binary field op needs b");
  if (n == 0) return ECGPU_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nb = ecgpu_field_bytes(curve);
  Buf ba, bb, bo;
  int rc;
  if ((rc = buf_in(c, ba, 0, a, n * nb, mem))) return rc;
  if ((rc = buf_in(c, bb, 1, binary ? b : nullptr, n * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out, n * nb, mem))) return rc;
  CURVE_DISPATCH(c, curve, rc = field_op_launch<C>(c, op, (const u32*)ba.dev, (const u32*)bb.dev, (u32*)bo.dev, n));
  if (rc) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}

int ecgpu_point_add_batch(ecgpu_ctx* c, int curve, const uint8_t* p, const uint8_t* q, uint8_t* out, size_t n, int mem) {
  return point_op(c, curve, PT_ADD, p, q, 3 * ecgpu_field_bytes(curve), out, n, mem);
}
int ecgpu_point_add_mixed_batch(ecgpu_ctx* c, int curve, const uint8_t* p, const uint8_t* q, uint8_t* out, size_t n, int mem) {
  return point_op(c, curve, PT_ADD_MIXED, p, q, 2 * ecgpu_field_bytes(curve), out, n, mem);
}
int ecgpu_point_double_batch(ecgpu_ctx* c, int curve, const uint8_t* p, uint8_t* out, size_t n, int mem) {
  return point_op(c, curve, PT_DOUBLE, p, nullptr, 0, out, n, mem);
}

int ecgpu_batch_normalize(ecgpu_ctx* c, int curve, const uint8_t* p, uint8_t* out_xy, uint8_t* out_inf, size_t n, int mem) {
  if (!c || !p || !out_xy) return set_err(c, ECGPU_ERR_ARG, "null argument");
  if (!curve_ok(curve)) return set_err(c, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  if (n == 0) return ECGPU_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nb = ecgpu_field_bytes(curve);
  Buf bp, bo, bi;
  int rc;
  if ((rc = buf_in(c, bp, 0, p, n * 3 * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out_xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bi, 3, out_inf, n, mem))) return rc;
  CURVE_DISPATCH(c, curve, hipLaunchKernelGGL((normalize_kernel<C>), dim3(grid_for(c, n, 8)), dim3(256), 0, c->stream,
                                              (const u32*)bp.dev, (u32*)bo.dev, (uint8_t*)bi.dev, n));
  HIPCHK(c, hipGetLastError());
  if ((rc = buf_finish(c, bo))) return rc;
  if ((rc = buf_finish(c, bi))) return rc;
  return finish_host(c, mem);
}

// ---------------------------------------------------------------------------------------------
int ecgpu_lincomb_batch(ecgpu_ctx* c, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, size_t terms,
                        uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n, int mem, unsigned flags) {
  if (!c || !scalars || !out) return set_err(c, ECGPU_ERR_ARG, "null argument");
  if (!curve_ok(curve)) return set_err(c, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  if ((pt_fmt != FMT_AFFINE && pt_fmt != FMT_PROJECTIVE) || (out_fmt != FMT_AFFINE && out_fmt != FMT_PROJECTIVE))
    return set_err(c, ECGPU_ERR_ARG, "bad point format");
  if (n == 0) return ECGPU_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nb = ecgpu_field_bytes(curve);
  const size_t pin = (pt_fmt == FMT_PROJECTIVE ? 3 : 2) * nb, pout = (out_fmt == FMT_PROJECTIVE ? 3 : 2) * nb;
  Buf bs, bp, bo, bi;
  int rc;
  if ((rc = buf_in(c, bs, 0, scalars, n * terms * nb, mem))) return rc;
  if ((rc = buf_in(c, bp, 1, points, n * terms * pin, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out, n * pout, mem))) return rc;
  if ((rc = buf_out(c, bi, 3, out_fmt == FMT_AFFINE ? out_inf : nullptr, n, mem))) return rc;
  CURVE_DISPATCH(c, curve, rc = lincomb_launch<C>(c, (const u32*)bs.dev, (const u32*)bp.dev, pt_fmt, terms, (u32*)bo.dev, out_fmt,
                                                  (uint8_t*)bi.dev, n, flags));
  if (rc) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  if ((rc = buf_finish(c, bi))) return rc;
  return finish_host(c, mem);
}

int ecgpu_mul_batch(ecgpu_ctx* c, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt,
                    uint8_t* out_inf, size_t n, int mem, unsigned flags) {
  return ecgpu_lincomb_batch(c, curve, scalars, points, pt_fmt, 1, out, out_fmt, out_inf, n, mem, flags);
}

int ecgpu_msm(ecgpu_ctx* c, int curve, const uint8_t*, const uint8_t*, int, size_t, uint8_t*, int, int) {
  (void)curve;
  return set_err(c, ECGPU_ERR_UNSUPPORTED, "ecgpu_msm: not built yet");
}

// ---------------------------------------------------------------------------------------------
int ecgpu_validate_scalars(ecgpu_ctx* c, int curve, const uint8_t* scalars, uint8_t* ok, size_t n, int mem) {
  if (!c || !scalars || !ok) return set_err(c, ECGPU_ERR_ARG, "null argument");
  if (!curve_ok(curve)) return set_err(c, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  if (n == 0) return ECGPU_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nb = ecgpu_field_bytes(curve);
  Buf bs, bo;
  int rc;
  if ((rc = buf_in(c, bs, 0, scalars, n * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, ok, n, mem))) return rc;
  CURVE_DISPATCH(c, curve, hipLaunchKernelGGL((validate_scalars_kernel<C>), dim3(grid_for(c, n, 8)), dim3(256), 0, c->stream,
                                              (const u32*)bs.dev, (uint8_t*)bo.dev, n));
  HIPCHK(c, hipGetLastError());
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}
int ecgpu_validate_points(ecgpu_ctx* c, int curve, const uint8_t* xy, uint8_t* ok, size_t n, int mem) {
  if (!c || !xy || !ok) return set_err(c, ECGPU_ERR_ARG, "null argument");
  if (!curve_ok(curve)) return set_err(c, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  if (n == 0) return ECGPU_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nb = ecgpu_field_bytes(curve);
  Buf bs, bo;
  int rc;
  if ((rc = buf_in(c, bs, 0, xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, ok, n, mem))) return rc;
  CURVE_DISPATCH(c, curve, hipLaunchKernelGGL((validate_points_kernel<C>), dim3(grid_for(c, n, 8)), dim3(256), 0, c->stream,
                                              (const u32*)bs.dev, (uint8_t*)bo.dev, n));
  HIPCHK(c, hipGetLastError());
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}
int ecgpu_decompress_batch(ecgpu_ctx* c, int curve, const uint8_t* x, const uint8_t* y_is_odd, uint8_t* out_xy, uint8_t* ok, size_t n, int mem) {
  if (!c || !x || !y_is_odd || !out_xy || !ok) return set_err(c, ECGPU_ERR_ARG, "null argument");
  if (!curve_ok(curve)) return set_err(c, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  if (n == 0) return ECGPU_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nb = ecgpu_field_bytes(curve);
  Buf bx, by, bo, bk;
  int rc;
  if ((rc = buf_in(c, bx, 0, x, n * nb, mem))) return rc;
  if ((rc = buf_in(c, by, 1, y_is_odd, n, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out_xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bk, 3, ok, n, mem))) return rc;
  CURVE_DISPATCH(c, curve, hipLaunchKernelGGL((decompress_kernel<C>), dim3(grid_for(c, n, 8)), dim3(256), 0, c->stream,
                                              (const u32*)bx.dev, (const uint8_t*)by.dev, (u32*)bo.dev, (uint8_t*)bk.dev, n));
  HIPCHK(c, hipGetLastError());
  if ((rc = buf_finish(c, bo))) return rc;
  if ((rc = buf_finish(c, bk))) return rc;
  return finish_host(c, mem);
}

int ecgpu_synth_scalars(ecgpu_ctx* c, int curve, uint64_t seed, uint64_t first, uint8_t* d_scalars, size_t n) {
  if (!c || !d_scalars) return set_err(c, ECGPU_ERR_ARG, "null argument");
  if (!curve_ok(curve)) return set_err(c, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  if (n == 0) return ECGPU_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  CURVE_DISPATCH(c, curve, hipLaunchKernelGGL((synth_scalars_kernel<C>), dim3(grid_for(c, n, 8)), dim3(256), 0, c->stream,
                                              (u64)seed, (u64)first, (u32*)d_scalars, n));
  HIPCHK(c, hipGetLastError());
  return ECGPU_OK;
}
int ecgpu_synth_points(ecgpu_ctx* c, int curve, uint64_t seed, uint64_t first, uint8_t* d_points, size_t n) {
  if (!c || !d_points) return set_err(c, ECGPU_ERR_ARG, "null argument");
  if (!curve_ok(curve)) return set_err(c, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  if (n == 0) return ECGPU_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  CURVE_DISPATCH(c, curve, hipLaunchKernelGGL((synth_points_kernel<C>), dim3(grid_for(c, n, 8)), dim3(256), 0, c->stream,
                                              (u64)seed, (u64)first, (u32*)d_points, n));
  HIPCHK(c, hipGetLastError());
  return ECGPU_OK;
}

}  // extern "C"
