// C ABI of libecgpu.so (include/ecgpu.h): context, staging of host buffers, dispatch to the
// per-curve kernel launchers (ops_*.hip).  Host side of the boundary; all arithmetic happens in
// the gfx950 kernels.  There is deliberately no CPU code path in this library.
#include <string.h>
#include <condition_variable>
#include <thread>
#include <vector>

#include "ecgpu_internal.hpp"
#include "host_pipe.hpp"

static int stage_reserve(ecgpu_ctx* c, int slot, size_t bytes) {
  if (bytes <= c->stage_cap[slot]) return 0;
  if (c->stage[slot]) {
    HIPCHK(c, hipStreamSynchronize(c->stream));          // work queued on the old buffer
    HIPCHK(c, hipFree(c->stage[slot]));
  }
  c->stage[slot] = nullptr;
  c->stage_cap[slot] = 0;
  size_t cap = bytes + bytes / 4 + 256;
  HIPCHK(c, hipMalloc(&c->stage[slot], cap));
  c->stage_cap[slot] = cap;
  return 0;
}

// A device view of one caller buffer: either the pointer itself or a staged copy.
struct Buf {
  void* host_out = nullptr;
  void* dev = nullptr;
  size_t bytes = 0;
};
static int buf_in(ecgpu_ctx* c, Buf& b, int slot, const void* p, size_t bytes, int mem) {
  b.bytes = bytes;
  if (!p || bytes == 0) { b.dev = nullptr; return 0; }
  if (mem == ECGPU_MEM_DEVICE) { b.dev = const_cast<void*>(p); return 0; }
  int rc = stage_reserve(c, slot, bytes);
  if (rc) return rc;
  b.dev = c->stage[slot];
  HIPCHK(c, hipMemcpyAsync(b.dev, p, bytes, hipMemcpyHostToDevice, c->stream));
  return 0;
}
static int buf_out(ecgpu_ctx* c, Buf& b, int slot, void* p, size_t bytes, int mem) {
  b.bytes = bytes;
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

// ---------------------------------------------------------------------------------------------
// Host-buffer calls on large batches stream through the device in chunks (host_pipe.hpp): batches of PIPE_MIN units and more.
// `pass` is the number of units that gives every resident lane of the dominant kernel its full sub-batch (the curve's
// ops->pass_units): chunks grow from pass / 8 to pass and the last one is small again.
// ---------------------------------------------------------------------------------------------
static constexpr size_t PIPE_MIN = (size_t)1 << 21;
// one multi-scalar multiplication from host memory: sums of MSM_PIPE_MIN terms and more are cut into parts of MSM_PIPE_PART terms
// (a part is one slab of either window width: msm_kernels.hpp Geo<CB>::SLAB_TERMS >= 2^23)
static constexpr size_t MSM_PIPE_MIN = (size_t)1 << 22, MSM_PIPE_PART = (size_t)1 << 23;
using PipeArg = hostpipe::Arg;
extern "C" int ecgpu_host_chunk_schedule(size_t n, size_t pass_units, size_t* sizes, size_t cap);
template <class Launch>
static int host_pipeline(ecgpu_ctx* c, const PipeArg* args, int nargs, size_t n, size_t pass, bool secret, Launch launch) {
  std::vector<size_t> sizes((size_t)ecgpu_host_chunk_schedule(n, pass, nullptr, 0));
  (void)ecgpu_host_chunk_schedule(n, pass, sizes.data(), sizes.size());
  return hostpipe::run(c, args, nargs, sizes, secret, [&](int slot, size_t bytes) { return stage_reserve(c, slot, bytes); },
                       [&](void** d, size_t cnt, size_t) { return launch(d, cnt); });
}

static const ecgpu_curve_ops* ops_for(int curve) {
  switch (curve) {
    case ECGPU_K256: return ecgpu_ops_k256();
    case ECGPU_P256: return ecgpu_ops_p256();
    case ECGPU_P384: return ecgpu_ops_p384();
    default: return nullptr;
  }
}
#define ENTER(c, curve)                                                                              \
  const ecgpu_curve_ops* ops = ops_for(curve);                                                       \
  if (!ops) return ecgpu_set_err(c, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);         \
  std::lock_guard<std::mutex> lk(c->mu);                                                             \
  HIPCHK(c, hipSetDevice(c->device));                                                                \
  const size_t nb = ecgpu_field_bytes(curve);                                                        \
  (void)nb

extern "C" {

const char* ecgpu_version(void) { return "ecgpu 0.4 (gfx950)"; }

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
  if (hipSetDevice(device_index) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamDefault) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_switch, hipEventDisableTiming) != hipSuccess) {
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
  for (int i = 0; i < ecgpu_ctx::NSTAGE; i++)
    if (c->stage[i]) {                       // staging slots may have held secret scalars: clear before release
      (void)hipMemset(c->stage[i], 0, c->stage_cap[i]);
      (void)hipFree(c->stage[i]);
    }
  for (int i = 0; i < ecgpu_ctx::PIPE_NSLOT; i++) {
    if (c->ev_kernel[i]) (void)hipEventDestroy(c->ev_kernel[i]);
    if (c->ev_up[i]) (void)hipEventDestroy(c->ev_up[i]);
    if (c->ev_down[i]) (void)hipEventDestroy(c->ev_down[i]);
  }
  for (int d = 0; d < 2; d++)
    for (int w = 0; w < ecgpu_ctx::PIPE_NWORK; w++) {
      if (c->bounce[d][w]) { memset(c->bounce[d][w], 0, ecgpu_ctx::PIPE_BOUNCE); (void)hipHostFree(c->bounce[d][w]); }
      if (c->ev_bounce[d][w]) (void)hipEventDestroy(c->ev_bounce[d][w]);
    }
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
  for (int i = 0; i < 3; i++) if (c->gen_table[i]) (void)hipFree(c->gen_table[i]);
  for (int i = 0; i < 3; i++) if (c->fb_table[i]) (void)hipFree(c->fb_table[i]);
  for (int i = 0; i < 3; i++) if (c->fb16_table[i]) (void)hipFree(c->fb16_table[i]);
  for (int i = 0; i < 3; i++) if (c->fb20_table[i]) (void)hipFree(c->fb20_table[i]);
  for (int i = 0; i < 3; i++) if (c->fb24_table[i]) (void)hipFree(c->fb24_table[i]);
  for (int i = 0; i < 3; i++) if (c->fbct_table[i]) (void)hipFree(c->fbct_table[i]);
  for (int i = 0; i < 3; i++) if (c->fb26_table[i]) (void)hipFree(c->fb26_table[i]);
  if (c->msm_ws) (void)hipFree(c->msm_ws);
  if (c->tab_ws) (void)hipFree(c->tab_ws);
  if (c->ecdsa_ws) (void)hipFree(c->ecdsa_ws);
  if (c->sched_ctr) (void)hipFree(c->sched_ctr);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->ev_switch) (void)hipEventDestroy(c->ev_switch);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

// The per-context scratch (staging slots, table / MSM / ECDSA workspaces, lazily built tables) is shared by
// consecutive calls, so work queued on the old stream must be ordered before anything the new stream does:
// an event recorded on the old stream is waited for by the new one.  If the old stream cannot take the record any
// more (the caller destroyed it), a device-wide synchronisation gives the same ordering; the new stream is installed
// either way, so a context never stays pinned to a dead stream.
static int switch_stream(ecgpu_ctx* c, hipStream_t next) {
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  if (next == c->stream) return ECGPU_OK;
  hipError_t e = hipEventRecord(c->ev_switch, c->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(next, c->ev_switch, 0);
  if (e != hipSuccess) {
    (void)hipGetLastError();                  // not sticky: the next launch check must not report it
    (void)hipDeviceSynchronize();
  }
  c->stream = next;
  return ECGPU_OK;
}
// NULL is the legacy default stream, as for every HIP API (PyTorch's default stream has this handle)
int ecgpu_set_stream(ecgpu_ctx* c, void* s) {
  if (!c) return ECGPU_ERR_ARG;
  return switch_stream(c, (hipStream_t)s);
}
int ecgpu_use_own_stream(ecgpu_ctx* c) {
  if (!c) return ECGPU_ERR_ARG;
  return switch_stream(c, c->own_stream);
}
int ecgpu_synchronize(ecgpu_ctx* c) {
  if (!c) return ECGPU_ERR_ARG;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ECGPU_OK;
}
const char* ecgpu_last_error(const ecgpu_ctx* c) { return c ? c->err : "null context"; }
int ecgpu_last_error_copy(ecgpu_ctx* c, char* buf, size_t cap) {
  if (!c || !buf || cap == 0) return ECGPU_ERR_ARG;
  std::lock_guard<std::mutex> lk(c->err_mu);
  strncpy(buf, c->err, cap - 1);
  buf[cap - 1] = 0;
  return ECGPU_OK;
}

int ecgpu_set_option(ecgpu_ctx* c, int option, int64_t v) {
  if (!c) return ECGPU_ERR_ARG;
  bool ok = false;
  switch (option) {
    case ECGPU_OPT_FB_WINDOW: ok = (v == 0 || v == 8 || v == 16 || v == 20 || v == 24 || v == 26); break;
    case ECGPU_OPT_FB_MAX_WINDOW: ok = (v == 8 || v == 16 || v == 20 || v == 24 || v == 26); break;
    case ECGPU_OPT_MSM_WINDOW_BITS: ok = (v == 0 || v == 16 || v == 19); break;
    case ECGPU_OPT_MSM_SLAB_TERMS: ok = (v == 0 || (v >= 1024 && v <= ((int64_t)1 << 24))); break;
    case ECGPU_OPT_MSM_SMALL_PATH: ok = (v == 0 || v == 1); break;
    case ECGPU_OPT_MSM_ROUNDS: ok = (v >= 0 && v <= 64); break;
    case ECGPU_OPT_K256_WAVES: ok = (v == 3 || v == 4); break;
    case ECGPU_OPT_FB_MEMORY_BUDGET: ok = (v >= 0); break;
    case ECGPU_OPT_LINCOMB_TERM_BY_TERM: ok = (v == 0 || v == 1); break;
    default: return ecgpu_set_err(c, ECGPU_ERR_ARG, "ecgpu_set_option: unknown option %d", option);
  }
  if (!ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "ecgpu_set_option: value %lld is not allowed for option %d", (long long)v, option);
  std::lock_guard<std::mutex> lk(c->mu);
  c->opt[option] = v;
  return ECGPU_OK;
}
int ecgpu_get_option(ecgpu_ctx* c, int option, int64_t* v) {
  if (!c || !v) return ECGPU_ERR_ARG;
  if (option < 0 || option >= ECGPU_OPT_COUNT_) return ecgpu_set_err(c, ECGPU_ERR_ARG, "ecgpu_get_option: unknown option %d", option);
  std::lock_guard<std::mutex> lk(c->mu);
  *v = c->opt[option];
  return ECGPU_OK;
}
int ecgpu_fb_table_bytes(ecgpu_ctx* c, int curve, size_t* bytes, int* widest) {
  if (!c) return ECGPU_ERR_ARG;
  if (curve < 0 || curve > 2) return ecgpu_set_err(c, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  std::lock_guard<std::mutex> lk(c->mu);
  if (bytes) *bytes = c->fb_bytes[curve];
  if (widest) *widest = c->fb_widest[curve];
  return ECGPU_OK;
}

int ecgpu_host_chunk_schedule(size_t n, size_t pass_units, size_t* sizes, size_t cap) {
  size_t pass = pass_units;
  if (pass > ((size_t)1 << 23)) pass = (size_t)1 << 23;
  if (pass < ((size_t)1 << 20)) pass = (size_t)1 << 20;
  const std::vector<size_t> v = hostpipe::schedule(n, pass / 8, pass, pass / 8, pass / 16);
  for (size_t i = 0; i < v.size() && i < cap && sizes; i++) sizes[i] = v[i];
  return (int)v.size();
}

int ecgpu_debug_workspace(ecgpu_ctx* c, int which, void* host_copy, size_t cap, size_t* bytes) {
  if (!c || !bytes) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  void* p = nullptr;
  size_t sz = 0;
  if (which == 0) { p = c->tab_ws; sz = c->tab_ws_cap; }
  else if (which == 1) { p = c->ecdsa_ws; sz = c->ecdsa_ws_cap; }
  else if (which == 2) { p = c->msm_ws; sz = c->msm_ws_cap; }
  else if (which >= 16 && which < 16 + ecgpu_ctx::NSTAGE) { p = c->stage[which - 16]; sz = c->stage_cap[which - 16]; }
  else return ecgpu_set_err(c, ECGPU_ERR_ARG, "ecgpu_debug_workspace: unknown workspace %d", which);
  *bytes = sz;
  if (host_copy && p && sz && cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(host_copy, p, cap < sz ? cap : sz, hipMemcpyDeviceToHost));
  }
  return ECGPU_OK;
}

int ecgpu_host_alloc(ecgpu_ctx* c, size_t bytes, void** out) {
  if (!c || !out) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  *out = nullptr;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return ECGPU_OK;
}
int ecgpu_host_free(ecgpu_ctx* c, void* p) {
  if (!c) return ECGPU_ERR_ARG;
  if (!p) return ECGPU_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipHostFree(p));
  return ECGPU_OK;
}

int ecgpu_timer_start(ecgpu_ctx* c) {
  if (!c) return ECGPU_ERR_ARG;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  return ECGPU_OK;
}
int ecgpu_timer_stop(ecgpu_ctx* c, float* ms) {
  if (!c || !ms) return ECGPU_ERR_ARG;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev1));
  HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
  return ECGPU_OK;
}

// ---------------------------------------------------------------------------------------------
int ecgpu_field_op_batch(ecgpu_ctx* c, int curve, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int mem) {
  if (!c || !a || !out) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  const bool binary = (op == ECGPU_FE_MUL || op == ECGPU_FE_ADD || op == ECGPU_FE_SUB);
  if (binary && !b) return ecgpu_set_err(c, ECGPU_ERR_ARG, "binary field op needs b");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  Buf ba, bb, bo;
  int rc;
  if ((rc = buf_in(c, ba, 0, a, n * nb, mem))) return rc;
  if ((rc = buf_in(c, bb, 1, binary ? b : nullptr, n * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out, n * nb, mem))) return rc;
  if ((rc = ops->field_op(c, op, (const uint32_t*)ba.dev, (const uint32_t*)bb.dev, (uint32_t*)bo.dev, n))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}

static int point_op(ecgpu_ctx* c, int curve, int op, const uint8_t* p, const uint8_t* q, int q_coords, uint8_t* out, size_t n, int mem) {
  if (!c || !p || !out || (q_coords && !q)) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  Buf bp, bq, bo;
  int rc;
  if ((rc = buf_in(c, bp, 0, p, n * 3 * nb, mem))) return rc;
  if ((rc = buf_in(c, bq, 1, q, n * q_coords * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out, n * 3 * nb, mem))) return rc;
  if ((rc = ops->point_op(c, op, (const uint32_t*)bp.dev, (const uint32_t*)bq.dev, (uint32_t*)bo.dev, n))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}
int ecgpu_point_add_batch(ecgpu_ctx* c, int curve, const uint8_t* p, const uint8_t* q, uint8_t* out, size_t n, int mem) {
  return point_op(c, curve, 0, p, q, 3, out, n, mem);
}
int ecgpu_point_add_mixed_batch(ecgpu_ctx* c, int curve, const uint8_t* p, const uint8_t* q, uint8_t* out, size_t n, int mem) {
  return point_op(c, curve, 1, p, q, 2, out, n, mem);
}
int ecgpu_point_double_batch(ecgpu_ctx* c, int curve, const uint8_t* p, uint8_t* out, size_t n, int mem) {
  return point_op(c, curve, 2, p, nullptr, 0, out, n, mem);
}

int ecgpu_point_eq_batch(ecgpu_ctx* c, int curve, const uint8_t* p, const uint8_t* q, uint8_t* eq, size_t n, int mem) {
  if (!c || !p || !q || !eq) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  Buf bp, bq, bo;
  int rc;
  if ((rc = buf_in(c, bp, 0, p, n * 3 * nb, mem))) return rc;
  if ((rc = buf_in(c, bq, 1, q, n * 3 * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, eq, n, mem))) return rc;
  if ((rc = ops->point_eq(c, (const uint32_t*)bp.dev, (const uint32_t*)bq.dev, (uint8_t*)bo.dev, n))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}

int ecgpu_batch_normalize(ecgpu_ctx* c, int curve, const uint8_t* p, uint8_t* out_xy, uint8_t* out_inf, size_t n, int mem) {
  if (!c || !p || !out_xy) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  Buf bp, bo, bi;
  int rc;
  if ((rc = buf_in(c, bp, 0, p, n * 3 * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out_xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bi, 3, out_inf, n, mem))) return rc;
  if ((rc = ops->normalize(c, (const uint32_t*)bp.dev, (uint32_t*)bo.dev, (uint8_t*)bi.dev, n))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  if ((rc = buf_finish(c, bi))) return rc;
  return finish_host(c, mem);
}

// Staged copies of secret scalars do not outlive the call (the reference zeroizes its secrets): the guard clears the
// bytes this call staged in its slots on EVERY exit path, the error returns included, and waits for the clearing.
struct SecretWipe {
  ecgpu_ctx* c;
  static constexpr int CAP = 3 * ecgpu_ctx::PIPE_NSLOT + 3;
  int slot[CAP];
  size_t bytes[CAP];
  int cnt = 0;
  explicit SecretWipe(ecgpu_ctx* ctx) : c(ctx) {}
  void arm(int s, size_t b) { if (cnt < CAP) { slot[cnt] = s; bytes[cnt] = b; cnt++; } }
  // argument `a` of every pipeline slot, whole capacity (the chunk sizes vary)
  void arm_pipeline(int a) { for (int sl = 0; sl < ecgpu_ctx::PIPE_NSLOT; sl++) arm(hostpipe::stage_index(sl, a), (size_t)-1); }
  ~SecretWipe() {
    if (!cnt) return;
    for (int i = 0; i < cnt; i++) {
      const size_t b = bytes[i] < c->stage_cap[slot[i]] ? bytes[i] : c->stage_cap[slot[i]];
      if (c->stage[slot[i]] && b) (void)hipMemsetAsync(c->stage[slot[i]], 0, b, c->stream);
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
  }
};

static int lincomb_impl(ecgpu_ctx* c, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, size_t terms, uint8_t* out, int out_fmt,
                        uint8_t* out_inf, uint8_t* scalar_ok, size_t n, int mem, unsigned flags) {
  if (!c || !scalars || !out) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if ((pt_fmt != ECGPU_PT_AFFINE && pt_fmt != ECGPU_PT_PROJECTIVE) || (out_fmt != ECGPU_PT_AFFINE && out_fmt != ECGPU_PT_PROJECTIVE))
    return ecgpu_set_err(c, ECGPU_ERR_ARG, "bad point format");
  if (terms == 0) return ecgpu_set_err(c, ECGPU_ERR_ARG, "terms must be >= 1");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  const size_t pin = (pt_fmt == ECGPU_PT_PROJECTIVE ? 3 : 2) * nb, pout = (out_fmt == ECGPU_PT_PROJECTIVE ? 3 : 2) * nb;
  // the reference schedule is the one meant for secret scalars: their staged copies do not outlive the call
  const bool secret = (flags & (ECGPU_EXACT_REFERENCE | ECGPU_SECRET_SCALARS)) != 0;
  SecretWipe wipe(c);
  if (mem == ECGPU_MEM_HOST && n >= PIPE_MIN) {
    if (secret) { wipe.arm_pipeline(0); if (flags & ECGPU_SECRET_SCALARS) wipe.arm_pipeline(2); }    // the scalars; for a secret scalar the product is a secret too
    const PipeArg args[5] = {{scalars, nullptr, terms * nb}, {points, nullptr, points ? terms * pin : 0}, {nullptr, out, pout},
                             {nullptr, out_fmt == ECGPU_PT_AFFINE ? out_inf : nullptr, 1}, {nullptr, scalar_ok, 1}};
    int rc = host_pipeline(c, args, 5, n, ops->pass_units(c, points != nullptr, terms, flags), secret, [&](void** d, size_t cnt) {
      if (d[4]) {
        int r2 = ops->validate_scalars(c, (const uint32_t*)d[0], (uint8_t*)d[4], cnt, terms);
        if (r2) return r2;
      }
      return ops->lincomb(c, (const uint32_t*)d[0], (const uint32_t*)d[1], pt_fmt, terms, (uint32_t*)d[2], out_fmt, (uint8_t*)d[3], cnt, flags);
    });
    return rc;
  }
  Buf bs, bp, bo, bi, bk;
  int rc;
  if (secret && mem == ECGPU_MEM_HOST) { wipe.arm(0, n * terms * nb); if (flags & ECGPU_SECRET_SCALARS) wipe.arm(2, n * pout); }
  if ((rc = buf_in(c, bs, 0, scalars, n * terms * nb, mem))) return rc;
  if ((rc = buf_in(c, bp, 1, points, n * terms * pin, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out, n * pout, mem))) return rc;
  if ((rc = buf_out(c, bi, 3, out_fmt == ECGPU_PT_AFFINE ? out_inf : nullptr, n, mem))) return rc;
  if ((rc = buf_out(c, bk, 5, scalar_ok, n, mem))) return rc;
  if (bk.dev && (rc = ops->validate_scalars(c, (const uint32_t*)bs.dev, (uint8_t*)bk.dev, n, terms))) return rc;
  if ((rc = ops->lincomb(c, (const uint32_t*)bs.dev, (const uint32_t*)bp.dev, pt_fmt, terms, (uint32_t*)bo.dev, out_fmt,
                         (uint8_t*)bi.dev, n, flags)))
    return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  if ((rc = buf_finish(c, bi))) return rc;
  if ((rc = buf_finish(c, bk))) return rc;
  return finish_host(c, mem);
}

int ecgpu_lincomb_batch(ecgpu_ctx* c, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, size_t terms,
                        uint8_t* out, int out_fmt, uint8_t* out_inf, size_t n, int mem, unsigned flags) {
  return lincomb_impl(c, curve, scalars, points, pt_fmt, terms, out, out_fmt, out_inf, nullptr, n, mem, flags);
}
int ecgpu_lincomb_batch_checked(ecgpu_ctx* c, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, size_t terms,
                                uint8_t* out, int out_fmt, uint8_t* out_inf, uint8_t* scalar_ok, size_t n, int mem, unsigned flags) {
  if (!scalar_ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  return lincomb_impl(c, curve, scalars, points, pt_fmt, terms, out, out_fmt, out_inf, scalar_ok, n, mem, flags);
}

int ecgpu_mul_batch(ecgpu_ctx* c, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt,
                    uint8_t* out_inf, size_t n, int mem, unsigned flags) {
  return lincomb_impl(c, curve, scalars, points, pt_fmt, 1, out, out_fmt, out_inf, nullptr, n, mem, flags);
}
int ecgpu_mul_batch_checked(ecgpu_ctx* c, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt,
                            uint8_t* out_inf, uint8_t* scalar_ok, size_t n, int mem, unsigned flags) {
  if (!scalar_ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  return lincomb_impl(c, curve, scalars, points, pt_fmt, 1, out, out_fmt, out_inf, scalar_ok, n, mem, flags);
}

// `mem` is where the inputs live, `out_mem` where the one result point goes (the device group sums host-resident slices into
// device-resident partial points: group.hip)
static int msm_impl(ecgpu_ctx* c, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, size_t n, uint8_t* out, int out_fmt,
                    int mem, int out_mem) {
  if (!c || !out || (n && (!scalars || !points))) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if ((pt_fmt != ECGPU_PT_AFFINE && pt_fmt != ECGPU_PT_PROJECTIVE) || (out_fmt != ECGPU_PT_AFFINE && out_fmt != ECGPU_PT_PROJECTIVE))
    return ecgpu_set_err(c, ECGPU_ERR_ARG, "bad point format");
  ENTER(c, curve);
  const size_t pin = (pt_fmt == ECGPU_PT_PROJECTIVE ? 3 : 2) * nb, pout = (out_fmt == ECGPU_PT_PROJECTIVE ? 3 : 2) * nb;
  Buf bs, bp, bo;
  int rc;
  if (n == 0) {                              // the empty sum is the identity: affine zeros, projective (0 : 1 : 0)
    if (out_mem == ECGPU_MEM_HOST) {
      memset(out, 0, pout);
      if (out_fmt == ECGPU_PT_PROJECTIVE) out[2 * nb - 1] = 1;
      return ECGPU_OK;
    }
    HIPCHK(c, hipMemsetAsync(out, 0, pout, c->stream));
    if (out_fmt == ECGPU_PT_PROJECTIVE) HIPCHK(c, hipMemsetAsync(out + 2 * nb - 1, 1, 1, c->stream));
    return ECGPU_OK;
  }
  if (mem == ECGPU_MEM_HOST && n >= MSM_PIPE_MIN) {
    // A large sum from host memory is cut into parts that stream through the pipeline's slots (round 3 staged the WHOLE input - 6.4 GB
    // at 2^26 terms - before the first kernel started): part i is summed while part i + 1 uploads; every part leaves ONE projective
    // point on the device and the parts are added up at the end (a sum with unit scalars: same output formatting as one call).
    const std::vector<size_t> sizes = hostpipe::schedule(n, MSM_PIPE_PART / 2, MSM_PIPE_PART, 0, MSM_PIPE_PART / 8);
    const size_t parts = sizes.size();
    if ((rc = stage_reserve(c, 3, parts * 3 * nb))) return rc;
    if ((rc = stage_reserve(c, 4, parts * nb))) return rc;
    uint8_t* partial = (uint8_t*)c->stage[3];
    const PipeArg args[2] = {{scalars, nullptr, nb}, {points, nullptr, pin}};
    rc = hostpipe::run(c, args, 2, sizes, false, [&](int slot, size_t bytes) { return stage_reserve(c, slot, bytes); },
                       [&](void** d, size_t cnt, size_t ci) {
                         return ops->msm(c, (const uint32_t*)d[0], (const uint32_t*)d[1], pt_fmt, cnt, (uint32_t*)(partial + ci * 3 * nb), ECGPU_PT_PROJECTIVE);
                       });
    if (rc) return rc;
    std::vector<uint8_t> ones(parts * nb, 0);
    for (size_t i = 0; i < parts; i++) ones[i * nb + nb - 1] = 1;
    HIPCHK(c, hipMemcpyAsync(c->stage[4], ones.data(), parts * nb, hipMemcpyHostToDevice, c->stream));
    if ((rc = buf_out(c, bo, 2, out, pout, out_mem))) return rc;
    if ((rc = ops->msm(c, (const uint32_t*)c->stage[4], (const uint32_t*)partial, ECGPU_PT_PROJECTIVE, parts, (uint32_t*)bo.dev, out_fmt))) return rc;
    if ((rc = buf_finish(c, bo))) return rc;
    return finish_host(c, ECGPU_MEM_HOST);   // `ones` lives until the stream has been synchronised here
  }
  if ((rc = buf_in(c, bs, 0, scalars, n * nb, mem))) return rc;
  if ((rc = buf_in(c, bp, 1, points, n * pin, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out, pout, out_mem))) return rc;
  if ((rc = ops->msm(c, (const uint32_t*)bs.dev, (const uint32_t*)bp.dev, pt_fmt, n, (uint32_t*)bo.dev, out_fmt))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, (mem == ECGPU_MEM_HOST || out_mem == ECGPU_MEM_HOST) ? ECGPU_MEM_HOST : ECGPU_MEM_DEVICE);    // staged inputs must have left the host buffers
}
int ecgpu_msm(ecgpu_ctx* c, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, size_t n, uint8_t* out, int out_fmt,
              int mem) {
  return msm_impl(c, curve, scalars, points, pt_fmt, n, out, out_fmt, mem, mem);
}


// ---------------------------------------------------------------------------------------------
int ecgpu_validate_scalars(ecgpu_ctx* c, int curve, const uint8_t* scalars, uint8_t* ok, size_t n, int mem) {
  if (!c || !scalars || !ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  Buf bs, bo;
  int rc;
  if ((rc = buf_in(c, bs, 0, scalars, n * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, ok, n, mem))) return rc;
  if ((rc = ops->validate_scalars(c, (const uint32_t*)bs.dev, (uint8_t*)bo.dev, n, 1))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}
int ecgpu_validate_points(ecgpu_ctx* c, int curve, const uint8_t* xy, uint8_t* ok, size_t n, int mem) {
  if (!c || !xy || !ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  Buf bs, bo;
  int rc;
  if ((rc = buf_in(c, bs, 0, xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, ok, n, mem))) return rc;
  if ((rc = ops->validate_points(c, (const uint32_t*)bs.dev, (uint8_t*)bo.dev, n))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}
int ecgpu_decompress_batch(ecgpu_ctx* c, int curve, const uint8_t* x, const uint8_t* y_is_odd, uint8_t* out_xy, uint8_t* ok, size_t n, int mem) {
  if (!c || !x || !y_is_odd || !out_xy || !ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  Buf bx, by, bo, bk;
  int rc;
  if ((rc = buf_in(c, bx, 0, x, n * nb, mem))) return rc;
  if ((rc = buf_in(c, by, 1, y_is_odd, n, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out_xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bk, 3, ok, n, mem))) return rc;
  if ((rc = ops->decompress(c, (const uint32_t*)bx.dev, (const uint8_t*)by.dev, (uint32_t*)bo.dev, (uint8_t*)bk.dev, n))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  if ((rc = buf_finish(c, bk))) return rc;
  return finish_host(c, mem);
}

int ecgpu_to_bytes_batch(ecgpu_ctx* c, int curve, const uint8_t* points, int pt_fmt, uint8_t* out, size_t n, int mem) {
  if (!c || !points || !out) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (pt_fmt != ECGPU_PT_AFFINE && pt_fmt != ECGPU_PT_PROJECTIVE) return ecgpu_set_err(c, ECGPU_ERR_ARG, "bad point format");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  Buf bp, bo;
  int rc;
  if ((rc = buf_in(c, bp, 0, points, n * (pt_fmt == ECGPU_PT_PROJECTIVE ? 3 : 2) * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out, n * (nb + 1), mem))) return rc;
  if ((rc = ops->to_bytes(c, (const uint32_t*)bp.dev, pt_fmt, (uint8_t*)bo.dev, n))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}
int ecgpu_from_bytes_batch(ecgpu_ctx* c, int curve, const uint8_t* in, uint8_t* out_xy, uint8_t* ok, size_t n, int mem) {
  if (!c || !in || !out_xy || !ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  Buf bi, bo, bk;
  int rc;
  if ((rc = buf_in(c, bi, 0, in, n * (nb + 1), mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out_xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bk, 3, ok, n, mem))) return rc;
  if ((rc = ops->from_bytes(c, (const uint8_t*)bi.dev, (uint32_t*)bo.dev, (uint8_t*)bk.dev, n))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  if ((rc = buf_finish(c, bk))) return rc;
  return finish_host(c, mem);
}

int ecgpu_sec1_encode_batch(ecgpu_ctx* c, int curve, const uint8_t* points, int pt_fmt, int compress, uint8_t* out, size_t n, int mem) {
  if (!c || !points || !out) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (pt_fmt != ECGPU_PT_AFFINE && pt_fmt != ECGPU_PT_PROJECTIVE) return ecgpu_set_err(c, ECGPU_ERR_ARG, "bad point format");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  const size_t rec = 1 + (compress ? 1 : 2) * nb;
  Buf bp, bo;
  int rc;
  if ((rc = buf_in(c, bp, 0, points, n * (pt_fmt == ECGPU_PT_PROJECTIVE ? 3 : 2) * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out, n * rec, mem))) return rc;
  if ((rc = ops->sec1_encode(c, (const uint32_t*)bp.dev, pt_fmt, compress ? 1 : 0, (uint8_t*)bo.dev, n))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}
int ecgpu_sec1_decode_batch(ecgpu_ctx* c, int curve, const uint8_t* in, size_t record_bytes, uint8_t* out_xy, uint8_t* ok, size_t n, int mem) {
  if (!c || !in || !out_xy || !ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  if (record_bytes != 1 + nb && record_bytes != 1 + 2 * nb)
    return ecgpu_set_err(c, ECGPU_ERR_ARG, "ecgpu_sec1_decode_batch: record_bytes must be %zu (compressed) or %zu (uncompressed)", 1 + nb, 1 + 2 * nb);
  Buf bi, bo, bk;
  int rc;
  if ((rc = buf_in(c, bi, 0, in, n * record_bytes, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out_xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bk, 3, ok, n, mem))) return rc;
  if ((rc = ops->sec1_decode(c, (const uint8_t*)bi.dev, record_bytes, (uint32_t*)bo.dev, (uint8_t*)bk.dev, n))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  if ((rc = buf_finish(c, bk))) return rc;
  return finish_host(c, mem);
}

// ---------------------------------------------------------------------------------------------
int ecgpu_ecdsa_verify_batch(ecgpu_ctx* c, int curve, const uint8_t* prehash, const uint8_t* sig_rs, const uint8_t* pubkeys_xy, uint8_t* ok,
                             size_t n, int mem, unsigned flags) {
  if (!c || !prehash || !sig_rs || !pubkeys_xy || !ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  if (mem == ECGPU_MEM_HOST && n >= PIPE_MIN) {
    const PipeArg args[4] = {{prehash, nullptr, nb}, {sig_rs, nullptr, 2 * nb}, {pubkeys_xy, nullptr, 2 * nb}, {nullptr, ok, 1}};
    return host_pipeline(c, args, 4, n, ops->pass_units(c, 1, 1, 0), false, [&](void** d, size_t cnt) {
      return ops->ecdsa_verify(c, (const uint32_t*)d[0], (const uint32_t*)d[1], (const uint32_t*)d[2], (uint8_t*)d[3], cnt, flags);
    });
  }
  Buf bz, bs, bq, bo;
  int rc;
  if ((rc = buf_in(c, bz, 0, prehash, n * nb, mem))) return rc;
  if ((rc = buf_in(c, bs, 1, sig_rs, n * 2 * nb, mem))) return rc;
  if ((rc = buf_in(c, bq, 4, pubkeys_xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, ok, n, mem))) return rc;
  if ((rc = ops->ecdsa_verify(c, (const uint32_t*)bz.dev, (const uint32_t*)bs.dev, (const uint32_t*)bq.dev, (uint8_t*)bo.dev, n, flags))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}
int ecgpu_map_to_curve_batch(ecgpu_ctx* c, int curve, const uint8_t* u, int count, uint8_t* out_xy, uint8_t* out_inf, size_t n, int mem) {
  if (!c || !u || !out_xy) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (count != 1 && count != 2) return ecgpu_set_err(c, ECGPU_ERR_ARG, "count must be 1 (map_to_curve) or 2 (hash_to_curve: Q0 + Q1)");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  Buf bu, bo, bi;
  int rc;
  if ((rc = buf_in(c, bu, 0, u, n * count * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out_xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bi, 3, out_inf, n, mem))) return rc;
  if ((rc = ops->h2c_map(c, (const uint32_t*)bu.dev, count, (uint32_t*)bo.dev, (uint8_t*)bi.dev, n))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  if ((rc = buf_finish(c, bi))) return rc;
  return finish_host(c, mem);
}
int ecgpu_ecdsa_recover_batch(ecgpu_ctx* c, int curve, const uint8_t* prehash, const uint8_t* sig_rs, const uint8_t* recovery_id,
                              uint8_t* pubkeys_xy, uint8_t* ok, size_t n, int mem, unsigned flags) {
  if (!c || !prehash || !sig_rs || !recovery_id || !pubkeys_xy || !ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  if (mem == ECGPU_MEM_HOST && n >= PIPE_MIN) {
    const PipeArg args[5] = {{prehash, nullptr, nb}, {sig_rs, nullptr, 2 * nb}, {recovery_id, nullptr, 1}, {nullptr, pubkeys_xy, 2 * nb}, {nullptr, ok, 1}};
    return host_pipeline(c, args, 5, n, ops->pass_units(c, 1, 1, 0), false, [&](void** d, size_t cnt) {
      return ops->ecdsa_recover(c, (const uint32_t*)d[0], (const uint32_t*)d[1], (const uint8_t*)d[2], (uint32_t*)d[3], (uint8_t*)d[4], cnt, flags);
    });
  }
  Buf bz, bs, br, bq, bo;
  int rc;
  if ((rc = buf_in(c, bz, 0, prehash, n * nb, mem))) return rc;
  if ((rc = buf_in(c, bs, 1, sig_rs, n * 2 * nb, mem))) return rc;
  if ((rc = buf_in(c, br, 4, recovery_id, n, mem))) return rc;
  if ((rc = buf_out(c, bq, 2, pubkeys_xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 3, ok, n, mem))) return rc;
  if ((rc = ops->ecdsa_recover(c, (const uint32_t*)bz.dev, (const uint32_t*)bs.dev, (const uint8_t*)br.dev, (uint32_t*)bq.dev, (uint8_t*)bo.dev, n,
                               flags)))
    return rc;
  if ((rc = buf_finish(c, bq))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}
int ecgpu_schnorr_verify_batch(ecgpu_ctx* c, int curve, const uint8_t* pubkeys_x, const uint8_t* sig_rs, const uint8_t* challenges, uint8_t* ok,
                               size_t n, int mem) {
  if (!c || !pubkeys_x || !sig_rs || !challenges || !ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  if (mem == ECGPU_MEM_HOST && n >= PIPE_MIN) {
    const PipeArg args[4] = {{pubkeys_x, nullptr, nb}, {sig_rs, nullptr, 2 * nb}, {challenges, nullptr, nb}, {nullptr, ok, 1}};
    return host_pipeline(c, args, 4, n, ops->pass_units(c, 1, 1, 0), false, [&](void** d, size_t cnt) {
      return ops->schnorr_verify(c, (const uint32_t*)d[0], (const uint32_t*)d[1], (const uint32_t*)d[2], (uint8_t*)d[3], cnt);
    });
  }
  Buf bx, bs, be, bo;
  int rc;
  if ((rc = buf_in(c, bx, 0, pubkeys_x, n * nb, mem))) return rc;
  if ((rc = buf_in(c, bs, 1, sig_rs, n * 2 * nb, mem))) return rc;
  if ((rc = buf_in(c, be, 4, challenges, n * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, ok, n, mem))) return rc;
  if ((rc = ops->schnorr_verify(c, (const uint32_t*)bx.dev, (const uint32_t*)bs.dev, (const uint32_t*)be.dev, (uint8_t*)bo.dev, n))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}
int ecgpu_ecdsa_sign_batch(ecgpu_ctx* c, int curve, const uint8_t* secret_d, const uint8_t* nonce_k, const uint8_t* prehash, uint8_t* sig_rs,
                           uint8_t* recovery_id, uint8_t* ok, size_t n, int mem, unsigned flags) {
  if (!c || !secret_d || !nonce_k || !prehash || !sig_rs || !ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  SecretWipe wipe(c);                        // staged secret keys and nonces are cleared on every exit path
  if (mem == ECGPU_MEM_HOST && n >= PIPE_MIN) {
    wipe.arm_pipeline(0); wipe.arm_pipeline(1);          // secret keys, nonces
    const PipeArg args[6] = {{secret_d, nullptr, nb}, {nonce_k, nullptr, nb}, {prehash, nullptr, nb}, {nullptr, sig_rs, 2 * nb},
                             {nullptr, recovery_id, 1}, {nullptr, ok, 1}};
    const unsigned fb_flags = (flags & ECGPU_PUBLIC_SCALARS) ? 0u : (flags & ECGPU_EXACT_REFERENCE) ? (unsigned)ECGPU_EXACT_REFERENCE : (unsigned)ECGPU_SECRET_SCALARS;
    int prc = host_pipeline(c, args, 6, n, ops->pass_units(c, 0, 1, fb_flags), true, [&](void** d, size_t cnt) {
      return ops->ecdsa_sign(c, (const uint32_t*)d[0], (const uint32_t*)d[1], (const uint32_t*)d[2], (uint32_t*)d[3], (uint8_t*)d[4], (uint8_t*)d[5], cnt,
                             flags);
    });
    return prc;
  }
  Buf bd, bk, bz, bs, br, bo;
  int rc;
  if (mem == ECGPU_MEM_HOST) { wipe.arm(0, n * nb); wipe.arm(1, n * nb); }
  if ((rc = buf_in(c, bd, 0, secret_d, n * nb, mem))) return rc;
  if ((rc = buf_in(c, bk, 1, nonce_k, n * nb, mem))) return rc;
  if ((rc = buf_in(c, bz, 4, prehash, n * nb, mem))) return rc;
  if ((rc = buf_out(c, bs, 2, sig_rs, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, br, 3, recovery_id, n, mem))) return rc;
  if ((rc = buf_out(c, bo, 5, ok, n, mem))) return rc;
  if ((rc = ops->ecdsa_sign(c, (const uint32_t*)bd.dev, (const uint32_t*)bk.dev, (const uint32_t*)bz.dev, (uint32_t*)bs.dev, (uint8_t*)br.dev,
                            (uint8_t*)bo.dev, n, flags)))
    return rc;
  if ((rc = buf_finish(c, bs))) return rc;
  if ((rc = buf_finish(c, br))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}

int ecgpu_ecdh_batch(ecgpu_ctx* c, int curve, const uint8_t* secret_d, const uint8_t* pubkeys_xy, uint8_t* shared_x, uint8_t* ok, size_t n, int mem) {
  if (!c || !secret_d || !pubkeys_xy || !shared_x || !ok) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  SecretWipe wipe(c);                        // the staged secret scalars are cleared on every exit path
  if (mem == ECGPU_MEM_HOST && n >= PIPE_MIN) {
    wipe.arm_pipeline(0); wipe.arm_pipeline(2);          // secrets, shared values
    const PipeArg args[4] = {{secret_d, nullptr, nb}, {pubkeys_xy, nullptr, 2 * nb}, {nullptr, shared_x, nb}, {nullptr, ok, 1}};
    return host_pipeline(c, args, 4, n, ops->pass_units(c, 1, 1, ECGPU_SECRET_SCALARS), true, [&](void** d, size_t cnt) {
      return ops->ecdh(c, (const uint32_t*)d[0], (const uint32_t*)d[1], (uint32_t*)d[2], (uint8_t*)d[3], cnt);
    });
  }
  Buf bd, bq, bs, bo;
  int rc;
  if (mem == ECGPU_MEM_HOST) { wipe.arm(0, n * nb); wipe.arm(2, n * nb); }   // the staged secrets AND the staged shared values, whichever way the call ends
  if ((rc = buf_in(c, bd, 0, secret_d, n * nb, mem))) return rc;
  if ((rc = buf_in(c, bq, 1, pubkeys_xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bs, 2, shared_x, n * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 3, ok, n, mem))) return rc;
  if ((rc = ops->ecdh(c, (const uint32_t*)bd.dev, (const uint32_t*)bq.dev, (uint32_t*)bs.dev, (uint8_t*)bo.dev, n))) return rc;
  if ((rc = buf_finish(c, bs))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
}

int ecgpu_synth_scalars(ecgpu_ctx* c, int curve, uint64_t seed, uint64_t first, uint8_t* d_scalars, size_t n) {
  if (!c || !d_scalars) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  return ops->synth_scalars(c, seed, first, (uint32_t*)d_scalars, n);
}
int ecgpu_synth_points(ecgpu_ctx* c, int curve, uint64_t seed, uint64_t first, uint8_t* d_points, size_t n) {
  if (!c || !d_points) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if (n == 0) return ECGPU_OK;
  ENTER(c, curve);
  return ops->synth_points(c, seed, first, (uint32_t*)d_points, n);
}

}  // extern "C"

// internal (C++ linkage: not exported, csrc/ecgpu.map): ecgpu_msm with the inputs and the result in different kinds of memory
int ecgpuint_msm_mixed(ecgpu_ctx* c, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, size_t n, uint8_t* out, int out_fmt,
                       int mem_in, int mem_out) {
  return msm_impl(c, curve, scalars, points, pt_fmt, n, out, out_fmt, mem_in, mem_out);
}
