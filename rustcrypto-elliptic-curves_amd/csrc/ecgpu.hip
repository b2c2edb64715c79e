// C ABI of libecgpu.so (include/ecgpu.h): context, staging of host buffers, dispatch to the
// per-curve kernel launchers (ops_*.hip).  Host side of the boundary; all arithmetic happens in
// the gfx950 kernels.  There is deliberately no CPU code path in this library.
#include <string.h>
#include <condition_variable>
#include <thread>
#include <vector>

#include "ecgpu_internal.hpp"

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
// Host-buffer calls on large batches: chunks of PIPE_CHUNK units flow through two device slots so that the
// upload of chunk i+1, the kernels of chunk i and the download of chunk i-1 overlap: uploads run on their own stream (the
// chunk's kernels wait for its event; until late in round 3 they were queued on the compute stream itself, behind the previous
// chunk's kernels, and only the downloads overlapped: 8.5 x 10^7 /s from page-locked buffers against 1.2 x 10^8 device-resident).
// Pageable host memory makes hipMemcpyAsync block its caller, hence the download runs on a helper thread (own stream, ordered
// after the chunk's kernels by an event); the calling thread uploads and launches.  Element i of every argument must depend
// only on element i of the inputs (true for every batch entry point that uses this).
// ---------------------------------------------------------------------------------------------
static constexpr size_t PIPE_CHUNK = (size_t)1 << 20;
static constexpr int PIPE_MAXARGS = 6;
struct PipeArg {
  const void* in;      // host input  (or nullptr)
  void* out;           // host output (or nullptr)
  size_t unit;         // bytes per batch element
};
template <class Launch>
static int host_pipeline(ecgpu_ctx* c, const PipeArg* args, int nargs, size_t n, Launch launch) {
  if (nargs > PIPE_MAXARGS) return ecgpu_set_err(c, ECGPU_ERR_ARG, "host_pipeline: too many arguments");
  if (!c->copy_stream) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPCHK(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_kernel[i], hipEventDisableTiming));
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming));
    }
  }
  for (int s = 0; s < 2; s++)
    for (int a = 0; a < nargs; a++) {
      if (!args[a].in && !args[a].out) continue;
      int rc = stage_reserve(c, 6 + s * PIPE_MAXARGS + a, PIPE_CHUNK * args[a].unit);
      if (rc) return rc;
    }
  const size_t nchunks = (n + PIPE_CHUNK - 1) / PIPE_CHUNK;
  std::mutex mu;
  std::condition_variable cv;
  size_t launched = 0, drained = 0;        // chunks whose kernels are enqueued / whose outputs are back on the host
  hipError_t copy_err = hipSuccess;
  bool abort_flag = false;
  std::thread drain([&] {
    (void)hipSetDevice(c->device);
    for (size_t ci = 0; ci < nchunks; ci++) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return launched > ci || abort_flag; });
        if (abort_flag) return;
      }
      const int slot = (int)(ci & 1);
      const size_t lo = ci * PIPE_CHUNK, cnt = (n - lo < PIPE_CHUNK) ? n - lo : PIPE_CHUNK;
      hipError_t e = hipStreamWaitEvent(c->copy_stream, c->ev_kernel[slot], 0);
      for (int a = 0; a < nargs && e == hipSuccess; a++)
        if (args[a].out)
          e = hipMemcpyAsync((char*)args[a].out + lo * args[a].unit, c->stage[6 + slot * PIPE_MAXARGS + a], cnt * args[a].unit,
                             hipMemcpyDeviceToHost, c->copy_stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->copy_stream);
      std::lock_guard<std::mutex> lk(mu);
      if (e != hipSuccess && copy_err == hipSuccess) copy_err = e;
      drained = ci + 1;
      cv.notify_all();
    }
  });
  int rc = 0;
  hipError_t up_err = hipSuccess;
  for (size_t ci = 0; ci < nchunks && rc == 0 && up_err == hipSuccess; ci++) {
    const int slot = (int)(ci & 1);
    if (ci >= 2) {                         // the slot is free once chunk ci-2 has been downloaded
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return drained + 2 > ci; });
    }
    const size_t lo = ci * PIPE_CHUNK, cnt = (n - lo < PIPE_CHUNK) ? n - lo : PIPE_CHUNK;
    void* dev[PIPE_MAXARGS];
    for (int a = 0; a < nargs; a++) {
      dev[a] = (args[a].in || args[a].out) ? c->stage[6 + slot * PIPE_MAXARGS + a] : nullptr;
      if (args[a].in && up_err == hipSuccess)
        up_err = hipMemcpyAsync(dev[a], (const char*)args[a].in + lo * args[a].unit, cnt * args[a].unit, hipMemcpyHostToDevice, c->up_stream);
    }
    // the chunk's kernels start when its inputs have arrived; the slot itself is free (chunk ci-2 has been drained, see above)
    if (up_err == hipSuccess) up_err = hipEventRecord(c->ev_up[slot], c->up_stream);
    if (up_err == hipSuccess) up_err = hipStreamWaitEvent(c->stream, c->ev_up[slot], 0);
    if (up_err == hipSuccess) rc = launch(dev, cnt);
    if (rc == 0 && up_err == hipSuccess) up_err = hipEventRecord(c->ev_kernel[slot], c->stream);
    {
      std::lock_guard<std::mutex> lk(mu);
      if (rc != 0 || up_err != hipSuccess) abort_flag = true; else launched = ci + 1;
      cv.notify_all();
    }
  }
  drain.join();
  (void)hipStreamSynchronize(c->up_stream);          // an aborted run may still have an upload in flight
  (void)hipStreamSynchronize(c->stream);
  if (rc) return rc;
  if (up_err != hipSuccess) return ecgpu_set_err(c, ECGPU_ERR_RUNTIME, "host pipeline upload: %s", hipGetErrorString(up_err));
  if (copy_err != hipSuccess) return ecgpu_set_err(c, ECGPU_ERR_RUNTIME, "host pipeline download: %s", hipGetErrorString(copy_err));
  return 0;
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
  for (int i = 0; i < 2; i++) if (c->ev_kernel[i]) (void)hipEventDestroy(c->ev_kernel[i]);
  for (int i = 0; i < 2; i++) if (c->ev_up[i]) (void)hipEventDestroy(c->ev_up[i]);
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
  int slot[4];
  size_t bytes[4];
  int cnt = 0;
  explicit SecretWipe(ecgpu_ctx* ctx) : c(ctx) {}
  void arm(int s, size_t b) { if (cnt < 4) { slot[cnt] = s; bytes[cnt] = b; cnt++; } }
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
  if (mem == ECGPU_MEM_HOST && n >= 2 * PIPE_CHUNK) {
    if (secret) for (int sl = 0; sl < 2; sl++) wipe.arm(6 + sl * PIPE_MAXARGS, PIPE_CHUNK * terms * nb);
    const PipeArg args[5] = {{scalars, nullptr, terms * nb}, {points, nullptr, points ? terms * pin : 0}, {nullptr, out, pout},
                             {nullptr, out_fmt == ECGPU_PT_AFFINE ? out_inf : nullptr, 1}, {nullptr, scalar_ok, 1}};
    int rc = host_pipeline(c, args, 5, n, [&](void** d, size_t cnt) {
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
  if (secret && mem == ECGPU_MEM_HOST) wipe.arm(0, n * terms * nb);
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

int ecgpu_msm(ecgpu_ctx* c, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, size_t n, uint8_t* out, int out_fmt,
              int mem) {
  if (!c || !out || (n && (!scalars || !points))) return ecgpu_set_err(c, ECGPU_ERR_ARG, "null argument");
  if ((pt_fmt != ECGPU_PT_AFFINE && pt_fmt != ECGPU_PT_PROJECTIVE) || (out_fmt != ECGPU_PT_AFFINE && out_fmt != ECGPU_PT_PROJECTIVE))
    return ecgpu_set_err(c, ECGPU_ERR_ARG, "bad point format");
  ENTER(c, curve);
  const size_t pin = (pt_fmt == ECGPU_PT_PROJECTIVE ? 3 : 2) * nb, pout = (out_fmt == ECGPU_PT_PROJECTIVE ? 3 : 2) * nb;
  Buf bs, bp, bo;
  int rc;
  if (n == 0) {                              // the empty sum is the identity: affine zeros, projective (0 : 1 : 0)
    if (mem == ECGPU_MEM_HOST) {
      memset(out, 0, pout);
      if (out_fmt == ECGPU_PT_PROJECTIVE) out[2 * nb - 1] = 1;
      return ECGPU_OK;
    }
    HIPCHK(c, hipMemsetAsync(out, 0, pout, c->stream));
    if (out_fmt == ECGPU_PT_PROJECTIVE) HIPCHK(c, hipMemsetAsync(out + 2 * nb - 1, 1, 1, c->stream));
    return ECGPU_OK;
  }
  if ((rc = buf_in(c, bs, 0, scalars, n * nb, mem))) return rc;
  if ((rc = buf_in(c, bp, 1, points, n * pin, mem))) return rc;
  if ((rc = buf_out(c, bo, 2, out, pout, mem))) return rc;
  if ((rc = ops->msm(c, (const uint32_t*)bs.dev, (const uint32_t*)bp.dev, pt_fmt, n, (uint32_t*)bo.dev, out_fmt))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  return finish_host(c, mem);
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
  if (mem == ECGPU_MEM_HOST && n >= 2 * PIPE_CHUNK) {
    const PipeArg args[4] = {{prehash, nullptr, nb}, {sig_rs, nullptr, 2 * nb}, {pubkeys_xy, nullptr, 2 * nb}, {nullptr, ok, 1}};
    return host_pipeline(c, args, 4, n, [&](void** d, size_t cnt) {
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
  if (mem == ECGPU_MEM_HOST && n >= 2 * PIPE_CHUNK) {
    const PipeArg args[5] = {{prehash, nullptr, nb}, {sig_rs, nullptr, 2 * nb}, {recovery_id, nullptr, 1}, {nullptr, pubkeys_xy, 2 * nb}, {nullptr, ok, 1}};
    return host_pipeline(c, args, 5, n, [&](void** d, size_t cnt) {
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
  if (mem == ECGPU_MEM_HOST && n >= 2 * PIPE_CHUNK) {
    const PipeArg args[4] = {{pubkeys_x, nullptr, nb}, {sig_rs, nullptr, 2 * nb}, {challenges, nullptr, nb}, {nullptr, ok, 1}};
    return host_pipeline(c, args, 4, n, [&](void** d, size_t cnt) {
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
  if (mem == ECGPU_MEM_HOST && n >= 2 * PIPE_CHUNK) {
    for (int sl = 0; sl < 2; sl++)
      for (int a = 0; a < 2; a++) wipe.arm(6 + sl * PIPE_MAXARGS + a, PIPE_CHUNK * nb);
    const PipeArg args[6] = {{secret_d, nullptr, nb}, {nonce_k, nullptr, nb}, {prehash, nullptr, nb}, {nullptr, sig_rs, 2 * nb},
                             {nullptr, recovery_id, 1}, {nullptr, ok, 1}};
    int prc = host_pipeline(c, args, 6, n, [&](void** d, size_t cnt) {
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
  if (mem == ECGPU_MEM_HOST && n >= 2 * PIPE_CHUNK) {
    for (int sl = 0; sl < 2; sl++) { wipe.arm(6 + sl * PIPE_MAXARGS, PIPE_CHUNK * nb); wipe.arm(6 + sl * PIPE_MAXARGS + 2, PIPE_CHUNK * nb); }   // secrets, shared values
    const PipeArg args[4] = {{secret_d, nullptr, nb}, {pubkeys_xy, nullptr, 2 * nb}, {nullptr, shared_x, nb}, {nullptr, ok, 1}};
    return host_pipeline(c, args, 4, n, [&](void** d, size_t cnt) {
      return ops->ecdh(c, (const uint32_t*)d[0], (const uint32_t*)d[1], (uint32_t*)d[2], (uint8_t*)d[3], cnt);
    });
  }
  Buf bd, bq, bs, bo;
  int rc;
  if (mem == ECGPU_MEM_HOST) wipe.arm(0, n * nb);
  if ((rc = buf_in(c, bd, 0, secret_d, n * nb, mem))) return rc;
  if ((rc = buf_in(c, bq, 1, pubkeys_xy, n * 2 * nb, mem))) return rc;
  if ((rc = buf_out(c, bs, 2, shared_x, n * nb, mem))) return rc;
  if ((rc = buf_out(c, bo, 3, ok, n, mem))) return rc;
  if ((rc = ops->ecdh(c, (const uint32_t*)bd.dev, (const uint32_t*)bq.dev, (uint32_t*)bs.dev, (uint8_t*)bo.dev, n))) return rc;
  if ((rc = buf_finish(c, bs))) return rc;
  if ((rc = buf_finish(c, bo))) return rc;
  if (mem == ECGPU_MEM_HOST) wipe.arm(2, n * nb);   // the staged copy of the shared secrets goes as well (after the download, ordered on the stream)
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
