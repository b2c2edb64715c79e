// Device group behind the C ABI (include/ecgpu.h, "device groups"): one call, several GPUs.
//
// The reference's bulk entry is ONE call - LinearCombinationExt::lincomb_ext(&[(P, k)]) (k256/src/arithmetic/mul.rs:325-340) -
// and BASELINE.json's config 4 is one 2^26-term sum over 8 GPUs, so the multi-device split has to live on this side of the
// boundary, not in a Python launcher (missing 1).  SURVEY.md section 8(e):
//   * independent batches (configs 2, 3, 5): contiguous index ranges, one host thread + context + stream per device, generator
//     tables per device, NO collective;
//   * one split sum (config 4): every device runs the whole bucket method over its range of terms -> one projective point per
//     device -> all-gather of those points (RCCL ncclAllGather over xGMI when the devices are distinct: 96 / 144 bytes per device,
//     latency-bound; elliptic-curve addition is not an RCCL reduction operator, so all-reduce does not apply) -> the leader
//     (device 0 of the group) adds them up with the library's own point arithmetic.
// RCCL is loaded at first use with dlopen (librccl.so.1): single-device users carry no dependency on it, and in a process that has
// PyTorch loaded the same copy is shared.  A group whose devices are not distinct (two contexts on one card: how a one-GPU box
// exercises the threading and the fold), or a system without RCCL, gathers through host memory instead - at 96 bytes per device
// that costs nothing.
#include <dlfcn.h>
#include <string.h>
#include <thread>
#include <vector>

#include <rccl/rccl.h>        // types and prototypes only: the functions are resolved with dlsym

#include "ecgpu_internal.hpp"

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load() {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd && GetErrorString) return true;
    dlclose(lib);
    lib = nullptr;
    return false;
  }
};

constexpr size_t MAX_PT = 3 * 48;             // bytes of a projective point of the widest curve

}  // namespace

struct ecgpu_group {
  std::vector<ecgpu_ctx*> ctx;
  std::vector<int> dev;
  unsigned flags = 0;
  bool distinct = true;
  std::mutex mu;                                // one group call at a time
  char err[640] = {0};
  Rccl rccl;
  std::vector<ncclComm_t> comms;
  int rccl_state = 0;                           // 0 not tried, 1 usable, -1 unavailable (gather through the host)
  char gather_path[64] = "none yet";
  std::vector<void*> d_part, d_all;             // per device: its partial sum; the partial sums of all devices
  void* d_fold = nullptr;                       // leader: scratch of the fold (as large as d_all)
  void* d_ones = nullptr;                       // leader: the normalised result (x || y || infinity flag)
};

static int group_err(ecgpu_group* g, int code, const char* fmt, ...) {
  if (g) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g->err, sizeof(g->err), fmt, ap);
    va_end(ap);
  }
  return code;
}
static int device_err(ecgpu_group* g, int i, int code) {
  char buf[512];
  (void)ecgpu_last_error_copy(g->ctx[i], buf, sizeof(buf));
  return group_err(g, code, "device %d (group member %d): %s", g->dev[i], i, buf);
}
#define GHIP(g, call)                                                                                      \
  do {                                                                                                     \
    hipError_t e_ = (call);                                                                                \
    if (e_ != hipSuccess) return group_err(g, ECGPU_ERR_RUNTIME, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

// run fn(i) for every member: member 0 on the calling thread, the others on a thread each; -> first failing member or -1
template <class Fn>
static int for_each_member(ecgpu_group* g, std::vector<int>& rc, Fn fn) {
  const int k = (int)g->ctx.size();
  rc.assign(k, 0);
  std::vector<std::thread> th;
  for (int i = 1; i < k; i++) th.emplace_back([&, i] { rc[i] = fn(i); });
  rc[0] = fn(0);
  for (auto& t : th) t.join();
  for (int i = 0; i < k; i++)
    if (rc[i]) return i;
  return -1;
}

extern "C" {

int ecgpu_shard_range(size_t n, int parts, int index, size_t* first, size_t* count) {
  if (parts <= 0 || index < 0 || index >= parts || !first || !count) return ECGPU_ERR_ARG;
  // balanced contiguous ranges: the first n mod parts members get one element more
  const size_t q = n / (size_t)parts, r = n % (size_t)parts, i = (size_t)index;
  *first = i * q + (i < r ? i : r);
  *count = q + (i < r ? 1 : 0);
  return ECGPU_OK;
}

int ecgpu_group_create(ecgpu_group** out, const int* devices, int n_devices, unsigned flags) {
  if (!out) return ECGPU_ERR_ARG;
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64) return ECGPU_ERR_ARG;
  ecgpu_group* g = new ecgpu_group();
  g->flags = flags;
  for (int i = 0; i < n_devices; i++) {
    ecgpu_ctx* c = nullptr;
    int rc = ecgpu_create(&c, devices[i]);
    if (rc != ECGPU_OK) {
      for (ecgpu_ctx* p : g->ctx) ecgpu_destroy(p);
      delete g;
      return rc;
    }
    g->ctx.push_back(c);
    g->dev.push_back(devices[i]);
    for (int j = 0; j < i; j++)
      if (devices[j] == devices[i]) g->distinct = false;
  }
  g->d_part.assign(n_devices, nullptr);
  g->d_all.assign(n_devices, nullptr);
  bool ok = true;
  for (int i = 0; i < n_devices && ok; i++) {
    ok = hipSetDevice(devices[i]) == hipSuccess && hipMalloc(&g->d_part[i], MAX_PT) == hipSuccess && hipMalloc(&g->d_all[i], MAX_PT * n_devices) == hipSuccess;
  }
  ok = ok && hipSetDevice(devices[0]) == hipSuccess && hipMalloc(&g->d_ones, 256) == hipSuccess && hipMalloc(&g->d_fold, MAX_PT * n_devices) == hipSuccess;
  if (!ok) {
    ecgpu_group_destroy(g);
    return ECGPU_ERR_RUNTIME;
  }
  *out = g;
  return ECGPU_OK;
}

void ecgpu_group_destroy(ecgpu_group* g) {
  if (!g) return;
  for (size_t i = 0; i < g->comms.size(); i++)
    if (g->comms[i]) (void)g->rccl.CommDestroy(g->comms[i]);
  for (size_t i = 0; i < g->ctx.size(); i++) {
    (void)hipSetDevice(g->dev[i]);
    if (i < g->d_part.size() && g->d_part[i]) (void)hipFree(g->d_part[i]);
    if (i < g->d_all.size() && g->d_all[i]) (void)hipFree(g->d_all[i]);
    if (i == 0 && g->d_ones) (void)hipFree(g->d_ones);
    if (i == 0 && g->d_fold) (void)hipFree(g->d_fold);
  }
  for (ecgpu_ctx* c : g->ctx) ecgpu_destroy(c);
  delete g;
}

int ecgpu_group_size(const ecgpu_group* g) { return g ? (int)g->ctx.size() : 0; }
ecgpu_ctx* ecgpu_group_context(ecgpu_group* g, int index) { return (g && index >= 0 && index < (int)g->ctx.size()) ? g->ctx[index] : nullptr; }
const char* ecgpu_group_last_error(const ecgpu_group* g) { return g ? g->err : "null group"; }
const char* ecgpu_group_gather_path(const ecgpu_group* g) { return g ? g->gather_path : "null group"; }

int ecgpu_group_synchronize(ecgpu_group* g) {
  if (!g) return ECGPU_ERR_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  for (size_t i = 0; i < g->ctx.size(); i++) {
    int rc = ecgpu_synchronize(g->ctx[i]);
    if (rc) return device_err(g, (int)i, rc);
  }
  return ECGPU_OK;
}

// ---- independent batches: contiguous index ranges, no collective ---------------------------------------------------------
int ecgpu_group_lincomb_batch(ecgpu_group* g, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, size_t terms, uint8_t* out, int out_fmt,
                              uint8_t* out_inf, size_t n, unsigned flags) {
  if (!g) return ECGPU_ERR_ARG;
  const size_t nb = ecgpu_field_bytes(curve);
  if (!nb) return group_err(g, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  if (!scalars || !out || terms == 0) return group_err(g, ECGPU_ERR_ARG, "null argument or zero terms");
  std::lock_guard<std::mutex> lk(g->mu);
  const int k = (int)g->ctx.size();
  const size_t pin = (pt_fmt == ECGPU_PT_PROJECTIVE ? 3 : 2) * nb, pout = (out_fmt == ECGPU_PT_PROJECTIVE ? 3 : 2) * nb;
  std::vector<int> rc;
  const int bad = for_each_member(g, rc, [&](int i) {
    size_t lo, cnt;
    (void)ecgpu_shard_range(n, k, i, &lo, &cnt);
    if (!cnt) return 0;
    return ecgpu_lincomb_batch(g->ctx[i], curve, scalars + lo * terms * nb, points ? points + lo * terms * pin : nullptr, pt_fmt, terms, out + lo * pout, out_fmt,
                               out_inf ? out_inf + lo : nullptr, cnt, ECGPU_MEM_HOST, flags);
  });
  return bad < 0 ? ECGPU_OK : device_err(g, bad, rc[bad]);
}
int ecgpu_group_mul_batch(ecgpu_group* g, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, uint8_t* out, int out_fmt, uint8_t* out_inf,
                          size_t n, unsigned flags) {
  return ecgpu_group_lincomb_batch(g, curve, scalars, points, pt_fmt, 1, out, out_fmt, out_inf, n, flags);
}
// device-resident shards: member i's pointers live on ITS device; asynchronous on every member's stream (ecgpu_group_synchronize)
int ecgpu_group_lincomb_sharded(ecgpu_group* g, int curve, const uint8_t* const* scalars, const uint8_t* const* points, int pt_fmt, size_t terms,
                                uint8_t* const* out, int out_fmt, uint8_t* const* out_inf, const size_t* counts, unsigned flags) {
  if (!g) return ECGPU_ERR_ARG;
  if (!scalars || !out || !counts || terms == 0) return group_err(g, ECGPU_ERR_ARG, "null argument or zero terms");
  std::lock_guard<std::mutex> lk(g->mu);
  std::vector<int> rc;
  const int bad = for_each_member(g, rc, [&](int i) {
    if (!counts[i]) return 0;
    return ecgpu_lincomb_batch(g->ctx[i], curve, scalars[i], points ? points[i] : nullptr, pt_fmt, terms, out[i], out_fmt, out_inf ? out_inf[i] : nullptr, counts[i],
                               ECGPU_MEM_DEVICE, flags);
  });
  return bad < 0 ? ECGPU_OK : device_err(g, bad, rc[bad]);
}

// ---- one split sum ------------------------------------------------------------------------------------------------------------
// every member has left its partial sum (projective, 3 nb bytes) in d_part[i], queued on its stream: gather them on the leader
// and add them up into `out` (host memory)
static int gather_and_fold(ecgpu_group* g, int curve, uint8_t* out, int out_fmt) {
  const size_t nb = ecgpu_field_bytes(curve), pt = 3 * nb;
  const int k = (int)g->ctx.size();
  if (g->rccl_state == 0) {
    g->rccl_state = -1;
    if (g->distinct && !(g->flags & ECGPU_GROUP_NO_RCCL) && g->rccl.load()) {
      g->comms.assign(k, nullptr);
      ncclResult_t r = g->rccl.CommInitAll(g->comms.data(), k, g->dev.data());
      if (r == ncclSuccess) g->rccl_state = 1;
      else g->comms.clear();
    }
  }
  if (g->rccl_state == 1) {
    snprintf(g->gather_path, sizeof(g->gather_path), "rccl ncclAllGather, %d rank(s)", k);
    ncclResult_t r = g->rccl.GroupStart();
    for (int i = 0; i < k && r == ncclSuccess; i++) {                  // no early return in here: the group must be closed whatever happens
      if (hipSetDevice(g->dev[i]) != hipSuccess) { r = ncclUnhandledCudaError; break; }
      r = g->rccl.AllGather(g->d_part[i], g->d_all[i], pt, ncclUint8, g->comms[i], g->ctx[i]->stream);
    }
    ncclResult_t r2 = g->rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return group_err(g, ECGPU_ERR_RUNTIME, "ncclAllGather of the partial sums: %s", g->rccl.GetErrorString(r));
  } else {
    snprintf(g->gather_path, sizeof(g->gather_path), "host copy, %d member(s)%s", k, g->distinct ? "" : " (devices not distinct)");
    uint8_t tmp[64 * MAX_PT];
    for (int i = 0; i < k; i++) {
      int rc = ecgpu_synchronize(g->ctx[i]);
      if (rc) return device_err(g, i, rc);
      GHIP(g, hipSetDevice(g->dev[i]));
      GHIP(g, hipMemcpy(tmp + (size_t)i * pt, g->d_part[i], pt, hipMemcpyDeviceToHost));
    }
    GHIP(g, hipSetDevice(g->dev[0]));
    GHIP(g, hipMemcpy(g->d_all[0], tmp, (size_t)k * pt, hipMemcpyHostToDevice));
  }
  // The leader adds the k points by a tree of complete additions (ceil(log2 k) small launches; a sum with unit scalars would spend a
  // whole scalar multiplication's latency, ~1 ms, on it) and normalises the result; d_all[0] was written on the leader's stream
  // (RCCL) or by a blocking copy, and everything below is queued on that stream.  The odd point of a level is left where it is -
  // later levels write below it - and added at the end.
  ecgpu_ctx* c0 = g->ctx[0];
  uint8_t *cur = (uint8_t*)g->d_all[0], *other = (uint8_t*)g->d_fold;
  int count = k, nleft = 0, rc = 0;
  const uint8_t* left[8];
  while (count > 1 && !rc) {
    const int half = count / 2;
    if (count & 1) left[nleft++] = cur + (size_t)(count - 1) * pt;
    rc = ecgpu_point_add_batch(c0, curve, cur, cur + (size_t)half * pt, other, (size_t)half, ECGPU_MEM_DEVICE);
    count = half;
    uint8_t* t = cur; cur = other; other = t;
  }
  for (int j = 0; j < nleft && !rc; j++) {
    rc = ecgpu_point_add_batch(c0, curve, cur, left[j], other, 1, ECGPU_MEM_DEVICE);
    uint8_t* t = cur; cur = other; other = t;
  }
  uint8_t* d_xy = (uint8_t*)g->d_ones;                          // 2 nb bytes of x || y, then the infinity flag
  if (!rc) rc = ecgpu_batch_normalize(c0, curve, cur, d_xy, d_xy + 2 * nb, 1, ECGPU_MEM_DEVICE);
  if (rc) return device_err(g, 0, rc);
  uint8_t res[2 * 48 + 1];
  GHIP(g, hipSetDevice(g->dev[0]));
  GHIP(g, hipMemcpyAsync(res, d_xy, 2 * nb + 1, hipMemcpyDeviceToHost, c0->stream));
  GHIP(g, hipStreamSynchronize(c0->stream));
  // formatted as ecgpu_msm formats its result: affine x || y (zeros for the identity), projective (x : y : 1) or (0 : 1 : 0)
  const bool inf = res[2 * nb] != 0;
  memset(out, 0, (out_fmt == ECGPU_PT_PROJECTIVE ? 3 : 2) * nb);
  if (!inf) memcpy(out, res, 2 * nb);
  if (out_fmt == ECGPU_PT_PROJECTIVE) out[(inf ? 2 : 3) * nb - 1] = 1;
  if (g->rccl_state == 1)                                     // the other members' streams still hold their all-gather: drain them before the buffers are reused
    for (int i = 1; i < k; i++) {
      rc = ecgpu_synchronize(g->ctx[i]);
      if (rc) return device_err(g, i, rc);
    }
  return ECGPU_OK;
}

int ecgpu_group_msm(ecgpu_group* g, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, size_t n, uint8_t* out, int out_fmt) {
  if (!g) return ECGPU_ERR_ARG;
  const size_t nb = ecgpu_field_bytes(curve);
  if (!nb) return group_err(g, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  if (!out || (n && (!scalars || !points))) return group_err(g, ECGPU_ERR_ARG, "null argument");
  if ((pt_fmt != ECGPU_PT_AFFINE && pt_fmt != ECGPU_PT_PROJECTIVE) || (out_fmt != ECGPU_PT_AFFINE && out_fmt != ECGPU_PT_PROJECTIVE))
    return group_err(g, ECGPU_ERR_ARG, "bad point format");
  std::lock_guard<std::mutex> lk(g->mu);
  const int k = (int)g->ctx.size();
  const size_t pin = (pt_fmt == ECGPU_PT_PROJECTIVE ? 3 : 2) * nb;
  std::vector<int> rc;
  const int bad = for_each_member(g, rc, [&](int i) {
    size_t lo, cnt;
    (void)ecgpu_shard_range(n, k, i, &lo, &cnt);
    return ecgpuint_msm_mixed(g->ctx[i], curve, cnt ? scalars + lo * nb : nullptr, cnt ? points + lo * pin : nullptr, pt_fmt, cnt, (uint8_t*)g->d_part[i],
                              ECGPU_PT_PROJECTIVE, ECGPU_MEM_HOST, ECGPU_MEM_DEVICE);
  });
  if (bad >= 0) return device_err(g, bad, rc[bad]);
  return gather_and_fold(g, curve, out, out_fmt);
}
int ecgpu_group_msm_sharded(ecgpu_group* g, int curve, const uint8_t* const* scalars, const uint8_t* const* points, int pt_fmt, const size_t* counts, uint8_t* out,
                            int out_fmt) {
  if (!g) return ECGPU_ERR_ARG;
  const size_t nb = ecgpu_field_bytes(curve);
  if (!nb) return group_err(g, ECGPU_ERR_UNSUPPORTED, "curve %d not supported", curve);
  if (!out || !scalars || !points || !counts) return group_err(g, ECGPU_ERR_ARG, "null argument");
  if ((pt_fmt != ECGPU_PT_AFFINE && pt_fmt != ECGPU_PT_PROJECTIVE) || (out_fmt != ECGPU_PT_AFFINE && out_fmt != ECGPU_PT_PROJECTIVE))
    return group_err(g, ECGPU_ERR_ARG, "bad point format");
  std::lock_guard<std::mutex> lk(g->mu);
  std::vector<int> rc;
  const int bad = for_each_member(g, rc, [&](int i) {
    return ecgpuint_msm_mixed(g->ctx[i], curve, scalars[i], points[i], pt_fmt, counts[i], (uint8_t*)g->d_part[i], ECGPU_PT_PROJECTIVE, ECGPU_MEM_DEVICE, ECGPU_MEM_DEVICE);
  });
  if (bad >= 0) return device_err(g, bad, rc[bad]);
  return gather_and_fold(g, curve, out, out_fmt);
}

}  // extern "C"
