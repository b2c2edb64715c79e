// Shared by the API translation unit (ecgpu.hip) and the per-curve kernel translation units
// (ops_*.hip): context layout, error helpers and the table of kernel launchers of one curve.
// Each curve's kernels are compiled in their own translation unit so the library builds in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>

#include "../../include/ecgpu.h"

struct ecgpu_ctx {
  int device = -1;
  int num_cus = 0;
  hipStream_t own_stream = nullptr;           // blocking stream: ordered with the legacy default stream like any ordinary stream
  hipStream_t stream = nullptr;               // where launches go; nullptr is a legitimate value (the legacy default stream)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_switch = nullptr;            // orders work across ecgpu_set_stream
  char err[512] = {0};
  std::mutex mu;
  std::mutex err_mu;                          // guards err[] (ecgpu_set_err / ecgpu_last_error_copy)
  int64_t opt[ECGPU_OPT_COUNT_] = {0, 26, 0, 0, 1, 0, 4, 0, 0};      // ecgpu_option defaults
  // grow-only device staging buffers for ECGPU_MEM_HOST calls: 6 for whole-batch staging, then PIPE_NSLOT pipeline slots x PIPE_MAXARGS arguments
  static constexpr int PIPE_NSLOT = 3, PIPE_MAXARGS = 6, PIPE_STAGE0 = 6;
  static constexpr int NSTAGE = PIPE_STAGE0 + PIPE_NSLOT * PIPE_MAXARGS;
  void* stage[NSTAGE] = {};
  size_t stage_cap[NSTAGE] = {};
  // chunked host-buffer pipeline (host_pipe.hpp): download and upload streams, per-slot events (inputs arrived / kernels done /
  // outputs read), and the page-locked bounce buffers pageable caller memory is copied through ([0] uploads, [1] downloads)
  static constexpr int PIPE_NWORK = 4;
  static constexpr size_t PIPE_BOUNCE = (size_t)4 << 20;
  hipStream_t copy_stream = nullptr;
  hipStream_t up_stream = nullptr;
  hipEvent_t ev_kernel[PIPE_NSLOT] = {};
  hipEvent_t ev_up[PIPE_NSLOT] = {};
  hipEvent_t ev_down[PIPE_NSLOT] = {};
  void* bounce[2][PIPE_NWORK] = {};
  hipEvent_t ev_bounce[2][PIPE_NWORK] = {};
  // precomputed generator tables, one per curve, built on first use
  void* gen_table[3] = {nullptr, nullptr, nullptr};
  // fixed-base tables of the throughput schedule (fixedbase.hpp)
  void* fb_table[3] = {nullptr, nullptr, nullptr};
  void* fb16_table[3] = {nullptr, nullptr, nullptr};   // 16-bit-window variant for large batches
  void* fb20_table[3] = {nullptr, nullptr, nullptr};   // 20-bit-window variant for very large batches
  void* fbct_table[3] = {nullptr, nullptr, nullptr};   // 5-bit windows, read in full by the constant-time kernel (signing)
  void* fb24_table[3] = {nullptr, nullptr, nullptr};   // 24-bit windows (5.9 GB for a 256-bit curve): batches of 2^23 and more
  void* fb26_table[3] = {nullptr, nullptr, nullptr};   // 26-bit windows (21.5 GB): batches of 2^24 and more on the 256-bit curves
  size_t fb_bytes[3] = {0, 0, 0};                      // device memory held by the generator tables of a curve
  int fb_widest[3] = {0, 0, 0};
  // per-lane table workspace of the k256 variable-base kernel (grow-only)
  void* tab_ws = nullptr;
  size_t tab_ws_cap = 0;
  // MSM workspace (grow-only)
  void* msm_ws = nullptr;
  size_t msm_ws_cap = 0;
  // work counters of the dynamically scheduled kernels (sched.hpp): a small ring, one 8-byte counter per launch, zeroed on the stream before it
  unsigned long long* sched_ctr = nullptr;
  unsigned sched_next = 0;
  // intermediate scalars / points of the ECDSA pipelines (grow-only)
  void* ecdsa_ws = nullptr;
  size_t ecdsa_ws_cap = 0;
};

static inline int ecgpu_set_err(ecgpu_ctx* c, int code, const char* fmt, ...) {
  if (c) {
    std::lock_guard<std::mutex> lk(c->err_mu);
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
  }
  return code;
}
#define HIPCHK(c, call)                                                                                        \
  do {                                                                                                         \
    hipError_t e_ = (call);                                                                                    \
    if (e_ != hipSuccess) return ecgpu_set_err(c, ECGPU_ERR_RUNTIME, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

// the counter of the next dynamically scheduled launch (sched.hpp), zeroed on the context's stream; nullptr on failure (error text set)
static inline unsigned long long* ecgpu_sched_counter(ecgpu_ctx* c) {
  constexpr unsigned RING = 64;
  if (!c->sched_ctr) {
    if (hipMalloc((void**)&c->sched_ctr, RING * sizeof(unsigned long long)) != hipSuccess) { ecgpu_set_err(c, ECGPU_ERR_RUNTIME, "hipMalloc of the work counters failed"); return nullptr; }
  }
  unsigned long long* p = c->sched_ctr + (c->sched_next++ % RING);
  if (hipMemsetAsync(p, 0, sizeof(unsigned long long), c->stream) != hipSuccess) { ecgpu_set_err(c, ECGPU_ERR_RUNTIME, "hipMemsetAsync of a work counter failed"); return nullptr; }
  return p;
}
static inline unsigned ecgpu_grid_for(const ecgpu_ctx* c, size_t n, int per_cu) {
  size_t blocks = (n + 255) / 256;
  size_t cap = (size_t)c->num_cus * per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks == 0) blocks = 1;
  return (unsigned)blocks;
}

// Grid of a kernel with STATIC work (lane t takes units t, t + T, ...) and one inversion per `batch` results of a lane: up to `max_mult` times the
// resident workgroups, as far as every lane keeps a full batch.  Waves that share a SIMD do not progress evenly, so a grid of exactly the resident
// workgroups ends ragged (work distribution); more, shorter workgroups fill the gaps and the choice depends on the batch size only
// (constant-time kernels cannot draw work dynamically).  Signing, 2^22 per call: 17.6 -> 16.4 ms (k256), 19.0 -> 18.0 ms (P-256); at 2^20, where a lane
// has four results, oversubscribing would halve the inversion's amortisation and loses 1-4 %: the rule leaves that size alone.
static inline unsigned ecgpu_grid_oversubscribed(const ecgpu_ctx* c, size_t n, int per_cu, int batch, int max_mult) {
  const size_t resident = (size_t)c->num_cus * per_cu;
  size_t mult = n / (resident * 256 * (size_t)batch);
  if (mult < 1) mult = 1;
  if (mult > (size_t)max_mult) mult = max_mult;
  size_t blocks = (n + 255) / 256;
  if (blocks > resident * mult) blocks = resident * mult;
  if (blocks == 0) blocks = 1;
  return (unsigned)blocks;
}

// Kernel launchers of one curve; all pointers are device pointers, everything is asynchronous on c->stream.
struct ecgpu_curve_ops {
  int (*field_op)(ecgpu_ctx* c, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n);
  int (*point_op)(ecgpu_ctx* c, int op, const uint32_t* p, const uint32_t* q, uint32_t* out, size_t n);
  int (*point_eq)(ecgpu_ctx* c, const uint32_t* p, const uint32_t* q, uint8_t* eq, size_t n);
  int (*normalize)(ecgpu_ctx* c, const uint32_t* p, uint32_t* out_xy, uint8_t* out_inf, size_t n);
  int (*lincomb)(ecgpu_ctx* c, const uint32_t* scalars, const uint32_t* points, int pt_fmt, size_t terms, uint32_t* out,
                 int out_fmt, uint8_t* out_inf, size_t n, unsigned flags);
  int (*msm)(ecgpu_ctx* c, const uint32_t* scalars, const uint32_t* points, int pt_fmt, size_t n, uint32_t* out, int out_fmt);
  int (*validate_scalars)(ecgpu_ctx* c, const uint32_t* scalars, uint8_t* ok, size_t n, size_t terms);
  int (*validate_points)(ecgpu_ctx* c, const uint32_t* xy, uint8_t* ok, size_t n);
  int (*decompress)(ecgpu_ctx* c, const uint32_t* x, const uint8_t* y_is_odd, uint32_t* out_xy, uint8_t* ok, size_t n);
  int (*synth_scalars)(ecgpu_ctx* c, uint64_t seed, uint64_t first, uint32_t* out, size_t n);
  int (*synth_points)(ecgpu_ctx* c, uint64_t seed, uint64_t first, uint32_t* out_xy, size_t n);
  int (*to_bytes)(ecgpu_ctx* c, const uint32_t* pts, int pt_fmt, uint8_t* out, size_t n);
  int (*from_bytes)(ecgpu_ctx* c, const uint8_t* in, uint32_t* out_xy, uint8_t* ok, size_t n);
  int (*sec1_encode)(ecgpu_ctx* c, const uint32_t* pts, int pt_fmt, int compress, uint8_t* out, size_t n);
  int (*sec1_decode)(ecgpu_ctx* c, const uint8_t* in, size_t record_bytes, uint32_t* out_xy, uint8_t* ok, size_t n);
  int (*ecdsa_verify)(ecgpu_ctx* c, const uint32_t* z, const uint32_t* sig, const uint32_t* q_xy, uint8_t* ok, size_t n, unsigned flags);
  int (*h2c_map)(ecgpu_ctx* c, const uint32_t* u, int count, uint32_t* out_xy, uint8_t* out_inf, size_t n);
  int (*ecdsa_recover)(ecgpu_ctx* c, const uint32_t* z, const uint32_t* sig, const uint8_t* recid, uint32_t* out_xy, uint8_t* ok, size_t n,
                       unsigned flags);
  int (*schnorr_verify)(ecgpu_ctx* c, const uint32_t* px, const uint32_t* sig, const uint32_t* e, uint8_t* ok, size_t n);
  int (*ecdsa_sign)(ecgpu_ctx* c, const uint32_t* d, const uint32_t* k, const uint32_t* z, uint32_t* sig, uint8_t* recid, uint8_t* ok,
                    size_t n, unsigned flags);
  int (*ecdh)(ecgpu_ctx* c, const uint32_t* d, const uint32_t* q_xy, uint32_t* shared_x, uint8_t* ok, size_t n);
  // units of one whole pass of the kernel `lincomb` would pick: every resident lane gets its full sub-batch (the results that share
  // one inversion).  The host-buffer pipeline sizes its chunks by it (host_pipe.hpp).
  size_t (*pass_units)(const ecgpu_ctx* c, int has_points, size_t terms, unsigned flags);
};
// Pippenger MSM, one translation unit per curve (msm_*.hip); `mul` is the curve's batch scalar multiplication (affine in / out
// on device memory), used for small sums
typedef int (*ecgpu_msm_mul_fn)(ecgpu_ctx* c, const uint32_t* scalars, const uint32_t* points, int pt_fmt, uint32_t* out_xy, size_t n);
int ecgpu_msm_k256(ecgpu_ctx* c, const uint32_t* sc, const uint32_t* pts, int pt_fmt, size_t n, uint32_t* out, int out_fmt, ecgpu_msm_mul_fn mul);
int ecgpu_msm_p256(ecgpu_ctx* c, const uint32_t* sc, const uint32_t* pts, int pt_fmt, size_t n, uint32_t* out, int out_fmt, ecgpu_msm_mul_fn mul);
int ecgpu_msm_p384(ecgpu_ctx* c, const uint32_t* sc, const uint32_t* pts, int pt_fmt, size_t n, uint32_t* out, int out_fmt, ecgpu_msm_mul_fn mul);
const ecgpu_curve_ops* ecgpu_ops_k256();
const ecgpu_curve_ops* ecgpu_ops_p256();
const ecgpu_curve_ops* ecgpu_ops_p384();
// ecgpu_msm with the inputs and the result in different kinds of memory (ecgpu.hip; used by the device group, group.hip)
int ecgpuint_msm_mixed(ecgpu_ctx* c, int curve, const uint8_t* scalars, const uint8_t* points, int pt_fmt, size_t n, uint8_t* out, int out_fmt,
                       int mem_in, int mem_out);
// secp256k1 kernels that run on SECRET scalars live in their own translation unit (ops_k256_ct.hip), compiled with
// ECGPU_K256_BRANCHFREE: the field additions / subtractions / folds there ripple their rare carries unconditionally, so no branch
// depends on data.  These launchers are what the generic code (curve_ops.hpp) calls for curve 0.
int ecgpuint_k256_mul_ct(ecgpu_ctx* c, const uint32_t* sc, const uint32_t* pts, int pt_fmt, uint32_t* out, int out_fmt, uint8_t* out_inf, size_t n);        // k P, varbase_ct_k256.hpp
int ecgpuint_k256_mul_gen_ct(ecgpu_ctx* c, const uint32_t* sc, const void* table, uint32_t* out, int out_fmt, uint8_t* out_inf, size_t n);                 // k G, fixedbase_ct.hpp
// the reference schedules (exact X, Y, Z; constant-time table scans): points == NULL is mul_by_generator over `gen_table`
int ecgpuint_k256_reference(ecgpu_ctx* c, const uint32_t* sc, const uint32_t* pts, int pt_fmt, size_t terms, const void* gen_table, uint32_t* out, int out_fmt,
                            uint8_t* out_inf, size_t n);
size_t ecgpuint_k256_ct_pass_units(const ecgpu_ctx* c);
