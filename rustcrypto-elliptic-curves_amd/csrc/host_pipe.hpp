// Host-buffer calls on large batches (ECGPU_MEM_HOST): the batch flows through NSLOT device slots in chunks so that the
// upload of chunk i+1, the kernels of chunk i and the download of chunk i-1 overlap.  The reference's trait surface hands
// over host values (k256/src/arithmetic/mul.rs:442-481, projective.rs:337-348: slices in, Vec out), so this is the path a
// drop-in caller actually takes; its PCIe-inclusive rate is reported by bench.py as `host_io`.
//
// Round 4.  What round 3 had: fixed chunks of 2^20 units through two slots - a lane of the k256
// variable-base kernel then shares its output inversion between 4 results instead of 32, eight chunks' fill and drain on top;
// pageable memory went through hipMemcpyAsync's internal staging at ~10 GB/s and blocked the issuing thread
// (6.5-8.7 x 10^7 /s pageable, 9.4-9.7 x 10^7 /s pinned against 1.23 x 10^8 /s device-resident).  Now:
//   * chunk sizes GROW from pass / 8 to one whole pass of the kernel (the number of units that gives every resident lane its
//     full sub-batch: 2^23 for the k256 variable-base kernel) and the LAST chunk is small again: the pipeline's fill is the
//     upload of the small first chunk, its drain the download of the small last one, everything between runs at the
//     device-resident rate;
//   * three slots, three streams (upload / compute / download), ordering by events only - the host never waits for the
//     device between chunks, it only keeps the ISSUE order of records and waits straight (a wait must be issued after the
//     record it refers to);
//   * page-locked caller buffers (ecgpu_host_alloc, hipHostRegister) are copied directly; pageable ones go through a small
//     pool of page-locked bounce buffers filled / drained by NWORK helper threads per direction (memcpy at DRAM rate in
//     parallel with the DMA of the previous piece), which is what hipMemcpyAsync does internally but single-threaded.
// Element i of every argument must depend only on element i of the inputs (true for every batch entry point that uses this);
// `launch(dev, cnt, chunk_index)` enqueues the chunk's kernels on c->stream.
#pragma once
#include <string.h>
#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#include "ecgpu_internal.hpp"

namespace hostpipe {

struct Arg {
  const void* in;      // host input  (or nullptr)
  void* out;           // host output (or nullptr)
  size_t unit;         // bytes per batch element
};

// Chunk sizes: first, 2 first, 4 first, .. up to `pass`, whole passes, and a last chunk of at most `tail` units (a larger
// remainder is split when both parts stay >= tail; tail = 0: no split).  A remainder below `crumb` joins the chunk before it.
static inline std::vector<size_t> schedule(size_t n, size_t first, size_t pass, size_t tail, size_t crumb) {
  std::vector<size_t> sizes;
  if (first < 256) first = 256;
  if (pass < first) pass = first;
  size_t rem = n, s = first;
  while (rem > 0) {
    size_t take = s < rem ? s : rem;
    if (rem - take > 0 && rem - take < crumb) take = rem;                  // no crumbs
    if (take == rem && !sizes.empty() && tail && take >= 2 * tail) {       // a big last chunk: split a small tail off (short drain)
      sizes.push_back(take - tail);
      sizes.push_back(tail);
      break;
    }
    sizes.push_back(take);
    rem -= take;
    s = (2 * s < pass) ? 2 * s : pass;
  }
  return sizes;
}

// true iff p is page-locked host memory known to HIP (hipHostMalloc / hipHostRegister): DMA reads it directly
static inline bool is_pinned(const void* p) {
  if (!p) return true;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();                   // pageable memory is reported as an error by some runtimes: not sticky
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

// A few threads that execute fn(index, worker) for index = 0 .. count-1; parallel_for returns when all are done.
class Team {
 public:
  Team(int nworkers, int device) {
    for (int w = 0; w < nworkers; w++)
      th_.emplace_back([this, w, device] {
        (void)hipSetDevice(device);
        uint64_t seen = 0;
        for (;;) {
          std::unique_lock<std::mutex> lk(mu_);
          cv_work_.wait(lk, [&] { return stop_ || (gen_ != seen && next_ < total_) ; });
          if (stop_) return;
          const uint64_t g = gen_;
          while (next_ < total_ && gen_ == g) {
            const size_t i = next_++;
            lk.unlock();
            (*fn_)(i, w);
            lk.lock();
            if (++done_ == total_) cv_done_.notify_all();
          }
          seen = g;
        }
      });
  }
  ~Team() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_work_.notify_all();
    for (auto& t : th_) t.join();
  }
  void parallel_for(size_t count, const std::function<void(size_t, int)>& fn) {
    if (!count) return;
    std::unique_lock<std::mutex> lk(mu_);
    fn_ = &fn; next_ = 0; done_ = 0; total_ = count; gen_++;
    cv_work_.notify_all();
    cv_done_.wait(lk, [&] { return done_ == total_; });
    total_ = 0;
  }

 private:
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_work_, cv_done_;
  const std::function<void(size_t, int)>* fn_ = nullptr;
  size_t next_ = 0, total_ = 0, done_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

static inline int ensure_resources(ecgpu_ctx* c, bool need_bounce_up, bool need_bounce_dn) {
  if (!c->copy_stream) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPCHK(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    for (int i = 0; i < ecgpu_ctx::PIPE_NSLOT; i++) {
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_kernel[i], hipEventDisableTiming));
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming));
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_down[i], hipEventDisableTiming));
    }
  }
  for (int dir = 0; dir < 2; dir++) {
    if (!(dir ? need_bounce_dn : need_bounce_up)) continue;
    for (int w = 0; w < ecgpu_ctx::PIPE_NWORK; w++) {
      if (c->bounce[dir][w]) continue;
      HIPCHK(c, hipHostMalloc(&c->bounce[dir][w], ecgpu_ctx::PIPE_BOUNCE, hipHostMallocDefault));
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_bounce[dir][w], hipEventDisableTiming));
    }
  }
  return 0;
}

struct Piece { int arg; size_t off, len; };     // byte range of one argument's chunk

// stage slot of argument a in pipeline slot s (ecgpu.hip: stage_reserve)
static inline int stage_index(int slot, int a) { return ecgpu_ctx::PIPE_STAGE0 + slot * ecgpu_ctx::PIPE_MAXARGS + a; }

template <class Reserve, class Launch>
static int run(ecgpu_ctx* c, const Arg* args, int nargs, const std::vector<size_t>& sizes, bool secret, Reserve reserve, Launch launch) {
  constexpr int NSLOT = ecgpu_ctx::PIPE_NSLOT, NWORK = ecgpu_ctx::PIPE_NWORK;
  constexpr size_t BOUNCE = ecgpu_ctx::PIPE_BOUNCE;
  if (nargs > ecgpu_ctx::PIPE_MAXARGS) return ecgpu_set_err(c, ECGPU_ERR_ARG, "host pipeline: too many arguments");
  const size_t nchunks = sizes.size();
  if (!nchunks) return 0;
  size_t maxchunk = 0;
  for (size_t s : sizes) maxchunk = s > maxchunk ? s : maxchunk;
  bool pinned[ecgpu_ctx::PIPE_MAXARGS], any_up = false, any_dn = false;
  for (int a = 0; a < nargs; a++) {
    pinned[a] = is_pinned(args[a].in ? args[a].in : args[a].out);
    if (!pinned[a] && args[a].in) any_up = true;
    if (!pinned[a] && args[a].out) any_dn = true;
  }
  int rc0 = ensure_resources(c, any_up, any_dn);
  if (rc0) return rc0;
  const int nslot = nchunks < (size_t)NSLOT ? (int)nchunks : NSLOT;
  for (int s = 0; s < nslot; s++)
    for (int a = 0; a < nargs; a++) {
      if (!args[a].in && !args[a].out) continue;
      int rc = reserve(stage_index(s, a), maxchunk * args[a].unit);
      if (rc) return rc;
    }
  std::vector<size_t> start(nchunks + 1, 0);
  for (size_t i = 0; i < nchunks; i++) start[i + 1] = start[i] + sizes[i];

  std::mutex mu;
  std::condition_variable cv;
  size_t up_issued = 0, launched = 0, down_issued = 0;     // chunks whose uploads / kernels / downloads have been ISSUED
  bool abort_flag = false;
  hipError_t first_err = hipSuccess;
  const char* err_where = "";
  auto fail = [&](hipError_t e, const char* where) {
    std::lock_guard<std::mutex> lk(mu);
    if (first_err == hipSuccess) { first_err = e; err_where = where; }
    abort_flag = true;
    cv.notify_all();
  };
  auto pieces_of = [&](size_t ci, bool inputs) {
    std::vector<Piece> v;
    for (int a = 0; a < nargs; a++) {
      if (pinned[a] || !(inputs ? (const void*)args[a].in : (const void*)args[a].out)) continue;
      const size_t bytes = sizes[ci] * args[a].unit;
      for (size_t o = 0; o < bytes; o += BOUNCE) v.push_back({a, o, bytes - o < BOUNCE ? bytes - o : BOUNCE});
    }
    return v;
  };

  // ---- uploads: own thread (a pageable source is copied through the bounce pool by its helpers)
  std::thread uploader([&] {
    (void)hipSetDevice(c->device);
    std::unique_ptr<Team> team;
    if (any_up) team.reset(new Team(NWORK, c->device));
    for (size_t ci = 0; ci < nchunks; ci++) {
      const int slot = (int)(ci % NSLOT);
      if (ci >= (size_t)NSLOT) {                // the slot's inputs are free once the kernels of chunk ci - NSLOT have run
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return launched + NSLOT > ci || abort_flag; });
        if (abort_flag) return;
        lk.unlock();
        hipError_t e = hipStreamWaitEvent(c->up_stream, c->ev_kernel[slot], 0);
        if (e != hipSuccess) return fail(e, "upload: wait for the slot");
      } else {
        std::lock_guard<std::mutex> lk(mu);
        if (abort_flag) return;
      }
      const size_t lo = start[ci], cnt = sizes[ci];
      for (int a = 0; a < nargs; a++)
        if (args[a].in && pinned[a]) {
          hipError_t e = hipMemcpyAsync(c->stage[stage_index(slot, a)], (const char*)args[a].in + lo * args[a].unit, cnt * args[a].unit, hipMemcpyHostToDevice, c->up_stream);
          if (e != hipSuccess) return fail(e, "upload");
        }
      if (any_up) {
        const std::vector<Piece> pieces = pieces_of(ci, true);
        team->parallel_for(pieces.size(), [&](size_t pi, int w) {
          const Piece& p = pieces[pi];
          hipError_t e = hipEventSynchronize(c->ev_bounce[0][w]);          // the previous piece of this buffer has left it
          if (e == hipSuccess) {
            memcpy(c->bounce[0][w], (const char*)args[p.arg].in + lo * args[p.arg].unit + p.off, p.len);
            e = hipMemcpyAsync((char*)c->stage[stage_index(slot, p.arg)] + p.off, c->bounce[0][w], p.len, hipMemcpyHostToDevice, c->up_stream);
          }
          if (e == hipSuccess) e = hipEventRecord(c->ev_bounce[0][w], c->up_stream);
          if (e != hipSuccess) fail(e, "upload through the bounce pool");
        });
      }
      hipError_t e = hipEventRecord(c->ev_up[slot], c->up_stream);
      if (e != hipSuccess) return fail(e, "upload: record");
      std::lock_guard<std::mutex> lk(mu);
      up_issued = ci + 1;
      cv.notify_all();
    }
  });

  // ---- downloads: own thread
  std::thread downloader([&] {
    (void)hipSetDevice(c->device);
    std::unique_ptr<Team> team;
    if (any_dn) team.reset(new Team(NWORK, c->device));
    for (size_t ci = 0; ci < nchunks; ci++) {
      const int slot = (int)(ci % NSLOT);
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return launched > ci || abort_flag; });
        if (abort_flag) return;
      }
      hipError_t e = hipStreamWaitEvent(c->copy_stream, c->ev_kernel[slot], 0);
      if (e != hipSuccess) return fail(e, "download: wait for the kernels");
      const size_t lo = start[ci], cnt = sizes[ci];
      for (int a = 0; a < nargs; a++)
        if (args[a].out && pinned[a]) {
          e = hipMemcpyAsync((char*)args[a].out + lo * args[a].unit, c->stage[stage_index(slot, a)], cnt * args[a].unit, hipMemcpyDeviceToHost, c->copy_stream);
          if (e != hipSuccess) return fail(e, "download");
        }
      if (any_dn) {
        const std::vector<Piece> pieces = pieces_of(ci, false);
        team->parallel_for(pieces.size(), [&](size_t pi, int w) {
          const Piece& p = pieces[pi];
          hipError_t e2 = hipMemcpyAsync(c->bounce[1][w], (const char*)c->stage[stage_index(slot, p.arg)] + p.off, p.len, hipMemcpyDeviceToHost, c->copy_stream);
          if (e2 == hipSuccess) e2 = hipEventRecord(c->ev_bounce[1][w], c->copy_stream);
          if (e2 == hipSuccess) e2 = hipEventSynchronize(c->ev_bounce[1][w]);
          if (e2 == hipSuccess) memcpy((char*)args[p.arg].out + lo * args[p.arg].unit + p.off, c->bounce[1][w], p.len);
          else fail(e2, "download through the bounce pool");
        });
      }
      e = hipEventRecord(c->ev_down[slot], c->copy_stream);
      if (e != hipSuccess) return fail(e, "download: record");
      std::lock_guard<std::mutex> lk(mu);
      down_issued = ci + 1;
      cv.notify_all();
    }
  });

  // ---- the calling thread launches
  int rc = 0;
  for (size_t ci = 0; ci < nchunks; ci++) {
    const int slot = (int)(ci % NSLOT);
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return (up_issued > ci && down_issued + NSLOT > ci) || abort_flag; });
      if (abort_flag) break;
    }
    hipError_t e = hipStreamWaitEvent(c->stream, c->ev_up[slot], 0);
    if (e == hipSuccess && ci >= (size_t)NSLOT) e = hipStreamWaitEvent(c->stream, c->ev_down[slot], 0);   // the slot's outputs have been read
    if (e != hipSuccess) { fail(e, "launch: wait"); break; }
    void* dev[ecgpu_ctx::PIPE_MAXARGS];
    for (int a = 0; a < nargs; a++) dev[a] = (args[a].in || args[a].out) ? c->stage[stage_index(slot, a)] : nullptr;
    rc = launch(dev, sizes[ci], ci);
    if (rc == 0) {
      e = hipEventRecord(c->ev_kernel[slot], c->stream);
      if (e != hipSuccess) { fail(e, "launch: record"); break; }
    }
    std::lock_guard<std::mutex> lk(mu);
    if (rc != 0) abort_flag = true; else launched = ci + 1;
    cv.notify_all();
    if (rc != 0) break;
  }
  uploader.join();
  downloader.join();
  (void)hipStreamSynchronize(c->up_stream);          // an aborted run may still have copies in flight
  (void)hipStreamSynchronize(c->stream);
  hipError_t es = hipStreamSynchronize(c->copy_stream);
  if (secret)                                          // bounce buffers saw the secrets (scalars up, shared values down)
    for (int dir = 0; dir < 2; dir++)
      for (int w = 0; w < NWORK; w++)
        if (c->bounce[dir][w]) memset(c->bounce[dir][w], 0, BOUNCE);
  if (rc) return rc;
  if (first_err != hipSuccess) return ecgpu_set_err(c, ECGPU_ERR_RUNTIME, "host pipeline (%s): %s", err_where, hipGetErrorString(first_err));
  if (es != hipSuccess) return ecgpu_set_err(c, ECGPU_ERR_RUNTIME, "host pipeline (final synchronisation): %s", hipGetErrorString(es));
  return 0;
}

}  // namespace hostpipe
