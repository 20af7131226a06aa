// Host runtime shared by the C-ABI translation units: context, model handle, HIP helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mrk.h"
#include "forest.hpp"

namespace mrk {

struct StatusError : std::runtime_error {
  int status;
  StatusError(int s, const std::string &m) : std::runtime_error(m), status(s) {}
};

void set_last_error(const std::string &msg);

#define MRK_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      throw ::mrk::StatusError(MRK_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// hipFree / hipHostFree wait for EVERY resident kernel of the device.  While gangs of the serving queue are resident (capi_rank.cpp:
// they stay up to MRK_SERVE_LIFE_US and are relaunched at once under traffic, so with 8 of them the device is hardly ever idle) a
// buffer that regrows - the scratch batches of mrk_rank's front finding their sizes - stalled its caller for 70 ... 190 ms
// (profiles/r06_al_front_trace.txt: the `build` phase of a combined batch).  So while any gang is resident, memory is not given
// back at once: release_device / release_pinned put it on a list that is emptied when the last gang has been waited for.
std::atomic<int> &resident_gangs();           // capi.cpp
void release_device(void *p);                 // hipFree now, or later
void release_pinned(void *p);                 // hipHostFree now, or later
void flush_deferred_releases();               // (no gang resident: give everything back)

// RAII device buffer (hipMalloc / hipFree), grow-only reserve.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) release_device(p);
    p = nullptr;
    cap = 0;
  }
  // The first allocation is exact; a buffer that has to GROW is one whose size follows the traffic (the scratch batches of
  // mrk_rank's front: the combined batch is whatever was queued), and every regrowth is a hipFree - a device-wide
  // synchronisation - plus a hipMalloc: milliseconds, seen as 40-90 ms p99 by 64 callers while eight lanes found their sizes
  // (profiles/r06_c).  So regrowth at least doubles (x 1.25 beyond 256 MB).
  void reserve(size_t bytes) {
    if (bytes <= cap) return;
    const size_t grown = cap >= (256u << 20) ? cap + cap / 4 : cap * 2;
    release();
    size_t want = bytes < 256 ? 256 : bytes;
    if (grown > want) want = grown;
    MRK_HIP(hipMalloc(&p, want));
    cap = want;
  }
  template <typename T>
  T *as() const { return (T *)p; }
};

// pinned host staging buffer
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  PinBuf() = default;
  PinBuf(const PinBuf &) = delete;
  PinBuf &operator=(const PinBuf &) = delete;
  ~PinBuf() { if (p) release_pinned(p); }
  void reserve(size_t bytes) {  // (regrowth doubles, like DevBuf's)
    if (bytes <= cap) return;
    const size_t grown = cap >= (256u << 20) ? cap + cap / 4 : cap * 2;
    if (p) release_pinned(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes < 4096 ? 4096 : bytes;
    if (grown > want) want = grown;
    MRK_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
  }
  template <typename T>
  T *as() const { return (T *)p; }
};

struct KernelTimer {
  double total_ms = 0.0;
  int64_t launches = 0;
};

struct Store;     // store.hpp
struct Registry;  // features.hpp

// Experiment switches (DESIGN.md "Experiment switches"): environment variables read ONCE, when the library is first
// used - never on a launch path.  Tests and the measurement scripts that flip a switch inside one process call the
// exported mrk_debug_reload_switches() afterwards.
struct Switches {
  bool rank_fused = true;      // MRK_RANK_FUSED=0: pre-pass and assembly as separate kernels, tables in an HBM arena
  bool rank_cells = true;      // MRK_RANK_CELLS=0: keep the f64 matrix between assembly and scoring
  bool scorer_walk = false;    // MRK_SCORER=walk: the tree-walk scorer even where the bit-vector scorer applies
  int fused_threads = 0;       // MRK_FUSED_THREADS: item lanes of the fused kernel's workgroups (0: by request size)
  bool prepass_lds = true;     // MRK_PREPASS_LDS=0: the stand-alone pre-pass kernel builds its hash tables in the HBM arena
  int fused_slices = 0;        // MRK_FUSED_SLICES=n: workgroups per request of the fused kernel (0: by batch shape; 1: off)
  int fused_split = 0;         // MRK_FUSED_SPLIT=1|2|4: op split of the fused kernel's workgroups (0: 4 / 2 for batches of <= 16 requests)
  bool rank_combine = true;    // MRK_RANK_COMBINE=0: no batching front in mrk_rank
  bool rank_serve = true;      // MRK_RANK_SERVE=0: mrk_serve_rank never takes the persistent-workgroup queue (everything through mrk_rank)
  int serve_life_us = 20000;   // MRK_SERVE_LIFE_US: ... and leaves after the request it is serving once it is this old, idle or not (bounds what a hipFree on another thread waits for)
  int serve_spin_callers = 8;  // MRK_SERVE_SPIN_CALLERS: up to this many callers of the serving queue wait for their answer spinning; the ones beyond
                               // sleep through most of the device's time first (a host with a CPU quota throttles 64 spinning threads)
  int serve_sleep_extra_us = 8;   // MRK_SERVE_SLEEP_EXTRA_US: ... that sleep = the device's smoothed time per request + this
  int serve_poll_us = 4;          // MRK_SERVE_POLL_US: ... and after it looks for its answer this often instead of spinning (0: spin)
  int serve_overload_ms = 200;    // MRK_SERVE_OVERLOAD_MS: a queue with more slots than the process has CPUs sends everybody through the front for this long
                                  // once callers overflow its slots in numbers (0: never)
  int serve_idle_us = 2000;    // MRK_SERVE_IDLE_US: a serving workgroup without a request for this long leaves its CU (relaunched by the next request)
  bool rank_fused_score = false; // MRK_RANK_FUSED_SCORE=1: full batches of small requests in ONE launch (assembly, forest, ordering per request workgroup) - measured slower than the three launches, kept for A/B
  bool rank_one = true;        // MRK_RANK_ONE=0: mrk_rank's small batches take the three-launch path instead of the one-launch kernel
  int combine_max = 256;       // MRK_RANK_COMBINE_MAX
  int rank_lanes = 3;          // MRK_RANK_LANES: batches of mrk_rank's front in flight at once (1 ... 8)
  int table_load_pct = 75;     // MRK_TABLE_LOAD_PCT
  int host_threads = 0;        // MRK_HOST_THREADS (0: min(8, hardware threads))
  int jit_mode = 4;            // MRK_RANK_JIT: 0 off, 1 on (wait for the compiler), 2 require, 3 async, 4 auto (default: disk cache at once, else async)
  int jit_waves = 0;           // MRK_JIT_WAVES
  int fused_lds_min = 0;       // MRK_FUSED_LDS_MIN: least dynamic LDS the fused assembly kernel asks for (experiments: fewer resident workgroups)
  std::string jit_defines;     // MRK_JIT_DEFINES: macros prepended to the specialised kernels' source (experiments)
  bool thr_stage = true;       // MRK_THR_STAGE=0: the assembly kernels search threshold tables in global memory instead of staging them in LDS (experiments)
  bool jit_shipped = true;     // MRK_JIT_SHIPPED=0: ignore the code objects shipped next to the library (tests of the compile paths)
  bool items_lds = true;       // MRK_ITEMS_LDS=0: the item-parallel assembly kernel probes the pre-pass tables in the HBM arena even where a workgroup's request's tables fit its LDS
  // Round 6 (r06_x, native closed-loop callers of mrk_rank, same box): the one-launch kernel for combined batches of up to 128
  // requests (was 16: bigger ones took three launches + a copy) and op-split workgroups for up to 64: 188 k -> 228 k requests/s at
  // 64 callers, 315 k -> 339 k at 128, 351 k -> 397 k at 256 (the last measured with 128 / 32).
  int rank_one_max = 128;      // MRK_RANK_ONE_MAX: most requests of a batch the one-launch kernel takes (mrk_rank's combined batches)
  int split_max_req = 64;      // MRK_SPLIT_MAX_REQ: batches of up to this many small requests get op-split workgroups (launch_shape.hpp)
  bool items_rt = true;        // MRK_ITEMS_RT=0: the item-parallel kernel stages threshold tables per wavefront and column even where all of them fit in LDS (A/B of the resident-table kernel)
  int items_rt_threads = 0;    // MRK_ITEMS_RT_THREADS=256|512: lanes of the resident-table kernel's workgroups (default: by launch size)
  bool jit_sig = true;         // MRK_JIT_SIG=0: the specialised kernels are keyed by the program only and read the forest's column descriptors from memory (A/B of the view-signature folding)
  bool jit_record_regs = true; // MRK_JIT_REGS=0: the specialised kernel reads the candidate's record cell by cell instead of keeping it in registers
  std::string jit_cache_dir;   // MRK_JIT_CACHE_DIR, else $XDG_CACHE_HOME/mrk_jit, else ~/.cache/mrk_jit; "" / "off": none
  int big_sort_cap = 4096;     // MRK_BIG_SORT_CAP: pairs a bucket of the multi-workgroup sort orders in LDS (smaller: tests reach the global-memory path)
  bool jit_prepass = true;     // MRK_JIT_PREPASS=0: the stand-alone pre-pass (config 4) interprets the program even when the assembly is specialised
  int big_sort_bucket = 0;     // MRK_BIG_SORT_BUCKET: target pairs per bucket of the multi-workgroup sort (0: 1 024)
  int big_sort_tile = 0;       // MRK_BIG_SORT_TILE: candidates per workgroup of its classify / scatter passes (0: n / 512, at least 1 024)
  int qs_split = -1;           // MRK_QS_SPLIT
  int qs_kernel = 1;           // MRK_QS_KERNEL
  int qs_r = 2;                // MRK_QS_R
  int walk_tile = 0;           // MRK_WALK_TILE=256: the tree-walk scorer's rows per workgroup (default: 512 where the tile fits)
  bool encoder_graph = false;  // MRK_ENCODER_GRAPH
  int encoder_skinny = 15;     // MRK_ENCODER_SKINNY
  bool encoder_packed = true;  // MRK_ENCODER_PACKED=0: padded batches for pooled / logit calls too
  bool encoder_f32_mfma = true;  // MRK_ENCODER_F32_MFMA=0: the f32 products / attention on the vector unit (the test instrument)
};
const Switches &switches();
void reload_switches();


}  // namespace mrk

struct mrk_ctx {
  // owner reference (dropped by mrk_shutdown) + one per live model / batch: the context outlives its handles
  std::atomic<int> refs{1};
  bool closed = false;
  int device = 0;
  hipStream_t stream = nullptr;
  // the stream kernel launches and their HIP-event timers go to: the context stream, or - while a batch with
  // its own stream runs (ctx->mu held) - that batch's stream
  hipStream_t launch = nullptr;
  int n_cus = 0;
  size_t lds_per_block = 0;
  std::mutex mu;  // serialises kernel launches (ctx->launch routing, timers) + the context's scratch buffers
  // The feature store: puts / flushes take it exclusively, everything that reads the host mirror or launches kernels
  // that read the device tables takes it shared (a flush may reallocate the tables; it waits for the device first).
  // Lock order: store_mu before mu.
  std::shared_mutex store_mu;
  // libstdc++'s shared_mutex prefers readers; since round 6 several leaders of mrk_rank's front hold the store shared at
  // overlapping times, so a put could wait for ever.  Writers announce themselves (StoreWriteLock) and new readers
  // (StoreAccess, capi_rank.cpp) let them pass first.
  std::atomic<int> store_writers{0};
  // scratch for predict_f64
  mrk::DevBuf d_x, d_out, d_flag;
  mrk::DevBuf d_cells;  // binned tile of the bit-vector scorer (score_qs.hip), grow-only
  mrk::PinBuf h_flag;
  // profiling
  bool profile = false;
  std::map<std::string, mrk::KernelTimer> timers;
  std::vector<std::tuple<std::string, hipEvent_t, hipEvent_t>> pending_events;
  // feature side (created by mrk_config_load_json)
  mrk::Registry *registry = nullptr;  // owned; freed by mrk::free_rank_state
  mrk::Store *store = nullptr;
  // mrk_rank's scratch batches ("lanes": grow-only mrk_batch objects, lane 0 on the context stream, the others on streams of
  // their own; owned, freed by mrk::free_rank_state).  A lane is held by one leader of the batching front at a time (qmu).
  static constexpr int RANK_LANES_MAX = 8;
  void *rank_lane[RANK_LANES_MAX] = {};
  bool lane_busy[RANK_LANES_MAX] = {};
  // multi-GPU (comm.cpp): the RCCL communicator this context's device belongs to (ncclComm_t), nullptr = a world of one
  void *comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  mrk::DevBuf d_comm;                 // scratch of the host-value collectives
  // EVERY collective on `comm` is issued under this lock (RCCL does not allow concurrent calls on one communicator).  The
  // host must also issue collectives in the same order on every rank (mrk.h): the lock makes a mistake a wait, not a race.
  std::mutex comm_mu;
  std::mutex servers_mu;              // the serving queues of this context (capi_rank.cpp mrk_serve_*): a store flush stops their workgroups
  std::vector<void *> servers;        // mrk_server*
  std::atomic<int> n_servers{0};      // its size, for mrk_rank's lock-free "is there a queue at all"
  std::atomic<void *> hot_server{nullptr};   // the most recently started one: mrk_rank's way in without the mutex ...
  std::atomic<uint32_t> hot_epoch{0};        // ... guarded by reader counts per epoch parity (capi_rank.cpp rank_through_server)
  std::atomic<int> hot_readers[2] = {{0}, {0}};
  std::mutex hot_stop_mu;
  // batching front of mrk_rank: concurrent callers are combined into device batches by whichever waiting caller finds a
  // free lane (capi_rank.cpp); several batches are in flight at once - one per lane
  std::mutex qmu;
  std::condition_variable qcv;
  std::vector<void *> rank_queue;     // RankTicket*
  mrk_ctx();
  ~mrk_ctx();
};

// exclusive access to a context's feature store, with precedence over readers that have not started yet
struct StoreWriteLock {
  mrk_ctx *ctx;
  std::unique_lock<std::shared_mutex> lk;
  explicit StoreWriteLock(mrk_ctx *c) : ctx(c) {
    ctx->store_writers.fetch_add(1);
    lk = std::unique_lock<std::shared_mutex>(ctx->store_mu);
    ctx->store_writers.fetch_sub(1);
  }
};

struct mrk_model {
  mrk_ctx *ctx = nullptr;
  std::atomic<int> refs{1};
  mrk::Forest forest;
  mrk::PackedForest packed;
  std::vector<std::string> container_features;
  int n_warmup = 0;                    // container v3: RankingEventFormat records kept for mrk_model_warmup
  std::vector<uint8_t> warmup_bytes;
  mrk::DevBuf d_image, d_trees, d_chunks, d_cat;
  // bit-vector image (forests of <= 16-leaf trees); qs.ok == false => tree-walk kernel only
  mrk::PackedForestQS qs;
  mrk::QsSignature qs_sig;             // the image's view signature (forest.hpp): part of the key of the specialised assembly kernels
  mrk::DevBuf d_qs_nodes, d_qs_leaves, d_qs_thr, d_qs_feats, d_qs_views, d_qs_catnodes, d_qs_cat;
};

namespace mrk {

// Times a kernel with HIP events on ctx->stream when profiling is enabled.
struct ScopedKernelTimer {
  mrk_ctx *ctx;
  const char *name;
  hipEvent_t a = nullptr, b = nullptr;
  ScopedKernelTimer(mrk_ctx *c, const char *n);
  ~ScopedKernelTimer();
};
void drain_profile_events(mrk_ctx *ctx);

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a function ON A DEVICE: a process that drives several devices
// (mrk_init with n contexts) has to opt in once per (device, function), whichever thread launches first.  Cheap on the
// launch path: a thread-local memo of the last pair in front of a mutex-protected set.
void lds_optin(mrk_ctx *ctx, const void *fn, int bytes = 160 * 1024);

// Hardware queues the HIP runtime may spread this process's streams over (GPU_MAX_HW_QUEUES; ROCm's default is 4).  The first
// mrk_* call that touches HIP sets it to 24 unless the host has set it; returns what is in force as far as the library can tell.
int hw_queue_budget();

void ctx_retain(mrk_ctx *ctx);
void ctx_release(mrk_ctx *ctx);

// capi_rank.cpp: releases ctx->registry / ctx->store
void free_rank_state(mrk_ctx *ctx);
// comm.cpp
void comm_destroy(mrk_ctx *ctx);
void comm_allgather_f64_inplace(mrk_ctx *ctx, double *buf, size_t chunk, hipStream_t stream);
void comm_allgather_f64(mrk_ctx *ctx, const double *send, double *recv, size_t count, hipStream_t stream);
void comm_allgather_i32(mrk_ctx *ctx, const int32_t *send, int32_t *recv, size_t count, hipStream_t stream);
// features.cpp: drops the encoder references mrk_config_bind_encoder took
void unbind_encoders(mrk_ctx *ctx);

// score.hip
void launch_score(mrk_ctx *ctx, mrk_model *m, const double *d_x, int rows, int cols, double *d_out,
                  int *d_flag);
uint32_t score_chunk_budget();
// score_qs.hip: false => not applicable, use the tree-walk kernel
bool launch_score_qs(mrk_ctx *ctx, mrk_model *m, const double *d_x, int rows, int cols, double *d_out, int *d_flag,
                     const uint32_t *d_row_req);
void launch_score_qs_cells(mrk_ctx *ctx, mrk_model *m, const uint16_t *d_cells, int rows, double *d_out);

}  // namespace mrk
