// Host runtime shared by the C-ABI translation units: context, model handle, HIP helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
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
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  void reserve(size_t bytes) {
    if (bytes <= cap) return;
    release();
    size_t want = bytes < 256 ? 256 : bytes;
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
  ~PinBuf() { if (p) (void)hipHostFree(p); }
  void reserve(size_t bytes) {
    if (bytes <= cap) return;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    size_t want = bytes < 4096 ? 4096 : bytes;
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
  std::mutex mu;  // serialises stream use + scratch buffers (one in-flight call per ctx)
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
  void *rank_scratch = nullptr;       // mrk_batch reused by mrk_rank (owned; freed by mrk::free_rank_state)
  // batching front of mrk_rank: concurrent callers are combined into one device batch by whichever caller
  // finds no leader active (capi_rank.cpp)
  std::mutex qmu;
  std::condition_variable qcv;
  std::vector<void *> rank_queue;     // RankTicket*
  bool rank_leader = false;
  mrk_ctx();
  ~mrk_ctx();
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

void ctx_retain(mrk_ctx *ctx);
void ctx_release(mrk_ctx *ctx);

// capi_rank.cpp: releases ctx->registry / ctx->store
void free_rank_state(mrk_ctx *ctx);
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
