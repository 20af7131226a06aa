// Feature store + rank entry points of the C ABI (include/mrk.h).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <thread>

#include <linux/futex.h>
#include <sched.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <cstdlib>

#include "features.hpp"
#include "jit.hpp"
#include "launch_shape.hpp"
#include "qs_device.hpp"
#include "rank.hpp"
#include "runtime.hpp"

namespace mrk {

void launch_prepass(mrk_ctx *ctx, const StoreDev &st, const ProgramDev &prog, const BatchDev &b, uint32_t max_req_entries, void *jit_fn);
void launch_assemble(mrk_ctx *ctx, const StoreDev &st, const ProgramDev &prog, const BatchDev &b);
void launch_sort(mrk_ctx *ctx, const BatchDev &b, int max_items);
void launch_status_or(hipStream_t stream, const int32_t *all, int world, int n_req, int32_t *status);
void launch_normalize(mrk_ctx *ctx, const BatchDev &b, int dim, int col, int mode);
void launch_normalize_big(mrk_ctx *ctx, const BatchDev &b, int dim, int col, int item_begin, int n, int *order, void *scratch);
struct SortSrc { const double *vals; const unsigned long long *raw; long long stride; int negate; };  // sort_device.hpp
void launch_big_sort(mrk_ctx *ctx, hipStream_t stream, const SortSrc &src, int n, int *out_order, void *scratch);  // bigsort.hip
size_t big_sort_scratch_bytes(int n);
void launch_score_batch(mrk_ctx *ctx, mrk_model *m, const double *d_x, int rows, int cols, double *d_out,
                        int *d_status, const uint32_t *d_row_req);
void launch_assemble_cells(mrk_ctx *ctx, const StoreDev &st, const ProgramDev &prog, const BatchDev &b, const QsDev &q,
                           uint16_t *cells, bool f64, void *jit_fn, uint32_t max_req_entries, void *jit_rt_fn = nullptr, uint32_t thr_total = 0);
void launch_rank_fused(mrk_ctx *ctx, const StoreDev &st, const ProgramDev &prog, const BatchDev &b, uint32_t tab_entries,
                       int vals_cap, int threads, int op_split, int slices, const QsDev *q, uint16_t *cells, bool f64, void *jit_fn, size_t rt_bytes = 0);
size_t fused_lds_bytes(uint32_t tab_entries, int vals_cap, int threads, uint32_t thr_cap, size_t rt_bytes, bool split);
size_t fused_rt_max_bytes();
int fused_max_prep();
QsDev qs_device_view(const mrk_model *m);
QsForestDev qs_forest_view(const mrk_model *m);  // score_qs.hip
size_t rank_one_lds_bytes(uint32_t tab_entries, int vals_cap, int threads, uint32_t thr_cap, int n_views, bool f64, size_t rt_bytes);
size_t rank_fused_score_lds_bytes(uint32_t tab_entries, int vals_cap, int threads, uint32_t thr_cap, int n_views, bool f64, size_t rt_bytes);
void launch_rank_fused_score(mrk_ctx *ctx, const StoreDev &st, const ProgramDev &prog, const BatchDev &b, uint32_t tab_entries, int vals_cap,
                             int threads, const QsDev &q, const QsForestDev &f, uint16_t *cells, bool f64, void *jit_fn);
void launch_rank_serve(mrk_ctx *ctx, hipStream_t stream, const StoreDev &st, const ProgramDev &prog, const QsDev &q, const QsForestDev &f, const ServeGangDev &gang,
                       int n_slots, int threads, size_t lds, bool f64, void *jit_fn);
void launch_rank_one(mrk_ctx *ctx, const StoreDev &st, const ProgramDev &prog, const BatchDev &b, uint32_t tab_entries, int vals_cap,
                     int threads, int op_split, const QsDev &q, const QsForestDev &f, const OneOut &out, bool f64, void *jit_fn);
int load_feature_values(Store &store, const uint8_t *bytes, size_t len, int64_t now_ms);  // codec.cpp
void launch_resolve_ids(hipStream_t stream, const IdTableDev &tab, const uint8_t *d_bytes, const uint32_t *d_offs, uint32_t bytes_len, const ReqDev *d_reqs,
                        int n_req, int total, int32_t *d_item_slot, uint32_t *d_item_req, int32_t *d_load_status);  // resolve.hip

template <typename F>
static int guard(F &&f) {
  try {
    f();
    return MRK_OK;
  } catch (const StatusError &e) {
    set_last_error(e.what());
    return e.status;
  } catch (const std::bad_alloc &) {
    set_last_error("out of host memory");
    return MRK_ERR_DEVICE;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return MRK_ERR_PARSE;
  }
}

static Store &store_of(mrk_ctx *ctx) {
  if (!ctx) throw StatusError(MRK_ERR_INVALID_ARG, "null context");
  if (!ctx->store) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_config_load_json must be called first");
  return *ctx->store;
}

int status_to_code(int st, std::string &msg) {
  if (st & ST_ARITHMETIC) { msg = "java.lang.ArithmeticException: / by zero (normalised rate: global `top` counter is 0)"; return MRK_ERR_ARITHMETIC; }
  if (st & ST_DIM) { msg = "dim mismatch: item embedding is shorter than the query embedding"; return MRK_ERR_DIM_MISMATCH; }
  if (st & ST_ILLEGAL_ARG) { msg = "requirement failed: Duration is limited to +-(2^63-1)ns (ca. 292 years)"; return MRK_ERR_INVALID_ARG; }
  if (st & 32) { msg = "Input data contains `inf` or a value too large, while `missing` is not set to `inf`"; return MRK_ERR_INVALID_ARG; }
  if (st & ST_BAD_IDS) { msg = "item id offsets descend or pass bytes_len (mrk_item_ids)"; return MRK_ERR_INVALID_ARG; }
  if (st & ST_TOO_MANY) { msg = "diversity over more values than the device pre-pass supports: set `top`"; return MRK_ERR_UNSUPPORTED; }
  if (st & ST_TABLE_FULL) { msg = "internal: pre-pass hash table under-sized (store changed between prepare and run?)"; return MRK_ERR_DEVICE; }
  return MRK_OK;
}

}  // namespace mrk

using namespace mrk;

struct mrk_batch {
  mrk_ctx *ctx = nullptr;
  const Program *prog = nullptr;
  int n_req = 0, total_items = 0;
  std::mutex bmu;            // a batch is driven by one thread at a time; this makes a mistake a wait instead of a race
  DevBuf d_in, d_prep_out, d_arena, d_matrix;
  hipStream_t stream = nullptr;   // batches made by mrk_batch_prepare / _create own a stream: several can be in flight on one device
  hipStream_t s() const { return stream ? stream : ctx->stream; }
  DevBuf d_out;              // [scores: (T + shard padding) f64][order: T i32][status: n_req i32][load status: n_req i32], fetched with ONE copy
                             // (load status: what the id-resolution kernel found at load time - ST_BAD_IDS; status is zeroed by every run)
  size_t out_order_off = 0, out_status_off = 0, out_bytes = 0;
  PinBuf h_out;
  bool fetch_enqueued = false;          // the download of d_out into h_out is on the stream behind the last run
  bool direct_out = false;              // the last run was the one-launch kernel: scores / order / status were written into h_out by the device
  DevBuf d_cells;            // the scorer's binned tile (bit-vector models), grow-only
  DevBuf d_sort;                        // scratch of the multi-workgroup sort (bigsort.hip)
  DevBuf d_norm_order;                  // norm: position over a request of more than SORT_MAX_ITEMS candidates: its column's order
  DevBuf d_gather;                      // mrk_batch_gather_scores: the scores of every rank
  DevBuf d_gather_status;               // item-sharded runs: the status words of every rank ([world][n_req]) before they are OR-ed
  struct BigReq { int r, n_items, item_begin; };
  std::vector<BigReq> big;              // requests with n_items > SORT_MAX_ITEMS
  PinBuf h_in;
  HostBatch hb;              // host half of the last load (grow-only scratch)
  DevBuf d_ids;              // flat item ids of the batch: [offsets: (T + 1) u32][bytes]
  PinBuf h_ids;              // their staging copy when the caller's memory is not pinned
  BatchDev view{};
  std::vector<int32_t> h_status;
  std::vector<int32_t> codes;           // mrk_status per request (mrk_batch_host_outputs)
  bool ran = false;
  // one-workgroup-per-request path (tables in LDS)
  bool fused_ok = false;
  uint32_t fused_entries = 0;
  int fused_vals = 1, fused_threads = 64;
  int fused_slices = 1;      // workgroups per request of the fused kernel (rank_device.hpp rank_fused_body)
  int fused_split = 1;       // op split of the fused kernel's workgroups (1 | 2 | 4; rank_device.hpp op_owner)
  // the f64 matrix is materialised only on demand (explain / parity / models without a bit-vector image)
  bool want_matrix = false;
  bool matrix_valid = false;
};

namespace mrk {

void free_rank_state(mrk_ctx *ctx) {
  for (int i = 0; i < mrk_ctx::RANK_LANES_MAX; ++i) {
    mrk_batch *b = (mrk_batch *)ctx->rank_lane[i];
    if (!b) continue;
    (void)hipSetDevice(ctx->device);
    if (b->stream) {
      (void)hipStreamSynchronize(b->stream);
      (void)hipStreamDestroy(b->stream);
    }
    delete b;
    ctx->rank_lane[i] = nullptr;
  }
  delete ctx->registry;
  delete ctx->store;
  ctx->registry = nullptr;
  ctx->store = nullptr;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void quiesce_servers(mrk_ctx *ctx);

// Access to the feature store for the duration of a call.  Readers (resolving a batch, launching kernels that gather from
// the device tables) share it; whatever is dirty is flushed first, exclusively.  `exclusive`: the call mutates host-side
// store state itself (lazily tokenised cross-encoder texts).
struct StoreAccess {
  mrk_ctx *ctx;
  std::shared_lock<std::shared_mutex> shared;
  std::unique_lock<std::shared_mutex> unique;
  StoreAccess(mrk_ctx *c, bool exclusive = false, bool flush = true) : ctx(c) {
    if (!ctx->store) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_config_load_json must be called first");
    if (exclusive) {
      ctx->store_writers.fetch_add(1);
      unique = std::unique_lock<std::shared_mutex>(ctx->store_mu);
      ctx->store_writers.fetch_sub(1);
      if (flush) do_flush();
      return;
    }
    lock_shared();
    while (flush && ctx->store->dirty()) {
      shared.unlock();
      {
        StoreWriteLock x(ctx);
        do_flush();
      }
      lock_shared();
    }
  }
  // readers that have not started wait for announced writers (puts from the feedback stream must not starve behind the
  // overlapping leaders of mrk_rank's front); a reader never holds anything while it waits here
  void lock_shared() {
    if (ctx->store_writers.load(std::memory_order_acquire) > 0) {
      // ... for at most a millisecond: a stream of puts that never pauses must not starve the readers either
      const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(1);
      while (ctx->store_writers.load(std::memory_order_acquire) > 0 && std::chrono::steady_clock::now() < until) std::this_thread::yield();
    }
    shared = std::shared_lock<std::shared_mutex>(ctx->store_mu);
  }
  // the device work is enqueued: what follows (waiting for it, copying results) needs no store - a flush synchronises the device itself
  void release() {
    if (shared.owns_lock()) shared.unlock();
    if (unique.owns_lock()) unique.unlock();
  }
  void do_flush() {
    MRK_HIP(hipSetDevice(ctx->device));
    quiesce_servers(ctx);  // persistent workgroups hold the device views they were launched with: they leave, and come back with the next request
    ctx->store->flush(ctx->stream);
  }
};

static bool program_mutates_store(const Program &prog) {
  for (const HostOp &ho : prog.host_ops)
    if (ho.def->cross && ho.def->encoder) return true;
  return false;
}

static bool is_pinned_host(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // plain malloc memory: not an error worth keeping
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

// (re)builds the device-resident batch (inputs uploaded, outputs allocated, id resolution enqueued); the caller holds
// the store (StoreAccess) and the batch
static void build_batch(mrk_ctx *ctx, const Program &prog, const mrk_request *reqs, int n_req, const mrk_item_ids *ids, mrk_batch &b) {
  Store &store = *ctx->store;
  MRK_HIP(hipSetDevice(ctx->device));
  HostBatch &hb = b.hb;
  resolve_requests(prog, store, reqs, n_req, ids, hb);
  b.ctx = ctx;
  b.prog = &prog;
  b.n_req = n_req;
  b.total_items = hb.total_items;
  const int T = hb.total_items;
  // one staging blob for all inputs
  size_t off = 0;
  auto place = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t o_reqs = place(hb.reqs.size() * sizeof(ReqDev));
  const size_t o_consts = place(hb.consts.size() * 8);
  const size_t o_irf = place(hb.irf.size() * 4);
  const size_t o_ov = place(hb.overrides.size() * sizeof(Override));
  const size_t o_prep = place(hb.prep_out.size() * sizeof(PrepOut));
  const size_t host_bytes = std::max<size_t>(off, 256);  // everything the host fills; the per-item arrays follow
  const size_t o_slot = place((size_t)T * 4);
  const size_t o_ireq = place((size_t)T * 4);
  const size_t total_bytes = std::max<size_t>(off, 256);
  const size_t up_bytes = hb.device_ids ? host_bytes : total_bytes;  // device-resolved ids: item_slot / item_req are written by a kernel
  b.h_in.reserve(up_bytes);
  b.d_in.reserve(total_bytes);
  uint8_t *h = b.h_in.as<uint8_t>();
  auto put = [&](size_t o, const void *src, size_t bytes) { if (bytes) memcpy(h + o, src, bytes); };
  put(o_reqs, hb.reqs.data(), hb.reqs.size() * sizeof(ReqDev));
  put(o_consts, hb.consts.data(), hb.consts.size() * 8);
  put(o_irf, hb.irf.data(), hb.irf.size() * 4);
  put(o_ov, hb.overrides.data(), hb.overrides.size() * sizeof(Override));
  put(o_prep, hb.prep_out.data(), hb.prep_out.size() * sizeof(PrepOut));
  if (!hb.device_ids) {
    put(o_slot, hb.item_slot.data(), (size_t)T * 4);
    put(o_ireq, hb.item_req.data(), (size_t)T * 4);
  }
  MRK_HIP(hipMemcpyAsync(b.d_in.p, h, up_bytes, hipMemcpyHostToDevice, b.s()));
  uint8_t *d = b.d_in.as<uint8_t>();
  // outputs in one allocation: one device-to-host copy per fetch (single-request latency)
  const size_t score_slots = (size_t)T + 256 * QS_TILE_ROWS;  // room for the padded chunks of an item-sharded all-gather
  b.out_order_off = align_up(score_slots * 8, 256);
  b.out_status_off = align_up(b.out_order_off + std::max<size_t>(T, 1) * 4, 256);
  b.out_bytes = b.out_status_off + 2 * std::max<size_t>(n_req, 1) * 4;
  b.d_out.reserve(b.out_bytes);
  int32_t *d_load_status = (int32_t *)(b.d_out.as<uint8_t>() + b.out_status_off) + std::max(n_req, 1);
  MRK_HIP(hipMemsetAsync(d_load_status, 0, std::max<size_t>(n_req, 1) * 4, b.s()));
  if (hb.device_ids && T > 0) {
    // the ids as they arrived: [offsets][bytes] in one device buffer; pinned caller memory is read by the copy engine
    // directly, anything else goes through the batch's pinned staging buffer
    const size_t off_bytes = ((size_t)T + 1) * 4;
    const size_t id_bytes = ids->offsets[T];
    if (id_bytes > ids->bytes_len || ids->bytes_len > 0xffffffffull)
      throw StatusError(MRK_ERR_INVALID_ARG, "mrk_item_ids: the last offset passes bytes_len (or bytes_len >= 4 GiB)");
    const size_t o_bytes = align_up(off_bytes, 256);
    b.d_ids.reserve(o_bytes + std::max<size_t>(id_bytes, 1));
    const void *src_off = ids->offsets, *src_bytes = ids->bytes;
    const bool own_staging = (const void *)ids->offsets == (const void *)b.h_ids.p;   // mrk_rank's front flattens straight into this batch's pinned staging
    if (!own_staging && (!is_pinned_host(ids->offsets) || !is_pinned_host(ids->bytes))) {
      b.h_ids.reserve(o_bytes + std::max<size_t>(id_bytes, 1));
      memcpy(b.h_ids.p, ids->offsets, off_bytes);
      memcpy(b.h_ids.as<uint8_t>() + o_bytes, ids->bytes, id_bytes);
      src_off = b.h_ids.p;
      src_bytes = b.h_ids.as<uint8_t>() + o_bytes;
    }
    MRK_HIP(hipMemcpyAsync(b.d_ids.p, src_off, off_bytes, hipMemcpyHostToDevice, b.s()));
    if (id_bytes) MRK_HIP(hipMemcpyAsync(b.d_ids.as<uint8_t>() + o_bytes, src_bytes, id_bytes, hipMemcpyHostToDevice, b.s()));
    launch_resolve_ids(b.s(), store.tables[SC_ITEM].id_table_view(), b.d_ids.as<uint8_t>() + o_bytes, b.d_ids.as<uint32_t>(), (uint32_t)id_bytes,
                       (const ReqDev *)(d + o_reqs), n_req, T, (int32_t *)(d + o_slot), (uint32_t *)(d + o_ireq), d_load_status);
  }
  b.d_arena.reserve(std::max<size_t>(hb.arena_entries, 1) * 8);
  b.d_matrix.reserve(std::max<size_t>((size_t)T * prog.dim, 1) * 8);
  BatchDev &v = b.view;
  v.reqs = (const ReqDev *)(d + o_reqs);
  v.n_req = n_req;
  v.total_items = T;
  v.item_lo = 0;
  v.item_hi = T;
  v.item_slot = (const int32_t *)(d + o_slot);
  v.item_req = (const uint32_t *)(d + o_ireq);
  v.consts = (const double *)(d + o_consts);
  v.irf = (const int32_t *)(d + o_irf);
  v.overrides = (const Override *)(d + o_ov);
  v.n_overrides = (int)hb.overrides.size();
  v.prep_out = (PrepOut *)(d + o_prep);
  v.arena = (unsigned long long *)b.d_arena.p;
  v.status = (int32_t *)(b.d_out.as<uint8_t>() + b.out_status_off);
  v.matrix = (double *)b.d_matrix.p;
  v.scores = (double *)b.d_out.p;
  v.order = (int32_t *)(b.d_out.as<uint8_t>() + b.out_order_off);
  b.h_status.assign(n_req, 0);
  b.ran = false;
  b.matrix_valid = false;
  b.fetch_enqueued = false;
  b.big.clear();
  size_t big_bytes = 0;
  for (int r = 0; r < n_req; ++r)
    if (hb.reqs[r].n_items > SORT_MAX_ITEMS) {
      b.big.push_back({r, hb.reqs[r].n_items, (int)hb.reqs[r].item_begin});
      big_bytes = std::max(big_bytes, big_sort_scratch_bytes(hb.reqs[r].n_items));
    }
  if (big_bytes) b.d_sort.reserve(big_bytes);  // one scratch: the big requests of a batch are sorted one after the other on its stream
  // small requests: both phases in one workgroup, tables in LDS (<= 64 KB keeps two workgroups per CU)
  uint32_t vals = 1;
  while ((int)vals < hb.max_doubles) vals <<= 1;
  b.fused_entries = (uint32_t)std::max<uint64_t>(hb.max_req_entries, 1);
  b.fused_vals = (int)vals;
  const Switches &sw = switches();
  const FusedShape shape = fused_launch_shape(n_req, hb.max_items, sw.fused_threads, sw.fused_split, sw.fused_slices, sw.split_max_req);  // launch_shape.hpp
  b.fused_split = shape.split;
  b.fused_slices = shape.slices;
  b.fused_threads = shape.threads();
  b.fused_ok = sw.rank_fused && (int)prog.prep.size() <= fused_max_prep() && hb.max_items <= 1024 &&
               hb.max_req_entries <= (1u << 20) && fused_lds_bytes(b.fused_entries, b.fused_vals, b.fused_threads, QS_LDS_THR, fused_rt_max_bytes(), true) <= 64 * 1024;
}

static void check_model_fits(mrk_model *model, const Program &prog) {
  if (!model) return;
  if (model->refs.load() <= 0) throw StatusError(MRK_ERR_INVALID_ARG, "model is closed");
  int used = 0;
  for (auto &t : model->forest.trees)
    for (auto f : t.feat) used = std::max(used, f + 1);
  if (used > prog.dim)
    throw StatusError(MRK_ERR_DIM_MISMATCH, "booster splits on feature " + std::to_string(used - 1) + " but model '" + prog.model +
                                                "' has " + std::to_string(prog.dim) + " columns");
}

// pre-pass + assembly into the row-major f64 matrix (ClickthroughQuery's layout); inside a LaunchOn
static void assemble_matrix(mrk_batch &b, const StoreDev &st, const ProgramDev &pd, void *jit_matrix_fn = nullptr) {
  mrk_ctx *ctx = b.ctx;
  if (b.fused_ok) {
    launch_rank_fused(ctx, st, pd, b.view, b.fused_entries, b.fused_vals, b.fused_threads, b.fused_split, jit_matrix_fn ? 1 : b.fused_slices, nullptr, nullptr, true, jit_matrix_fn);
  } else {
    launch_prepass(ctx, st, pd, b.view, b.fused_entries, nullptr);
    launch_assemble(ctx, st, pd, b.view);
  }
  for (const Program::NormCol &nc : b.prog->norm_cols)  // schema.norm.scale over the request's column (Normalize.scala:13-45)
    if (!nc.cross || nc.cross->encoder) {
      launch_normalize(ctx, b.view, pd.dim, nc.col, nc.mode);
      if (nc.mode == NORM_POSITION)  // requests too large for one workgroup: ordered by the multi-workgroup sort
        for (auto &br : b.big) {
          b.d_norm_order.reserve((size_t)br.n_items * 4);
          launch_normalize_big(ctx, b.view, pd.dim, nc.col, br.item_begin, br.n_items, b.d_norm_order.as<int>(), b.d_sort.p);
        }
    }
  b.matrix_valid = true;
}

// items per shard of an item-sharded run: ceil(total / count) rounded up to whole scorer tiles
static_assert(MRK_SHARD_TILE == QS_TILE_ROWS, "a shard is a whole number of scorer tiles");
static int shard_chunk(const mrk_batch &b, int count) { return (int)mrk_shard_chunk(b.total_items, count); }

static void sort_batch(mrk_batch &b) {
  mrk_ctx *ctx = b.ctx;
  launch_sort(ctx, b.view, b.hb.max_items);
  if (b.big.empty()) return;
  ScopedKernelTimer timer(ctx, "sort");
  for (auto &br : b.big) {
    const SortSrc src{b.view.scores + br.item_begin, nullptr, 1, 1};
    launch_big_sort(ctx, ctx->launch, src, br.n_items, b.view.order + br.item_begin, b.d_sort.p);
  }
}

// routes kernel launches (and their timers) to a batch's stream; holds ctx->mu for as long as it lives
struct LaunchOn {
  mrk_ctx *ctx;
  std::lock_guard<std::mutex> lk;
  LaunchOn(mrk_ctx *c, hipStream_t s) : ctx(c), lk(c->mu) { ctx->launch = s; }
  ~LaunchOn() { ctx->launch = ctx->stream; }
};

// A handful of small requests (mrk_rank and its batching front): everything in ONE launch, results straight into the
// batch's pinned buffer (rank_device.hpp rank_one_body).  Only for callers that read the results through h_out.
static bool rank_one_applies(const mrk_batch &b, const mrk_model *model, bool cells) {
  const Switches &sw = switches();
  if (!sw.rank_one || !cells || !b.fused_ok || b.n_req < 1 || b.n_req > sw.rank_one_max || b.hb.max_items > QS_TILE_ROWS || b.view.n_overrides > 0) return false;
  if (b.fused_slices != 1 || b.fused_threads > 512) return false;
  const QsDev q = qs_device_view(model);
  return rank_one_lds_bytes(b.fused_entries, b.fused_vals, b.fused_threads, q.thr_cap, q.n_views, model->forest.backend == Backend::LightGBM, (size_t)q.rt_doubles * 8) <= 96 * 1024;
}

// enqueue the pipeline on the batch's stream for batch items [lo, hi); the caller holds the store (StoreAccess)
static void run_batch(mrk_batch &b, mrk_model *model, int lo, int hi, bool sort, bool direct = false) {
  mrk_ctx *ctx = b.ctx;
  MRK_HIP(hipSetDevice(ctx->device));
  check_model_fits(model, *b.prog);
  const Switches &sw = switches();
  const int rows = hi - lo;
  // MRK_RANK_CELLS=0 / MRK_SCORER=walk keep the f64 matrix between assembly and scoring (A/B measurements)
  // a column normalised across the request (bi- / cross-encoder `norm`) needs every raw value of the request before
  // any of them can be binned: such models go through the f64 matrix
  // A SHARD of such a model assembles and normalises the whole batch - every rank holds the whole store, and the
  // normalised column (one cosine per candidate) is the cheap part of that model - and scores its own slice: the forest
  // is what the ranks share out.  (Exchanging the raw column instead would put a collective between assembly and scoring.)
  const bool normalises = b.prog->normalises();
  const bool whole_matrix = normalises && (lo != 0 || hi != b.total_items);
  const bool cells = sw.rank_cells && !sw.scorer_walk && model && model->qs.ok && !b.want_matrix && rows > 0 && !normalises;
  const bool f64 = model && model->forest.backend == Backend::LightGBM;
  // the kernel specialised for this model (hiprtc, ~7 s the first time): compiled before the launch lock is taken
  // (the specialised matrix kernel has no op-split form: a split batch of a matrix-scored model runs the interpreting kernel)
  const bool one = direct && sort && lo == 0 && hi == b.total_items && rank_one_applies(b, model, cells);
  // full batches of small requests: assembly + forest + ordering in the request's workgroup (one launch, phases of different
  // kinds side by side on every CU)
  const bool fused_score = !one && sort && lo == 0 && hi == b.total_items && cells && sw.rank_fused_score && b.fused_ok && b.fused_split == 1 &&
                           b.fused_slices == 1 && b.hb.max_items <= QS_TILE_ROWS && b.view.n_overrides == 0 && b.big.empty() && b.fused_threads <= 256 &&
                           rank_fused_score_lds_bytes(b.fused_entries, b.fused_vals, b.fused_threads, qs_device_view(model).thr_cap, qs_device_view(model).n_views, f64, (size_t)qs_device_view(model).rt_doubles * 8) <= 64 * 1024;
  // (the kernels that write the scorer's tile are keyed by the forest's view signature too: their sinks hold it as constants)
  const QsSignature *sig = cells && sw.thr_stage ? &model->qs_sig : nullptr;
  void *jit_fn = !cells ? (b.fused_ok && model && b.fused_split == 1 && b.fused_slices == 1 ? jit_matrix_function(*b.prog) : nullptr)  // a model scored from the f64 matrix: the hot path too
                        : one ? jit_one_function(*b.prog, f64, sig)
                        : fused_score ? jit_fused_score_function(*b.prog, f64, sig)
                        : !b.fused_ok ? jit_items_function(*b.prog, f64, sig)
                        : b.fused_split > 1 || b.fused_slices > 1 ? jit_split_function(*b.prog, f64, sig) : jit_rank_function(*b.prog, f64, sig);
  void *jit_rt_fn = cells && !one && !fused_score && !b.fused_ok ? jit_items_rt_function(*b.prog, f64, sig) : nullptr;   // the resident-table form of the item-parallel kernel
  void *jit_prep_fn = cells && !one && !fused_score && !b.fused_ok && b.prog->prep.size() && sw.jit_prepass ? jit_prepass_function(*b.prog) : nullptr;   // (before LaunchOn: a first use may compile)
  LaunchOn on(ctx, b.s());
  const StoreDev st = ctx->store->device_view();
  const ProgramDev pd = b.prog->device_view();
  b.fetch_enqueued = false;
  b.direct_out = false;
  if (one) {
    b.h_out.reserve(b.out_bytes);
    uint8_t *h = b.h_out.as<uint8_t>();
    const OneOut out{(double *)h, (int32_t *)(h + b.out_order_off), (int32_t *)(h + b.out_status_off),
                     (const int32_t *)(b.d_out.as<uint8_t>() + b.out_status_off) + std::max(b.n_req, 1), std::max(b.n_req, 1)};
    b.view.item_lo = 0;
    b.view.item_hi = b.total_items;
    launch_rank_one(ctx, st, pd, b.view, b.fused_entries, b.fused_vals, b.fused_threads, b.fused_split, qs_device_view(model), qs_forest_view(model), out, f64, jit_fn);
    b.matrix_valid = false;
    b.direct_out = true;
    b.ran = true;
    return;
  }
  MRK_HIP(hipMemsetAsync(b.view.status, 0, std::max<size_t>(b.n_req, 1) * 4, b.s()));
  b.view.item_lo = whole_matrix ? 0 : lo;
  b.view.item_hi = whole_matrix ? b.total_items : hi;
  if (fused_score) {
    const QsDev q = qs_device_view(model);
    b.d_cells.reserve(std::max<size_t>((size_t)b.n_req * q.n_views * QS_TILE_ROWS * 2, 16));   // one tile per request
    launch_rank_fused_score(ctx, st, pd, b.view, b.fused_entries, b.fused_vals, b.fused_threads, q, qs_forest_view(model), b.d_cells.as<uint16_t>(), f64, jit_fn);
    b.matrix_valid = false;
    b.ran = true;
    return;
  }
  if (cells) {
    // hot path: the assembled values go straight into the scorer's binned tile; no f64 matrix
    const QsDev q = qs_device_view(model);
    const size_t tile_bytes = (size_t)q.n_views * QS_TILE_ROWS * 2;
    const size_t n_tiles = ((size_t)b.total_items + QS_TILE_ROWS - 1) / QS_TILE_ROWS;
    b.d_cells.reserve(std::max<size_t>(n_tiles * tile_bytes, 16));
    if (hi % QS_TILE_ROWS && tile_bytes)  // rows past the last item of the last tile of this range
      MRK_HIP(hipMemsetAsync(b.d_cells.as<uint8_t>() + (size_t)(hi / QS_TILE_ROWS) * tile_bytes, 0, tile_bytes, b.s()));
    if (b.fused_ok) {
      launch_rank_fused(ctx, st, pd, b.view, b.fused_entries, b.fused_vals, b.fused_threads, b.fused_split, b.fused_slices, &q, b.d_cells.as<uint16_t>(), f64, jit_fn,
                        sig && sig->ok ? (size_t)sig->rt_total * 8 : 0);
    } else {
      launch_prepass(ctx, st, pd, b.view, b.fused_entries, jit_prep_fn);
      launch_assemble_cells(ctx, st, pd, b.view, q, b.d_cells.as<uint16_t>(), f64, jit_fn, b.fused_entries, jit_rt_fn, jit_rt_fn ? model->qs_sig.rt_total : 0u);
    }
    b.matrix_valid = false;
    // lo is a multiple of the tile size: the scorer sees rows [lo, hi) as its rows [0, hi - lo)
    launch_score_qs_cells(ctx, model, b.d_cells.as<uint16_t>() + (size_t)(lo / QS_TILE_ROWS) * (tile_bytes / 2), rows, b.view.scores + lo);
  } else {
    assemble_matrix(b, st, pd, jit_fn);
    b.matrix_valid = whole_matrix || (lo == 0 && hi == b.total_items);
    if (model && rows > 0) {
      launch_score_batch(ctx, model, b.view.matrix + (size_t)lo * pd.dim, rows, pd.dim, b.view.scores + lo, b.view.status, b.view.item_req + lo);
    } else if (rows > 0) {
      // NoopModel (ml/rank/NoopRanker.scala:22-26): every score is 0.0
      MRK_HIP(hipMemsetAsync(b.view.scores + lo, 0, (size_t)rows * 8, b.s()));
    }
  }
  b.view.item_lo = 0;
  b.view.item_hi = b.total_items;
  if (sort) sort_batch(b);
  b.ran = true;
}

static void run_batch(mrk_batch &b, mrk_model *model) { run_batch(b, model, 0, b.total_items, true); }

// scores + order + status -> the batch's pinned result buffer, one copy (a copy into pageable caller memory is staged by
// the runtime anyway, and three small copies cost three round trips); asynchronous
static void enqueue_fetch(mrk_batch &b, bool scores, bool order) {
  if (b.direct_out) return;  // the device wrote h_out itself
  const size_t T = (size_t)b.total_items;
  b.h_out.reserve(b.out_bytes);
  uint8_t *h = b.h_out.as<uint8_t>();
  const uint8_t *d = b.d_out.as<uint8_t>();
  if (T * 8 <= 64 * 1024) {  // small batch: the whole blob
    MRK_HIP(hipMemcpyAsync(h, d, b.out_bytes, hipMemcpyDeviceToHost, b.s()));
  } else {                   // else the used ranges
    if (scores && T) MRK_HIP(hipMemcpyAsync(h, d, T * 8, hipMemcpyDeviceToHost, b.s()));
    if (order && T) MRK_HIP(hipMemcpyAsync(h + b.out_order_off, d + b.out_order_off, T * 4, hipMemcpyDeviceToHost, b.s()));
    MRK_HIP(hipMemcpyAsync(h + b.out_status_off, d + b.out_status_off, 2 * std::max<size_t>(b.n_req, 1) * 4, hipMemcpyDeviceToHost, b.s()));  // status + load status
  }
}

// the caller holds the store when `matrix` may have to be assembled (StoreAccess), and the batch
static void fetch_batch(mrk_batch &b, double *scores, int32_t *order, double *matrix) {
  mrk_ctx *ctx = b.ctx;
  MRK_HIP(hipSetDevice(ctx->device));
  const size_t T = (size_t)b.total_items;
  if (matrix && T && b.prog->dim && !b.matrix_valid) {
    // the last run assembled straight into the scorer's tile: materialise the f64 matrix now
    LaunchOn on(ctx, b.s());
    assemble_matrix(b, ctx->store->device_view(), b.prog->device_view());
  }
  if (!b.fetch_enqueued) enqueue_fetch(b, scores != nullptr, order != nullptr);
  b.fetch_enqueued = false;
  if (matrix && T && b.prog->dim) MRK_HIP(hipMemcpyAsync(matrix, b.d_matrix.p, T * b.prog->dim * 8, hipMemcpyDeviceToHost, b.s()));
  // (Measured and removed, round 6: a leader of mrk_rank's front sleeping on a hipEventBlockingSync event instead of this polling
  // wait when several lanes are in flight - 183 k vs 183 k requests/s at 64 callers, 289 k vs 301 k at 128, profiles/r06_e_callers.txt.)
  MRK_HIP(hipStreamSynchronize(b.s()));
  const uint8_t *h = b.h_out.as<uint8_t>();
  if (scores && T) memcpy(scores, h, T * 8);
  if (order && T) memcpy(order, h + b.out_order_off, T * 4);
  const int32_t *hs = (const int32_t *)(h + b.out_status_off), *hl = hs + std::max(b.n_req, 1);
  for (int r = 0; r < b.n_req; ++r) b.h_status[(size_t)r] = hs[r] | hl[r];
  if (ctx->profile) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    drain_profile_events(ctx);
  }
}

}  // namespace mrk

extern "C" {

int mrk_config_load_json(mrk_ctx *ctx, const char *json, size_t len) {
  return guard([&] {
    if (!ctx || !json) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
    StoreWriteLock sl(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->registry) throw StatusError(MRK_ERR_INVALID_ARG, "this context already has a configuration");
    MRK_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<Store> st(new Store());
    std::unique_ptr<Registry> reg = load_config(json, len, *st);
    ctx->store = st.release();
    ctx->registry = reg.release();
  });
}

int mrk_config_specialize(const char *json, size_t len, const char *model_name, int f64, int what, uint8_t *out, size_t cap, size_t *needed) {
  return guard([&] {
    const int kernel = (what >> 8) - 1;  // what = (1 + kernel) << 8 | form: one kernel's translation unit; high byte 0: all kernels
    what &= 0xff;
    if (!json || !model_name || !needed || (what != 0 && what != 1) || kernel < JIT_ALL || kernel >= JIT_KERNELS)
      throw StatusError(MRK_ERR_INVALID_ARG, "null argument / unknown `what`");
    Store st;
    std::unique_ptr<Registry> reg = load_config(json, len, st, /*upload=*/false);
    const Program *p = reg->program(model_name);
    if (!p) throw StatusError(MRK_ERR_NOT_FOUND, std::string("model ") + model_name + " is not configured");
    const std::string src = jit_source(*p, f64 != 0, kernel);
    std::vector<char> code;
    if (what == 1 && out) {  // sizing calls (out == NULL) do not compile
      std::string log;
      code = jit_compile(src, log);
    }
    const char *data = what == 0 ? src.data() : code.data();
    const size_t n = what == 0 ? src.size() : code.size();
    *needed = what == 1 && !out ? (size_t)1 << 22 : n;  // code objects: an upper bound for the sizing call
    if (cap < n || !out) throw StatusError(MRK_ERR_INVALID_ARG, "output buffer too small (see *needed)");
    memcpy(out, data, n);
  });
}

static const Program &program_of(mrk_ctx *ctx, const char *model_name);

// host only: the view signature of a serialised booster (what mrk_model_load would key this model's kernels by)
static QsSignature signature_of_bytes(int backend, const uint8_t *bytes, size_t len, bool &f64) {
  if (!bytes || !len) throw StatusError(MRK_ERR_INVALID_ARG, "null model bytes");
  Forest f;
  if (backend == MRK_BACKEND_LIGHTGBM) f = parse_lightgbm_text((const char *)bytes, len);
  else if (backend == MRK_BACKEND_XGBOOST) f = parse_xgboost(bytes, len);
  else throw StatusError(MRK_ERR_INVALID_ARG, "unsupported booster tag " + std::to_string(backend));
  f64 = f.backend == Backend::LightGBM;
  const PackedForestQS qs = pack_forest_qs(f, f.n_features);
  return qs.ok ? qs_signature(qs, qs_stage_cap(qs)) : QsSignature{};
}

int mrk_config_specialize_for_model(const char *json, size_t len, const char *model_name, int backend, const uint8_t *model_bytes, size_t model_len,
                                    int what, uint8_t *out, size_t cap, size_t *needed) {
  return guard([&] {
    const int kernel = (what >> 8) - 1;
    what &= 0xff;
    if (!json || !model_name || !needed || (what != 0 && what != 1) || kernel < JIT_ALL || kernel >= JIT_KERNELS)
      throw StatusError(MRK_ERR_INVALID_ARG, "null argument / unknown `what`");
    bool f64 = true;
    const QsSignature sig = signature_of_bytes(backend, model_bytes, model_len, f64);
    Store st;
    std::unique_ptr<Registry> reg = load_config(json, len, st, /*upload=*/false);
    const Program *p = reg->program(model_name);
    if (!p) throw StatusError(MRK_ERR_NOT_FOUND, std::string("model ") + model_name + " is not configured");
    const std::string src = jit_source(*p, f64, kernel, switches().jit_sig ? &sig : nullptr);
    std::vector<char> code;
    if (what == 1 && out) {
      std::string log;
      code = jit_compile(src, log);
    }
    const char *data = what == 0 ? src.data() : code.data();
    const size_t n = what == 0 ? src.size() : code.size();
    *needed = what == 1 && !out ? (size_t)1 << 22 : n;
    if (cap < n || !out) throw StatusError(MRK_ERR_INVALID_ARG, "output buffer too small (see *needed)");
    memcpy(out, data, n);
  });
}

int mrk_config_precompile_for_model(const char *json, size_t len, const char *model_name, int backend, const uint8_t *model_bytes, size_t model_len,
                                    unsigned kernel_mask, const char *dir, int *out_compiled) {
  return guard([&] {
    if (out_compiled) *out_compiled = 0;
    if (!json || !model_name || !dir) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
    bool f64 = true;
    const QsSignature sig = signature_of_bytes(backend, model_bytes, model_len, f64);
    Store st;
    std::unique_ptr<Registry> reg = load_config(json, len, st, /*upload=*/false);
    const Program *p = reg->program(model_name);
    if (!p) throw StatusError(MRK_ERR_NOT_FOUND, std::string("model ") + model_name + " is not configured");
    const int n = jit_precompile(*p, f64, kernel_mask, dir, &sig);
    if (out_compiled) *out_compiled = n;
  });
}

int mrk_config_precompile(const char *json, size_t len, const char *model_name, int f64, unsigned kernel_mask, const char *dir, int *out_compiled) {
  return guard([&] {
    if (out_compiled) *out_compiled = 0;
    if (!json || !model_name || !dir) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
    Store st;
    std::unique_ptr<Registry> reg = load_config(json, len, st, /*upload=*/false);
    const Program *p = reg->program(model_name);
    if (!p) throw StatusError(MRK_ERR_NOT_FOUND, std::string("model ") + model_name + " is not configured");
    const int n = jit_precompile(*p, f64 != 0, kernel_mask, dir);
    if (out_compiled) *out_compiled = n;
  });
}

int mrk_config_warmup(mrk_ctx *ctx, const char *model_name) {
  return guard([&] {
    if (!ctx || !model_name) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
    const Program *p;
    {
      std::shared_lock<std::shared_mutex> sl(ctx->store_mu);
      p = &program_of(ctx, model_name);
    }
    jit_wait(*p);
  });
}

int mrk_config_kernel_keys(mrk_ctx *ctx, const char *model_name, char *out, size_t cap, size_t *needed) {
  return guard([&] {
    if (!ctx || !model_name || !needed) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
    const Program *p;
    {
      std::shared_lock<std::shared_mutex> sl(ctx->store_mu);
      p = &program_of(ctx, model_name);
    }
    const std::string keys = jit_loaded_keys(*p);
    *needed = keys.size() + 1;
    if (!out || cap < keys.size() + 1) throw StatusError(MRK_ERR_INVALID_ARG, "output buffer too small (see *needed)");
    memcpy(out, keys.c_str(), keys.size() + 1);
  });
}

#ifdef MRK_PHASE_CLOCKS
extern "C" int mrk_debug_phase_clocks(const mrk::Program *prog, unsigned long long *out64);
// measurement builds only (not part of include/mrk.h): read-and-reset the per-phase cycle sums of the specialised kernel
int mrk_debug_phase(mrk_ctx *ctx, const char *model_name, unsigned long long *out64) {
  if (!ctx || !ctx->registry) return -1;
  std::lock_guard<std::mutex> lk(ctx->mu);
  (void)hipDeviceSynchronize();
  return mrk_debug_phase_clocks(ctx->registry->program(model_name), out64);
}
#endif

int mrk_model_dim(mrk_ctx *ctx, const char *model_name) {
  int dim = -1;
  int rc = guard([&] {
    if (!ctx || !model_name) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
    std::shared_lock<std::shared_mutex> lk(ctx->store_mu);
    if (!ctx->registry) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_config_load_json must be called first");
    const Program *p = ctx->registry->program(model_name);
    if (!p) throw StatusError(MRK_ERR_NOT_FOUND, std::string("model ") + model_name + " is not configured");
    dim = p->dim;
  });
  return rc == MRK_OK ? dim : rc;
}

#define STORE_PUT(call)                                   \
  return guard([&] {                                      \
    if (!key) throw StatusError(MRK_ERR_INVALID_ARG, "null key"); \
    Store &st = store_of(ctx);                            \
    StoreWriteLock lk(ctx); \
    (void)st.call;                                        \
  })

int mrk_store_put_double(mrk_ctx *ctx, const char *key, double v) { STORE_PUT(put_double(key, v)); }
int mrk_store_put_bool(mrk_ctx *ctx, const char *key, int v) { STORE_PUT(put_bool(key, v != 0)); }
int mrk_store_put_string(mrk_ctx *ctx, const char *key, const char *v) { STORE_PUT(put_string(key, v)); }
int mrk_store_put_string_list(mrk_ctx *ctx, const char *key, const char *const *v, int n) { STORE_PUT(put_string_list(key, v, n)); }
int mrk_store_put_double_list(mrk_ctx *ctx, const char *key, const double *v, int n) { STORE_PUT(put_double_list(key, v, n)); }
int mrk_store_put_counter(mrk_ctx *ctx, const char *key, int64_t v) { STORE_PUT(put_counter(key, v)); }
int mrk_store_put_periodic(mrk_ctx *ctx, const char *key, const int64_t *v, int n) { STORE_PUT(put_periodic(key, v, n)); }
int mrk_store_put_bounded_list(mrk_ctx *ctx, const char *key, const char *const *v, int n) { STORE_PUT(put_bounded_list(key, v, n)); }
int mrk_store_delete(mrk_ctx *ctx, const char *key) { STORE_PUT(erase(key)); }
int mrk_store_increment_periodic(mrk_ctx *ctx, const char *key, int64_t ts_ms, int64_t inc) { STORE_PUT(increment_periodic(key, ts_ms, inc)); }
static int put_binary(mrk_ctx *ctx, const uint8_t *bytes, size_t len, int64_t now_ms, int *out_records) {
  return guard([&] {
    if (out_records) *out_records = 0;
    if (!bytes && len) throw StatusError(MRK_ERR_INVALID_ARG, "null blob");
    Store &st = store_of(ctx);
    StoreWriteLock lk(ctx);
    const int n = load_feature_values(st, bytes, len, now_ms);
    if (out_records) *out_records = n;
  });
}

int mrk_store_put_binary(mrk_ctx *ctx, const uint8_t *bytes, size_t len, int *out_records) { return put_binary(ctx, bytes, len, -1, out_records); }

int mrk_store_put_binary_at(mrk_ctx *ctx, const uint8_t *bytes, size_t len, int64_t now_ms, int *out_records) {
  if (now_ms < 0) { set_last_error("mrk_store_put_binary_at: negative clock"); return MRK_ERR_INVALID_ARG; }
  return put_binary(ctx, bytes, len, now_ms, out_records);
}

int mrk_store_expire(mrk_ctx *ctx, int64_t now_ms, int64_t *out_expired) {
  return guard([&] {
    if (out_expired) *out_expired = 0;
    Store &st = store_of(ctx);
    StoreWriteLock lk(ctx);
    const int64_t n = st.ttl_expire(now_ms);
    if (out_expired) *out_expired = n;
  });
}

int mrk_store_increment_periodic_batch(mrk_ctx *ctx, const char *const *keys, const int64_t *ts_ms, const int64_t *inc, int n) {
  return guard([&] {
    if (n < 0 || (n > 0 && (!keys || !ts_ms || !inc))) throw StatusError(MRK_ERR_INVALID_ARG, "bad increment batch");
    Store &st = store_of(ctx);
    StoreWriteLock lk(ctx);
    for (int i = 0; i < n; ++i) {
      if (!keys[i]) throw StatusError(MRK_ERR_INVALID_ARG, "null key");
      (void)st.increment_periodic(keys[i], ts_ms[i], inc[i]);
    }
  });
}
int mrk_store_increment(mrk_ctx *ctx, const char *key, int64_t inc) { STORE_PUT(increment(key, inc)); }
int mrk_store_append(mrk_ctx *ctx, const char *key, const char *value, int64_t ts_ms) { STORE_PUT(append(key, value, ts_ms)); }

/* not part of include/mrk.h (measurement aid, Store::clone_items): grows the ITEM table to (copies + 1) x its size */
// launch_shape.hpp through the C ABI, for tests without a device: out = {lanes per workgroup, op split, slices}
int mrk_debug_fused_shape(int n_req, int max_items, int *out3) {
  const mrk::FusedShape s = mrk::fused_launch_shape(n_req, max_items);
  out3[0] = s.threads();
  out3[1] = s.split;
  out3[2] = s.slices;
  return MRK_OK;
}
int mrk_debug_scorer_split(int rows, int views, int f64, int n_cus) {
  return mrk::scorer_waves_per_tile(((long long)rows + mrk::QS_TILE_ROWS - 1) / mrk::QS_TILE_ROWS, views, f64 != 0, n_cus, mrk::QS_LEAVES, mrk::QS_TILE_ROWS);
}

int mrk_debug_clone_items(mrk_ctx *ctx, int copies, int64_t *out_items) {
  return guard([&] {
    Store &st = store_of(ctx);
    StoreWriteLock lk(ctx);
    const uint32_t n = st.clone_items(copies);
    if (out_items) *out_items = n;
  });
}

/* not part of include/mrk.h: layout of one scope's table - out[0..5] = slots, record stride, first byte of the inline heap,
 * heap bytes, token pool entries, f64 pool entries (benchmarks report bytes per record from it) */
int mrk_debug_store_info(mrk_ctx *ctx, int scope, int64_t *out) {
  return guard([&] {
    Store &st = store_of(ctx);
    if (scope < 0 || scope >= SC_COUNT || !out) throw StatusError(MRK_ERR_INVALID_ARG, "bad scope");
    std::shared_lock<std::shared_mutex> lk(ctx->store_mu);
    const Table &t = st.tables[scope];
    out[0] = t.n_slots; out[1] = t.stride; out[2] = t.heap_off; out[3] = t.heap_cap;
    out[4] = (int64_t)st.tok_pool.host.size(); out[5] = (int64_t)st.f64_pool.host.size();
  });
}

int mrk_store_flush(mrk_ctx *ctx) {
  return guard([&] {
    Store &st = store_of(ctx);
    StoreWriteLock lk(ctx);
    MRK_HIP(hipSetDevice(ctx->device));
    st.flush(ctx->stream);
  });
}

static const Program &program_of(mrk_ctx *ctx, const char *model_name) {
  if (!ctx || !model_name) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
  if (!ctx->registry) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_config_load_json must be called first");
  const Program *p = ctx->registry->program(model_name);
  if (!p) throw StatusError(MRK_ERR_NOT_FOUND, std::string("model ") + model_name + " is not configured");
  return *p;
}

// ---- mrk_rank with a batching front.  The reference serves every request on its own thread
// (cats-effect compute pool, one rerank per fiber: api/routes/RankApi.scala:25-41, ml/Ranker.scala:27-83); here concurrent
// callers of mrk_rank are combined.  A context has MRK_RANK_LANES scratch batches ("lanes", default 3), each with its own
// stream, pinned staging and grow-only device buffers.  A caller queues its ticket; whoever is waiting when a lane is free
// takes the lane and every compatible ticket queued so far (same model handle, same model name, same wish for the explain
// matrix; up to MRK_RANK_COMBINE_MAX; FIFO), ranks them as ONE device batch - one upload, the id bytes resolved by a kernel,
// one to three launches, results straight from the lane's pinned buffer into every caller's arrays - and gives the lane
// back.  While that batch is on the device the next leader builds and uploads the next one on another lane: host work and
// device work of consecutive batches overlap (round 5's front had one lane and one leader: 37 k requests/s at 32 callers).
// A single caller sees the old behaviour: a batch of one, lane 0 = the context stream, ids resolved on the host.
// MRK_RANK_COMBINE=0: no combining - every caller ranks alone on whichever lane is free.
namespace {
// A waiting caller sleeps on its OWN ticket's state word (futex): woken by exactly one system call from the leader that
// finished its batch or from whoever freed a lane, and it does not have to take the front's mutex to find out that it is done.
enum : uint32_t { TK_WAITING = 0, TK_LEAD = 1, TK_DONE = 2 };
inline void futex_wait(std::atomic<uint32_t> *w, uint32_t expected) {
  (void)syscall(SYS_futex, (uint32_t *)w, FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}
inline void futex_wake(std::atomic<uint32_t> *w) { (void)syscall(SYS_futex, (uint32_t *)w, FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0); }

struct RankTicket {
  mrk_model *model;
  std::string model_name;
  const mrk_request *req;
  double *scores;
  int32_t *order;
  double *matrix;
  int status = MRK_OK;
  std::string err;
  std::atomic<uint32_t> state{TK_WAITING};   // TK_*: set under ctx->qmu; TK_DONE is the LAST access of anybody else to this (stack) object
  bool taken = false;                        // travels in somebody's batch (ctx->qmu)
  RankTicket(mrk_model *m, const char *name, const mrk_request *r, double *s, int32_t *o, double *mat)
      : model(m), model_name(name), req(r), scores(s), order(o), matrix(mat) {}
};

// measurement aid (MRK_FRONT_TRACE=1; not part of include/mrk.h): per batch of the front [n, lane-wait-free build us, run us, fetch us, total us]
struct FrontTrace { int n; float build_us, run_us, fetch_us, copy_us; };
std::mutex g_trace_mu;
std::vector<FrontTrace> g_trace;
bool g_trace_on = getenv("MRK_FRONT_TRACE") != nullptr;

// ranks tickets [0, n) as one batch on the lane's scratch batch (owned by the calling leader); fills status / err of each.
// The store is held shared (several leaders build at once), the launch lock only while launching.
void rank_tickets(mrk_ctx *ctx, mrk_batch &b, RankTicket **tk, int n) {
  auto fail_all = [&](int code, const std::string &msg) {
    for (int i = 0; i < n; ++i) { tk[i]->status = code; tk[i]->err = msg; }
  };
  try {
    const Program *progp;
    {
      std::shared_lock<std::shared_mutex> sl(ctx->store_mu);
      progp = &program_of(ctx, tk[0]->model_name.c_str());
    }
    const Program &prog = *progp;
    StoreAccess access(ctx, program_mutates_store(prog));
    MRK_HIP(hipSetDevice(ctx->device));
    if (b.ctx) MRK_HIP(hipStreamSynchronize(b.s()));  // a previous call that failed before its fetch may have left an upload from h_in in flight
    std::vector<mrk_request> reqs(n);
    for (int i = 0; i < n; ++i) reqs[i] = *tk[i]->req;
    // Combined batches hand the candidates' ids to the device as they arrived (resolve.hip: one lane per candidate hashes and
    // probes the mirrored id map) instead of hashing them one after the other on the leader's thread: 7 us of host work per
    // 100-candidate request becomes a copy of its id bytes into the lane's pinned staging.  A lone request keeps the host path.
    mrk_item_ids flat{};
    bool use_flat = n > 1;
    size_t total = 0, id_bytes = 0;
    if (use_flat) {
      for (int i = 0; i < n && use_flat; ++i) {
        const mrk_request &rq = reqs[i];
        if (rq.n_items < 0 || (rq.n_items > 0 && !rq.item_ids)) { use_flat = false; break; }
        for (int k = 0; k < rq.n_items; ++k) id_bytes += rq.item_ids[k] ? strlen(rq.item_ids[k]) : 0;
        total += (size_t)rq.n_items;
      }
      if (total == 0 || id_bytes > 0x7fffffffull) use_flat = false;
    }
    if (use_flat) {
      const size_t off_bytes = align_up((total + 1) * 4, 256);
      b.h_ids.reserve(off_bytes + std::max<size_t>(id_bytes, 1));
      uint32_t *offs = b.h_ids.as<uint32_t>();
      uint8_t *bytes = b.h_ids.as<uint8_t>() + off_bytes;
      size_t at = 0, k_all = 0;
      for (int i = 0; i < n; ++i)
        for (int k = 0; k < reqs[i].n_items; ++k) {
          offs[k_all++] = (uint32_t)at;
          if (const char *id = reqs[i].item_ids[k]) {
            const size_t len = strlen(id);
            memcpy(bytes + at, id, len);
            at += len;
          }
        }
      offs[k_all] = (uint32_t)at;
      flat.bytes = bytes;
      flat.offsets = offs;
      flat.bytes_len = at;
    }
    const auto c0 = std::chrono::steady_clock::now();
    build_batch(ctx, prog, reqs.data(), n, use_flat ? &flat : nullptr, b);
    b.want_matrix = tk[0]->matrix != nullptr;
    const auto c1 = std::chrono::steady_clock::now();
    run_batch(b, tk[0]->model, 0, b.total_items, true, /*direct=*/true);
    const auto c2 = std::chrono::steady_clock::now();
    std::vector<double> mat;   // the explain matrix of a combined batch is the one thing still staged (rare path)
    if (b.want_matrix && n > 1) mat.resize((size_t)b.total_items * prog.dim);
    enqueue_fetch(b, true, true);
    b.fetch_enqueued = true;
    access.release();   // (want_matrix was set before the run: the matrix, if asked for, is already assembled)
    fetch_batch(b, nullptr, nullptr, !b.want_matrix ? nullptr : n == 1 ? tk[0]->matrix : mat.data());
    const auto c3 = std::chrono::steady_clock::now();
    // every caller's slice straight out of the lane's pinned result buffer
    const uint8_t *h = b.h_out.as<uint8_t>();
    const double *sc = (const double *)h;
    const int32_t *od = (const int32_t *)(h + b.out_order_off);
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
      const size_t m = (size_t)std::max(tk[i]->req->n_items, 0);
      if (tk[i]->scores && m) memcpy(tk[i]->scores, sc + off, m * 8);
      if (tk[i]->order && m) memcpy(tk[i]->order, od + off, m * 4);
      if (tk[i]->matrix && n > 1 && m) memcpy(tk[i]->matrix, mat.data() + off * prog.dim, m * prog.dim * 8);
      off += m;
    }
    for (int i = 0; i < n; ++i) tk[i]->status = status_to_code(b.h_status[i], tk[i]->err);
    if (g_trace_on) {
      const auto c4 = std::chrono::steady_clock::now();
      auto us = [](auto a, auto b2) { return (float)std::chrono::duration<double, std::micro>(b2 - a).count(); };
      std::lock_guard<std::mutex> tl(g_trace_mu);
      g_trace.push_back({n, us(c0, c1), us(c1, c2), us(c2, c3), us(c3, c4)});
    }
  } catch (const StatusError &e) {
    if (n == 1) fail_all(e.status, e.what());
    else {  // a request the host rejects (bad arguments, dim mismatch) must not fail its neighbours: one by one
      for (int i = 0; i < n; ++i) rank_tickets(ctx, b, tk + i, 1);
    }
  } catch (const std::bad_alloc &) {
    fail_all(MRK_ERR_DEVICE, "out of host memory");
  } catch (const std::exception &e) {
    fail_all(MRK_ERR_PARSE, e.what());
  }
}

// the lane's scratch batch, created on first use (caller holds the lane); lane 0 runs on the context stream
mrk_batch *lane_batch(mrk_ctx *ctx, int lane) {
  if (!ctx->rank_lane[lane]) {
    std::unique_ptr<mrk_batch> nb(new mrk_batch());
    if (lane > 0) {
      MRK_HIP(hipSetDevice(ctx->device));
      MRK_HIP(hipStreamCreateWithFlags(&nb->stream, hipStreamNonBlocking));
    }
    ctx->rank_lane[lane] = nb.release();
  }
  return (mrk_batch *)ctx->rank_lane[lane];
}
}  // namespace

static int rank_through_server(mrk_ctx *ctx, mrk_model *model, const char *model_name, const mrk_request *req, double *out_scores,
                               int32_t *out_order, bool &done);
static int rank_front(mrk_ctx *ctx, mrk_model *model, const char *model_name, const mrk_request *req, double *out_scores,
                      int32_t *out_order, double *out_matrix);

int mrk_rank(mrk_ctx *ctx, mrk_model *model, const char *model_name, const mrk_request *req, double *out_scores,
             int32_t *out_order, double *out_matrix) {
  if (!req || !ctx || !model_name) { set_last_error("null argument"); return MRK_ERR_INVALID_ARG; }
  if (model && model->ctx != ctx) { set_last_error("model belongs to another context"); return MRK_ERR_INVALID_ARG; }
  // A host that started a serving queue for this model (mrk_serve_start: Serve.scala's warm-up) has its per-request calls
  // answered by the queue's resident workgroups - no launch, no copy command -, whatever entry point its fibers use; what the
  // queue does not take (every slot busy, more than 128 candidates, a matrix asked for) goes through the batching front below.
  if (model && !out_matrix && ctx->n_servers.load(std::memory_order_acquire) != 0) {
    bool done = false;
    const int rc = rank_through_server(ctx, model, model_name, req, out_scores, out_order, done);
    if (done || rc != MRK_OK) return rc;
  }
  return rank_front(ctx, model, model_name, req, out_scores, out_order, out_matrix);
}

static int rank_front(mrk_ctx *ctx, mrk_model *model, const char *model_name, const mrk_request *req, double *out_scores,
                      int32_t *out_order, double *out_matrix) {
  RankTicket t(model, model_name, req, out_scores, out_order, out_matrix);
  const Switches &sw = switches();
  const int n_lanes = sw.rank_lanes;
  // Wake-ups are targeted: a finished batch wakes the callers of ITS tickets, a freed lane elects the OLDEST waiting ticket as
  // the next leader - nobody else.  (One condition variable and notify_all - round 6's first version - woke every waiting caller
  // on every batch: at 64 callers x 4 lanes 1.4 M wake-ups a second queued up on this mutex, profiles/r06_d.)
  auto free_lane = [&]() {
    for (int i = 0; i < n_lanes; ++i)
      if (!ctx->lane_busy[i]) return i;
    return -1;
  };
  auto elect = [&]() {  // (qmu held) a lane is free: the oldest waiting ticket leads
    if (ctx->rank_queue.empty() || free_lane() < 0) return;
    RankTicket *q = (RankTicket *)ctx->rank_queue.front();
    if (q->state.exchange(TK_LEAD, std::memory_order_acq_rel) == TK_WAITING) futex_wake(&q->state);
  };
  std::unique_lock<std::mutex> lk(ctx->qmu);
  bool queued = false;
  for (;;) {
    const int lane = t.taken ? -1 : free_lane();
    if (lane < 0) {
      // my ticket travels in somebody else's batch, or every lane is busy: wait for TK_DONE or for my election
      if (!queued && !t.taken) { ctx->rank_queue.push_back(&t); queued = true; }
      {  // an election that found the lane gone again (or the ticket taken meanwhile): back to waiting - unless it is done already
        uint32_t was = TK_LEAD;
        (void)t.state.compare_exchange_strong(was, TK_WAITING, std::memory_order_acq_rel);
      }
      lk.unlock();
      uint32_t st = t.state.load(std::memory_order_acquire);
      // (Measured and removed: up to four waiting callers polling their ticket for a batch's duration before sleeping - no gain
      //  at 4 / 16 callers, -3 % at 64 ... 256 on a box with a 16-CPU quota, profiles/r06_e_callers.txt.)
      while (st == TK_WAITING) {
        futex_wait(&t.state, TK_WAITING);
        st = t.state.load(std::memory_order_acquire);
      }
      if (st == TK_DONE) break;   // (without the mutex: whoever set it does not touch the ticket again)
      lk.lock();
      if (t.state.load(std::memory_order_acquire) == TK_DONE) { lk.unlock(); break; }
      continue;
    }
    // Lead ONE batch on this lane: my ticket plus everything queued that is compatible with it, in arrival order.  A caller
    // never serves other callers' batches after its own result is ready.
    ctx->lane_busy[lane] = true;
    std::vector<RankTicket *> take;
    std::vector<void *> rest;
    take.push_back(&t);
    for (void *p : ctx->rank_queue) {
      RankTicket *q = (RankTicket *)p;
      if (q == &t) continue;
      const bool ok = sw.rank_combine && (int)take.size() < sw.combine_max && q->model == t.model && q->model_name == t.model_name &&
                      (q->matrix != nullptr) == (t.matrix != nullptr);
      if (ok) take.push_back(q); else rest.push_back(p);
    }
    ctx->rank_queue.swap(rest);
    for (RankTicket *q : take) q->taken = true;
    elect();   // tickets this batch does not take (another model, an explain request) need not wait for it: another lane may be free
    lk.unlock();
    try {
      rank_tickets(ctx, *lane_batch(ctx, lane), take.data(), (int)take.size());
    } catch (const std::exception &e) {  // (lane creation: rank_tickets itself reports through the tickets)
      for (RankTicket *q : take) { q->status = MRK_ERR_DEVICE; q->err = e.what(); }
    }
    lk.lock();
    ctx->lane_busy[lane] = false;
    elect();
    lk.unlock();
    // results are in the callers' arrays: release them - outside the mutex, one system call each; TK_DONE is the last access
    // to a ticket (its caller may return, and the object is gone, the moment it is visible)
    for (RankTicket *q : take) {
      if (q == &t) continue;
      std::atomic<uint32_t> *w = &q->state;
      if (w->exchange(TK_DONE, std::memory_order_acq_rel) == TK_WAITING) futex_wake(w);
    }
    break;
  }
  if (t.status != MRK_OK) set_last_error(t.err);
  return t.status;
}

/* not part of include/mrk.h: the front's per-batch phase times since the last call (MRK_FRONT_TRACE=1); out = rows of 5 floats */
int mrk_debug_front_trace(float *out, int cap_rows) {
  std::lock_guard<std::mutex> tl(g_trace_mu);
  const int n = (int)std::min<size_t>(g_trace.size(), (size_t)std::max(cap_rows, 0));
  for (int i = 0; i < n; ++i) {
    out[5 * i] = (float)g_trace[i].n; out[5 * i + 1] = g_trace[i].build_us; out[5 * i + 2] = g_trace[i].run_us;
    out[5 * i + 3] = g_trace[i].fetch_us; out[5 * i + 4] = g_trace[i].copy_us;
  }
  const int total = (int)g_trace.size();
  g_trace.clear();
  return total;
}

int mrk_rank_binary(mrk_ctx *ctx, mrk_model *model, const char *model_name, const uint8_t *event, size_t len,
                    int *out_n_items, double *out_scores, int32_t *out_order, int capacity) {
  DecodedRequest dr;
  int rc = guard([&] {
    if (out_n_items) *out_n_items = 0;
    if (!event) throw StatusError(MRK_ERR_INVALID_ARG, "null event");
    dr.decode(event, len);
    if (out_n_items) *out_n_items = dr.req.n_items;
    if (dr.req.n_items > capacity) throw StatusError(MRK_ERR_INVALID_ARG, "the event has more items than the output buffers hold");
  });
  if (rc != MRK_OK) return rc;
  return mrk_rank(ctx, model, model_name, &dr.req, out_scores, out_order, nullptr);
}

int mrk_model_warmup(mrk_ctx *ctx, mrk_model *model, const char *model_name, int *out_replayed) {
  if (out_replayed) *out_replayed = 0;
  if (!ctx || !model || !model_name) { set_last_error("null argument"); return MRK_ERR_INVALID_ARG; }
  size_t off = 0;
  std::vector<double> sc;
  std::vector<int32_t> od;
  for (int i = 0; i < model->n_warmup; ++i) {
    DecodedRequest dr;
    int rc = guard([&] { off += dr.decode(model->warmup_bytes.data() + off, model->warmup_bytes.size() - off); });
    if (rc != MRK_OK) return rc;
    sc.resize((size_t)std::max(dr.req.n_items, 1));
    od.resize(sc.size());
    rc = mrk_rank(ctx, model, model_name, &dr.req, sc.data(), od.data(), nullptr);
    if (rc != MRK_OK) return rc;  // Serve.maybeWarmup propagates a failing warm-up request
    if (out_replayed) *out_replayed = i + 1;
  }
  return MRK_OK;
}

static const Program &locked_program(mrk_ctx *ctx, const char *model_name) {
  std::shared_lock<std::shared_mutex> sl(ctx->store_mu);
  return program_of(ctx, model_name);
}

int mrk_batch_create(mrk_ctx *ctx, mrk_batch **out) {
  return guard([&] {
    if (!out) throw StatusError(MRK_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!ctx) throw StatusError(MRK_ERR_INVALID_ARG, "null context");
    {
      std::lock_guard<std::mutex> lk(ctx->mu);
      if (ctx->closed) throw StatusError(MRK_ERR_INVALID_ARG, "context is shut down");
    }
    std::unique_ptr<mrk_batch> b(new mrk_batch());
    b->ctx = ctx;
    MRK_HIP(hipSetDevice(ctx->device));
    MRK_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    ctx_retain(ctx);
    *out = b.release();
  });
}

int mrk_batch_load(mrk_batch *batch, const char *model_name, const mrk_request *reqs, int n_req, const mrk_item_ids *ids) {
  return guard([&] {
    if (!batch || !batch->ctx || n_req < 0 || (n_req > 0 && !reqs)) throw StatusError(MRK_ERR_INVALID_ARG, "bad arguments");
    mrk_ctx *ctx = batch->ctx;
    const Program &prog = locked_program(ctx, model_name);
    std::lock_guard<std::mutex> bl(batch->bmu);
    MRK_HIP(hipSetDevice(ctx->device));
    MRK_HIP(hipStreamSynchronize(batch->s()));  // the previous run may still read the staging buffers
    StoreAccess access(ctx, program_mutates_store(prog));
    build_batch(ctx, prog, reqs, n_req, ids, *batch);
  });
}

int mrk_batch_prepare(mrk_ctx *ctx, const char *model_name, const mrk_request *reqs, int n_req, mrk_batch **out) {
  if (!out) { set_last_error("out is null"); return MRK_ERR_INVALID_ARG; }
  *out = nullptr;
  mrk_batch *b = nullptr;
  int rc = mrk_batch_create(ctx, &b);
  if (rc != MRK_OK) return rc;
  rc = mrk_batch_load(b, model_name, reqs, n_req, nullptr);
  if (rc == MRK_OK) rc = guard([&] { MRK_HIP(hipStreamSynchronize(b->s())); });
  if (rc != MRK_OK) {
    const std::string msg = mrk_last_error();
    mrk_batch_free(b);
    set_last_error(msg);
    return rc;
  }
  *out = b;
  return MRK_OK;
}

int mrk_batch_total_items(mrk_batch *batch) { return batch ? batch->total_items : MRK_ERR_INVALID_ARG; }

static void check_batch(mrk_batch *batch, mrk_model *model) {
  if (!batch || !batch->ctx) throw StatusError(MRK_ERR_INVALID_ARG, "null batch");
  if (!batch->prog) throw StatusError(MRK_ERR_INVALID_ARG, "the batch has not been loaded");
  if (model && model->ctx != batch->ctx) throw StatusError(MRK_ERR_INVALID_ARG, "model belongs to another context");
}

int mrk_batch_run(mrk_batch *batch, mrk_model *model) {
  return guard([&] {
    check_batch(batch, model);
    std::lock_guard<std::mutex> bl(batch->bmu);
    StoreAccess access(batch->ctx, false, /*flush=*/false);  // what the batch resolved against stays what it reads
    run_batch(*batch, model);
  });
}

int mrk_batch_shard_chunk(mrk_batch *batch, int shard_count) {
  if (!batch || shard_count < 1) return MRK_ERR_INVALID_ARG;
  return shard_chunk(*batch, shard_count);
}

int mrk_batch_run_shard(mrk_batch *batch, mrk_model *model, int shard_index, int shard_count) {
  return guard([&] {
    check_batch(batch, model);
    if (shard_count < 1 || shard_count > 256 || shard_index < 0 || shard_index >= shard_count)
      throw StatusError(MRK_ERR_INVALID_ARG, "bad shard index / count (1..256 shards)");
    std::lock_guard<std::mutex> bl(batch->bmu);
    StoreAccess access(batch->ctx, false, false);
    int64_t lo = 0, hi = 0;
    if (mrk_shard_range(batch->total_items, shard_index, shard_count, &lo, &hi) != MRK_OK) throw StatusError(MRK_ERR_INVALID_ARG, "bad shard");
    run_batch(*batch, model, (int)lo, (int)hi, false);
  });
}

int mrk_batch_allgather_scores(mrk_batch *batch) {
  return guard([&] {
    check_batch(batch, nullptr);
    std::lock_guard<std::mutex> bl(batch->bmu);
    mrk_ctx *ctx = batch->ctx;
    if (!ctx->comm) return;  // no communicator: a world of one, nothing to merge
    MRK_HIP(hipSetDevice(ctx->device));
    // the score buffer has room for the padded chunks of up to 256 shards (build_batch)
    comm_allgather_f64_inplace(ctx, batch->view.scores, (size_t)shard_chunk(*batch, ctx->comm_world), batch->s());
    // the status words too: an item that fails its request lies in ONE rank's slice, every rank must report it
    if (batch->n_req > 0) {
      batch->d_gather_status.reserve((size_t)ctx->comm_world * batch->n_req * 4);
      comm_allgather_i32(ctx, batch->view.status, batch->d_gather_status.as<int32_t>(), (size_t)batch->n_req, batch->s());
      launch_status_or(batch->s(), batch->d_gather_status.as<int32_t>(), ctx->comm_world, batch->n_req, batch->view.status);
    }
    batch->fetch_enqueued = false;
  });
}

int mrk_batch_run_sharded(mrk_batch *batch, mrk_model *model) {
  if (!batch || !batch->ctx) { set_last_error("null batch"); return MRK_ERR_INVALID_ARG; }
  mrk_ctx *ctx = batch->ctx;
  if (!ctx->comm) return mrk_batch_run(batch, model);
  int rc = mrk_batch_run_shard(batch, model, ctx->comm_rank, ctx->comm_world);
  if (rc == MRK_OK) rc = mrk_batch_allgather_scores(batch);
  if (rc == MRK_OK) rc = mrk_batch_sort(batch);
  return rc;
}

int mrk_batch_gather_scores(mrk_batch *batch, double **d_all_scores) {
  return guard([&] {
    check_batch(batch, nullptr);
    if (!d_all_scores) throw StatusError(MRK_ERR_INVALID_ARG, "null output");
    std::lock_guard<std::mutex> bl(batch->bmu);
    mrk_ctx *ctx = batch->ctx;
    const size_t T = (size_t)batch->total_items;
    MRK_HIP(hipSetDevice(ctx->device));
    if (!ctx->comm) { *d_all_scores = batch->view.scores; return; }
    batch->d_gather.reserve(std::max<size_t>(T * (size_t)ctx->comm_world, 1) * 8);
    comm_allgather_f64(ctx, batch->view.scores, batch->d_gather.as<double>(), T, batch->s());
    *d_all_scores = batch->d_gather.as<double>();
  });
}

int mrk_batch_sort(mrk_batch *batch) {
  return guard([&] {
    check_batch(batch, nullptr);
    std::lock_guard<std::mutex> bl(batch->bmu);
    MRK_HIP(hipSetDevice(batch->ctx->device));
    LaunchOn on(batch->ctx, batch->s());
    batch->fetch_enqueued = false;
    sort_batch(*batch);
  });
}

void *mrk_batch_stream(mrk_batch *batch) { return batch ? (void *)batch->s() : nullptr; }

int mrk_batch_sync(mrk_batch *batch) {
  return guard([&] {
    if (!batch) throw StatusError(MRK_ERR_INVALID_ARG, "null batch");
    MRK_HIP(hipSetDevice(batch->ctx->device));
    MRK_HIP(hipStreamSynchronize(batch->s()));
    if (batch->ctx->profile) {
      std::lock_guard<std::mutex> lk(batch->ctx->mu);
      drain_profile_events(batch->ctx);
    }
  });
}

int mrk_batch_device_outputs(mrk_batch *batch, double **d_scores, int32_t **d_order, double **d_matrix) {
  return guard([&] {
    check_batch(batch, nullptr);
    std::lock_guard<std::mutex> bl(batch->bmu);
    if (d_scores) *d_scores = batch->view.scores;
    if (d_order) *d_order = batch->view.order;
    if (d_matrix) {  // from now on every run materialises the f64 matrix
      batch->want_matrix = true;
      *d_matrix = batch->view.matrix;
    }
  });
}

int mrk_batch_fetch(mrk_batch *batch, double *out_scores, int32_t *out_order, double *out_matrix) {
  return guard([&] {
    check_batch(batch, nullptr);
    std::lock_guard<std::mutex> bl(batch->bmu);
    StoreAccess access(batch->ctx, false, false);
    fetch_batch(*batch, out_scores, out_order, out_matrix);
  });
}

int mrk_batch_enqueue_fetch(mrk_batch *batch) {
  return guard([&] {
    check_batch(batch, nullptr);
    std::lock_guard<std::mutex> bl(batch->bmu);
    MRK_HIP(hipSetDevice(batch->ctx->device));
    enqueue_fetch(*batch, true, true);
    batch->fetch_enqueued = true;
  });
}

int mrk_batch_host_outputs(mrk_batch *batch, const double **scores, const int32_t **order, const int32_t **status) {
  return guard([&] {
    check_batch(batch, nullptr);
    std::lock_guard<std::mutex> bl(batch->bmu);
    MRK_HIP(hipSetDevice(batch->ctx->device));
    if (!batch->fetch_enqueued) {  // mrk_batch_enqueue_fetch was not called behind the run: ask for everything now
      enqueue_fetch(*batch, true, true);
      batch->fetch_enqueued = true;
    }
    fetch_batch(*batch, nullptr, nullptr, nullptr);  // waits for the batch; status words -> h_status
    const uint8_t *h = batch->h_out.as<uint8_t>();
    if (scores) *scores = (const double *)h;
    if (order) *order = (const int32_t *)(h + batch->out_order_off);
    if (status) {
      batch->codes.resize((size_t)std::max(batch->n_req, 1));
      std::string msg;
      for (int r = 0; r < batch->n_req; ++r) batch->codes[(size_t)r] = status_to_code(batch->h_status[(size_t)r], msg);
      *status = batch->codes.data();
    }
  });
}

void *mrk_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void mrk_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int mrk_batch_status(mrk_batch *batch, int32_t *out_status) {
  return guard([&] {
    check_batch(batch, nullptr);
    if (!out_status) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> bl(batch->bmu);
    fetch_batch(*batch, nullptr, nullptr, nullptr);
    for (int r = 0; r < batch->n_req; ++r) {
      std::string msg;
      out_status[r] = status_to_code(batch->h_status[r], msg);
    }
  });
}

// ---------------------------------------------------------------- the serving queue
// main/command/Serve.scala:130-150 (warm-up, then the port opens) and api/routes/RankApi.scala:25-41 (one rerank per
// request thread): mrk_serve_start compiles what the model needs and prepares `n_slots` slots, each a persistent workgroup
// (rank_device.hpp rank_serve_body; launched in gangs of SERVE_GANG per kernel and stream) with its request / result blocks in
// pinned memory; mrk_serve_rank is then the whole
// request path: resolve the request on the calling thread, write it into a free slot, publish, spin on the acknowledgement -
// no HIP call, no launch, no copy command.  Whatever the one-workgroup path does not cover (more than 128 candidates,
// per-item overrides, tables beyond the slot's LDS, explain) goes through mrk_rank.
}  // extern "C"

namespace {

struct ServeSlot {
  int gang = 0;                 // index into mrk_server::gangs; this slot is workgroup `index - gang.first` of the gang's kernel
  void *pinned = nullptr;       // [ServeCtl 128 B][output block][input block], coherent host memory
  ServeCtl *ctl = nullptr;
  uint8_t *h_out = nullptr, *h_in = nullptr;
  DevBuf d_in;
  HostBatch hb;
  uint32_t seq = 0;
};

// One kernel launch = one gang of up to SERVE_GANG slots on one stream.  `mu` orders launches and stops of the gang; `launch_id`
// names the launch whose workgroups are (or were last) resident - a slot whose `ctl->exited` equals it has been left.
struct ServeGang {
  hipStream_t stream = nullptr;
  int first = 0, n = 0;
  DevBuf d_slots, d_clock;      // ServeSlotDev[n]; the gang's last-activity word
  std::mutex mu;
  std::atomic<uint32_t> launch_id{0};
  bool running = false;         // (mu) a kernel was launched and has not been waited for
  bool counted = false;         // (mu) ... and is counted in resident_gangs() (runtime.hpp: memory is not given back meanwhile)
  std::atomic<bool> dead{false};   // a workgroup of it did not answer in time: none of its slots is handed out again (a late answer would land in the next request's buffers)
  bool stuck = false;           // ... and was still resident after a flush waited 2 s for it: later flushes ask once, without waiting again
};

constexpr size_t SERVE_OUT_BYTES = 2048;          // scores 128 x 8 | order 128 x 4 | status 2 x 4
constexpr size_t SERVE_IN_CAP = 64 * 1024;
constexpr int SERVE_THREADS = 512;                // 128 item lanes x op split 4; 8 wavefronts split the trees
constexpr size_t SERVE_LDS = 128 * 1024;

}  // namespace

struct mrk_server {
  mrk_ctx *ctx = nullptr;
  mrk_model *model = nullptr;
  std::string model_name;
  const Program *prog = nullptr;
  bool f64 = true;
  void *jit_fn = nullptr;
  uint64_t idle_ticks = 0, life_ticks = 0;
  std::vector<std::unique_ptr<ServeSlot>> slots;
  std::vector<std::unique_ptr<ServeGang>> gangs;
  std::atomic<int> users{0};   // callers of mrk_rank that are inside serve_fast through the context's server list
  std::atomic<int> waiting{0};          // callers between publish and acknowledgement
  // overload guard (a process with fewer CPUs than the queue has slots): overflowing callers in numbers mean the CPUs are
  // gone - everybody goes through the front, whose waiters sleep, for `front_ns`; then the queue is tried again
  bool guard = false;
  int ovf_threshold = 0;
  int64_t front_ns = 0;
  std::atomic<int64_t> front_until_ns{0}, ovf_window_ns{0};
  std::atomic<int> ovf_count{0};
  std::atomic<uint64_t> est_dev_ns{0};  // what a request takes on the device (input copy + ranking + write-back), smoothed
  std::mutex mu;
  std::condition_variable cv;
  std::vector<int> free_slots;  // a stack: the most recently used slot is the one whose workgroup is still resident
  size_t dead_slots = 0;
  bool closing = false;
  std::atomic<uint64_t> n_queue{0}, n_fallback{0}, n_launches{0};
  std::atomic<uint64_t> dev_ticks[4] = {{0}, {0}, {0}, {0}};   // device-side 100 MHz ticks: input copy, ranking, result write-back; [3]: shader cycles of the ranking
  std::atomic<uint64_t> host_ns[3] = {{0}, {0}, {0}};     // host-side ns: resolve + pack, publish -> acknowledgement, copy-out
};

namespace {

void tell_gang(mrk_server &srv, ServeGang &g, uint32_t v) {
  for (int i = g.first; i < g.first + g.n; ++i) __atomic_store_n(&srv.slots[(size_t)i]->ctl->stop, v, __ATOMIC_SEQ_CST);
}

// (g.mu held) waits until every workgroup of the gang has left: they serve what is published in their slots first.  Bounded: a
// gang that is still resident after `seconds` is retired (false).
bool drain_gang(mrk_server &srv, ServeGang &g, int seconds) {
  if (!g.running) return true;
  tell_gang(srv, g, 1u);
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(seconds);
  hipError_t q = hipStreamQuery(g.stream);
  for (uint32_t spin = 1; q == hipErrorNotReady; ++spin) {
    if ((spin & 0xffu) == 0u && std::chrono::steady_clock::now() > deadline) break;
    __builtin_ia32_pause();
    q = hipStreamQuery(g.stream);
  }
  if (q == hipErrorNotReady) {
    g.dead.store(true);
    return false;
  }
  tell_gang(srv, g, 0u);
  g.running = false;   // it left (or its launch failed): nothing of this gang touches the device any more
  if (g.counted) { g.counted = false; resident_gangs().fetch_sub(1, std::memory_order_acq_rel); }
  return true;
}

// (g.mu held; the caller holds the store shared: the device views are current)
void launch_gang(mrk_server &srv, ServeGang &g) {
  mrk_ctx *ctx = srv.ctx;
  ServeGangDev d;
  d.slots = g.d_slots.as<ServeSlotDev>();
  d.clock = g.d_clock.as<unsigned long long>();
  d.launch_id = g.launch_id.load() + 1;
  d.pad = 0;
  d.idle_ticks = srv.idle_ticks;
  d.life_ticks = srv.life_ticks;
  launch_rank_serve(ctx, g.stream, ctx->store->device_view(), srv.prog->device_view(), qs_device_view(srv.model), qs_forest_view(srv.model), d,
                    g.n, SERVE_THREADS, SERVE_LDS, srv.f64, srv.jit_fn);
  g.launch_id.store(d.launch_id);
  g.running = true;
  if (!g.counted) { g.counted = true; resident_gangs().fetch_add(1, std::memory_order_acq_rel); }
  srv.n_launches.fetch_add(1);
}

// The caller has published a request in a slot of `g` and saw (or assumes) that no workgroup of launch `seen` will serve it:
// the gang is drained and launched again as a whole, unless somebody else did meanwhile.  Returns the launch now resident.
uint32_t revive_gang(mrk_server &srv, ServeGang &g, uint32_t seen) {
  std::lock_guard<std::mutex> lk(g.mu);
  if (g.dead.load()) throw StatusError(MRK_ERR_DEVICE, "the slot's serving gang was retired");
  if (g.running && g.launch_id.load() != seen) return g.launch_id.load();
  if (!drain_gang(srv, g, 5)) throw StatusError(MRK_ERR_DEVICE, "a serving gang did not leave within 5 s (its slots are retired)");
  launch_gang(srv, g);
  return g.launch_id.load();
}

// The CPUs this process may use: its affinity mask, cut by the cgroup's quota (v2 cpu.max, v1 cfs_quota_us / cfs_period_us).
int cpu_budget() {
  static const int budget = [] {
    cpu_set_t set;
    int n = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
      fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      if (fscanf(g, "%lld", &quota) != 1) quota = -1;
      fclose(g);
      if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (fscanf(h, "%lld", &period) != 1) period = 0;
        fclose(h);
      }
    }
    if (quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
    return n;
  }();
  return budget;
}

void note_overflow(mrk_server &srv, int64_t now_ns) {   // a caller found every slot busy
  const int64_t w = srv.ovf_window_ns.load(std::memory_order_relaxed);
  if (now_ns - w > 10000000) {   // 10 ms windows
    srv.ovf_window_ns.store(now_ns, std::memory_order_relaxed);
    srv.ovf_count.store(1, std::memory_order_relaxed);
    return;
  }
  if (srv.ovf_count.fetch_add(1, std::memory_order_relaxed) + 1 >= srv.ovf_threshold) {
    srv.ovf_count.store(0, std::memory_order_relaxed);
    srv.front_until_ns.store(now_ns + srv.front_ns, std::memory_order_relaxed);
  }
}

// true: ranked through the queue (status = the request's device status word); false: not a request the queue takes
bool serve_fast(mrk_server &srv, const mrk_request *req, double *out_scores, int32_t *out_order, int &status) {
  mrk_ctx *ctx = srv.ctx;
  if (!switches().rank_serve || req->n_items > QS_TILE_ROWS) return false;
  const auto h0 = std::chrono::steady_clock::now();
  const int64_t now_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(h0.time_since_epoch()).count();
  if (srv.guard && now_ns < srv.front_until_ns.load(std::memory_order_relaxed)) return false;   // overloaded a moment ago: the front
  int si = -1;
  {
    std::lock_guard<std::mutex> lk(srv.mu);
    if (srv.closing) return false;
    while (!srv.free_slots.empty()) {
      const int c = srv.free_slots.back();
      srv.free_slots.pop_back();
      if (srv.gangs[(size_t)srv.slots[(size_t)c]->gang]->dead.load()) { srv.dead_slots += 1; continue; }   // retired with its gang
      si = c;
      break;
    }
    if (si < 0) {  // every slot busy: the batching front of mrk_rank combines the overflow
      if (srv.guard) note_overflow(srv, now_ns);
      return false;
    }
  }
  struct Release {
    mrk_server &s;
    int i;
    bool dead = false;
    ~Release() {
      std::lock_guard<std::mutex> lk(s.mu);
      if (dead || s.gangs[(size_t)s.slots[(size_t)i]->gang]->dead.load()) s.dead_slots += 1;
      else s.free_slots.push_back(i);
      s.cv.notify_one();   // under the mutex: mrk_serve_stop deletes the server as soon as it sees the last slot back
    }
  } release{srv, si};
  ServeSlot &sl = *srv.slots[(size_t)si];
  const Program &prog = locked_program(ctx, srv.model_name.c_str());
  if (&prog != srv.prog || program_mutates_store(prog) || prog.normalises()) return false;
  StoreAccess access(ctx);
  MRK_HIP(hipSetDevice(ctx->device));
  check_model_fits(srv.model, prog);
  HostBatch &hb = sl.hb;
  resolve_requests(prog, *ctx->store, req, 1, nullptr, hb);
  const int T = hb.total_items;
  uint32_t vals = 1;
  while ((int)vals < hb.max_doubles) vals <<= 1;
  const uint32_t entries = (uint32_t)std::max<uint64_t>(hb.max_req_entries, 1);
  const QsDev q = qs_device_view(srv.model);
  if (!hb.overrides.empty() || (int)prog.prep.size() > fused_max_prep() || hb.max_req_entries > (1u << 20) ||
      rank_one_lds_bytes(entries, (int)vals, SERVE_THREADS, q.thr_cap, q.n_views, srv.f64, (size_t)q.rt_doubles * 8) > SERVE_LDS)
    return false;
  // the request's input block: build_batch's arrays, 16-byte aligned
  size_t off = 0;
  auto place = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 16); return o; };
  const size_t o_reqs = place(hb.reqs.size() * sizeof(ReqDev)), o_consts = place(hb.consts.size() * 8), o_irf = place(hb.irf.size() * 4);
  const size_t o_prep = place(hb.prep_out.size() * sizeof(PrepOut)), o_slot = place((size_t)T * 4), o_ireq = place((size_t)T * 4);
  if (off > SERVE_IN_CAP) return false;
  auto put = [&](size_t o, const void *src, size_t bytes) { if (bytes) memcpy(sl.h_in + o, src, bytes); };
  put(o_reqs, hb.reqs.data(), hb.reqs.size() * sizeof(ReqDev));
  put(o_consts, hb.consts.data(), hb.consts.size() * 8);
  put(o_irf, hb.irf.data(), hb.irf.size() * 4);
  put(o_prep, hb.prep_out.data(), hb.prep_out.size() * sizeof(PrepOut));
  put(o_slot, hb.item_slot.data(), (size_t)T * 4);
  put(o_ireq, hb.item_req.data(), (size_t)T * 4);
  ServeCtl &ctl = *sl.ctl;
  ctl.in_bytes = (uint32_t)std::max<size_t>(off, 16);
  ctl.o_reqs = (uint32_t)o_reqs; ctl.o_consts = (uint32_t)o_consts; ctl.o_irf = (uint32_t)o_irf;
  ctl.o_prep = (uint32_t)o_prep; ctl.o_slot = (uint32_t)o_slot; ctl.o_ireq = (uint32_t)o_ireq;
  ctl.total_items = (uint32_t)T; ctl.tab_entries = entries; ctl.vals_cap = vals; ctl.mode = 4;
  const uint32_t seq = ++sl.seq;
  if (seq == 0xffffffffu) sl.seq = 0;  // (never: 4 G requests through one slot)
  const auto h1 = std::chrono::steady_clock::now();
  __atomic_store_n(&ctl.seq, seq, __ATOMIC_SEQ_CST);  // publishes the block and the header
  ServeGang &g = *srv.gangs[(size_t)sl.gang];
  uint32_t seen = g.launch_id.load();   // (0: never launched - no slot's `exited` word holds it)
  auto gone = [&] { return seen == 0u || __atomic_load_n(&ctl.exited, __ATOMIC_SEQ_CST) == seen; };
  auto acked = [&] { return __atomic_load_n(&ctl.ack, __ATOMIC_ACQUIRE) == seq; };
  if (gone()) seen = revive_gang(srv, g, seen);
  // Waiting: the first few callers spin on the acknowledgement (nothing is faster); the callers beyond them sleep through
  // the time the device is known to need and spin for the rest.  64 callers spinning for 0.1 ms each are 64 busy CPUs: on a
  // host whose CPU quota is smaller (the measurement boxes: profiles/r06_ad) the scheduler parks them for whole periods -
  // p50 0.15 ms, maximum 77 ... 400 ms, and the closed-loop rate FELL from 32 to 64 callers.
  struct Waiting {
    std::atomic<int> &n;
    int mine;
    explicit Waiting(std::atomic<int> &c) : n(c), mine(c.fetch_add(1) + 1) {}
    ~Waiting() { n.fetch_sub(1); }
  } waiting(srv.waiting);
  const bool sleeper = waiting.mine > switches().serve_spin_callers;
  if (sleeper) {
    static thread_local bool slack_set = false;
    if (!slack_set) { (void)prctl(PR_SET_TIMERSLACK, 1000ul, 0ul, 0ul, 0ul); slack_set = true; }   // this thread's timers: 1 us instead of 50
    // (publish -> acknowledgement is the device's total + 15 ... 20 us of PCIe round trips: profiles/r06_ad, 96 against 77 us)
    const uint64_t est = srv.est_dev_ns.load(std::memory_order_relaxed);
    if (est > 20000 && !acked()) {
      struct timespec ts = {0, (long)std::min<uint64_t>(est + (uint64_t)switches().serve_sleep_extra_us * 1000, 2000000)};
      (void)nanosleep(&ts, nullptr);
    }
  }
  // ... and a sleeper does not spin for the rest either: it looks every few microseconds (`serve_poll_us`; 0 = spin).  Spinning
  // tails kept 64 callers just inside a 16-CPU quota and 80 callers outside it: the whole process was throttled for 100 - 200 ms
  // at a time and the rate fell from 360 k to 180 k requests/s (profiles/r06_aj).
  const long poll_ns = sleeper ? (long)switches().serve_poll_us * 1000 : 0;
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(5);
  for (uint32_t spin = 1; !acked(); ++spin) {
    if (poll_ns > 0 || (spin & 31u) == 0u) {
      if (gone()) {  // the workgroup left (idle / old / told to stop) - possibly after serving this request
        if (acked()) break;
        seen = revive_gang(srv, g, seen);
      } else if ((spin & (poll_ns > 0 ? 0xffu : 0xffffu)) == 0u && std::chrono::steady_clock::now() > deadline) {
        tell_gang(srv, g, 1u);  // should they still be alive: leave
        g.dead.store(true);
        release.dead = true;
        char why[256];
        snprintf(why, sizeof why, "the serving workgroup did not answer within 5 s (its gang is retired; slot %d: seq %u ack %u exited %u stop %u, gang launch %u seen %u)",
                 si, __atomic_load_n(&ctl.seq, __ATOMIC_SEQ_CST), __atomic_load_n(&ctl.ack, __ATOMIC_SEQ_CST), __atomic_load_n(&ctl.exited, __ATOMIC_SEQ_CST),
                 __atomic_load_n(&ctl.stop, __ATOMIC_SEQ_CST), g.launch_id.load(), seen);
        throw StatusError(MRK_ERR_DEVICE, why);
      }
    }
    if (poll_ns > 0) {
      struct timespec ts = {0, poll_ns};
      (void)nanosleep(&ts, nullptr);
    } else {
      __builtin_ia32_pause();
    }
  }
  const auto h2 = std::chrono::steady_clock::now();
  if (out_scores && T) memcpy(out_scores, sl.h_out, (size_t)T * 8);
  if (out_order && T) memcpy(out_order, sl.h_out + 1024, (size_t)T * 4);
  const int32_t *hs = (const int32_t *)(sl.h_out + 1536);
  status = hs[0] | hs[1];
  const unsigned long long *clk = (const unsigned long long *)(hs + 16);
  for (int k = 0; k < 4; ++k) srv.dev_ticks[k].fetch_add(clk[k]);
  {
    const uint64_t sample = (clk[0] + clk[1] + clk[2]) * 10, old = srv.est_dev_ns.load(std::memory_order_relaxed);   // 100 MHz ticks -> ns
    if (sample < 10000000) srv.est_dev_ns.store(old ? (old * 7 + sample) / 8 : sample, std::memory_order_relaxed);
  }
  const auto h3 = std::chrono::steady_clock::now();
  srv.host_ns[0].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(h1 - h0).count());
  srv.host_ns[1].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(h2 - h1).count());
  srv.host_ns[2].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(h3 - h2).count());
  srv.n_queue.fetch_add(1);
  return true;
}

}  // namespace

namespace mrk {
static void quiesce_servers(mrk_ctx *ctx) {  // the caller holds the store exclusively: no request is in flight in any slot
  std::lock_guard<std::mutex> lk(ctx->servers_mu);
  for (void *p : ctx->servers) {
    mrk_server *srv = (mrk_server *)p;
    for (auto &g : srv->gangs) {   // tell all of them first: they leave side by side
      std::lock_guard<std::mutex> gl(g->mu);
      if (g->running && !g->dead.load()) tell_gang(*srv, *g, 1u);
    }
    for (auto &g : srv->gangs) {
      std::lock_guard<std::mutex> gl(g->mu);
      if (!g->running) continue;
      if (!g->dead.load()) {
        if (drain_gang(*srv, *g, 5)) continue;
      }
      // A retired gang (a workgroup of it missed the 5 s deadline) was told to stop when it was retired.  Slow is not hung: if
      // its kernel is still on the device it still reads the store views it was launched with, and the caller is about to
      // reallocate them - wait for it a while, and refuse the flush rather than free memory under a live kernel.
      // (the 2 s are spent ONCE per stuck gang: while it stays resident every later flush fails at once instead of holding
      //  the store exclusively for another 2 s per request - one stuck workgroup is an error state, not a stall of the service)
      const auto deadline = std::chrono::steady_clock::now() + (g->stuck ? std::chrono::seconds(0) : std::chrono::seconds(2));
      hipError_t q = hipStreamQuery(g->stream);
      while (q == hipErrorNotReady && std::chrono::steady_clock::now() < deadline) {
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
        q = hipStreamQuery(g->stream);
      }
      if (q == hipErrorNotReady) {
        g->stuck = true;
        throw StatusError(MRK_ERR_DEVICE, "a retired serving workgroup is still resident: the store is not reallocated under it");
      }
      g->stuck = false;
      g->running = false;   // it left (or its launch failed): nothing of this gang touches the device any more
      if (g->counted) { g->counted = false; resident_gangs().fetch_sub(1, std::memory_order_acq_rel); }
    }
  }
  if (resident_gangs().load(std::memory_order_acquire) == 0) flush_deferred_releases();
}
}  // namespace mrk

extern "C" {

int mrk_serve_start(mrk_ctx *ctx, mrk_model *model, const char *model_name, int n_slots, mrk_server **out) {
  return guard([&] {
    if (!out) throw StatusError(MRK_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!ctx || !model || !model_name || n_slots < 1 || n_slots > 64) throw StatusError(MRK_ERR_INVALID_ARG, "bad arguments (1..64 slots)");
    if (model->ctx != ctx) throw StatusError(MRK_ERR_INVALID_ARG, "model belongs to another context");
    const Program &prog = locked_program(ctx, model_name);
    if (!model->qs.ok) throw StatusError(MRK_ERR_UNSUPPORTED, "the serving queue scores with the bit-vector scorer (trees of <= 16 leaves); use mrk_rank for this model");
    check_model_fits(model, prog);
    MRK_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<mrk_server> srv(new mrk_server());
    srv->ctx = ctx;
    srv->model = model;
    srv->model_name = model_name;
    srv->prog = &prog;
    srv->f64 = model->forest.backend == Backend::LightGBM;
    srv->idle_ticks = (uint64_t)std::max(1, switches().serve_idle_us) * 100ull;  // wall_clock64: 100 MHz
    srv->life_ticks = (uint64_t)std::max(1, switches().serve_life_us) * 100ull;
    srv->jit_fn = jit_serve_function(prog, srv->f64, switches().thr_stage ? &model->qs_sig : nullptr);  // warm-up: the compile happens here, not under the first request
    srv->front_ns = (int64_t)switches().serve_overload_ms * 1000000;
    // A resident kernel holds its stream's hardware queue for as long as it stays, and streams that share a hardware queue wait
    // for each other: round 5's "collapse at 32 callers" (p99 76 ms) was a slot whose stream had landed behind another slot's
    // resident kernel (ROCm maps a process's streams onto GPU_MAX_HW_QUEUES = 4 hardware queues by default; mrk_init asks for
    // 24, hw_queue_budget in capi.cpp).  One kernel per slot (round 6's first form) therefore capped the slots at the queues, and
    // the callers beyond them - sent through mrk_rank's front, whose lanes share the same queues - still met 150 ... 230 ms stalls.
    // Slots are launched in GANGS instead: SERVE_GANG workgroups per kernel, one stream per gang - 64 slots on 8 streams.
    const int n_gangs = std::min((n_slots + SERVE_GANG - 1) / SERVE_GANG, std::max(1, hw_queue_budget() - 4));   // 4: the context stream, the lanes of mrk_rank's front, a batch
    n_slots = std::min(n_slots, n_gangs * SERVE_GANG);
    for (int i = 0; i < n_slots; ++i) {
      std::unique_ptr<ServeSlot> sl(new ServeSlot());
      sl->gang = i / SERVE_GANG;
      MRK_HIP(hipHostMalloc(&sl->pinned, 128 + SERVE_OUT_BYTES + SERVE_IN_CAP, hipHostMallocCoherent | hipHostMallocMapped));
      memset(sl->pinned, 0, 128 + SERVE_OUT_BYTES + SERVE_IN_CAP);
      static_assert(sizeof(ServeCtl) == 128, "ServeCtl is one 128-byte line");
      sl->ctl = (ServeCtl *)sl->pinned;
      sl->h_out = (uint8_t *)sl->pinned + 128;
      sl->h_in = sl->h_out + SERVE_OUT_BYTES;
      sl->d_in.reserve(SERVE_IN_CAP);
      srv->slots.push_back(std::move(sl));
      srv->free_slots.push_back(n_slots - 1 - i);  // slot 0 on top
    }
    for (int k = 0; k < n_gangs; ++k) {
      std::unique_ptr<ServeGang> g(new ServeGang());
      g->first = k * SERVE_GANG;
      g->n = std::min(SERVE_GANG, n_slots - g->first);
      MRK_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
      std::vector<ServeSlotDev> view((size_t)g->n);
      for (int i = 0; i < g->n; ++i) {
        ServeSlot &sl = *srv->slots[(size_t)(g->first + i)];
        view[(size_t)i].ctl = sl.ctl;
        view[(size_t)i].in_host = sl.h_in;
        view[(size_t)i].in_dev = sl.d_in.as<uint8_t>();
        view[(size_t)i].out = OneOut{(double *)sl.h_out, (int32_t *)(sl.h_out + 1024), (int32_t *)(sl.h_out + 1536), nullptr, 1};
      }
      g->d_slots.reserve(view.size() * sizeof(ServeSlotDev));
      g->d_clock.reserve(8);
      MRK_HIP(hipMemcpy(g->d_slots.p, view.data(), view.size() * sizeof(ServeSlotDev), hipMemcpyHostToDevice));
      MRK_HIP(hipMemset(g->d_clock.p, 0, 8));
      srv->gangs.push_back(std::move(g));
    }
    srv->guard = srv->front_ns > 0 && cpu_budget() < n_slots;   // fewer CPUs than slots: callers in overflow are CPUs that are not there
    srv->ovf_threshold = std::max(3 * n_slots, 96);
    mrk_model_retain(model);
    ctx_retain(ctx);
    {
      std::lock_guard<std::mutex> lk(ctx->servers_mu);
      ctx->servers.push_back(srv.get());
      ctx->n_servers.store((int)ctx->servers.size(), std::memory_order_release);
      ctx->hot_server.store(srv.get(), std::memory_order_seq_cst);
    }
    *out = srv.release();
  });
}

static int serve_one(mrk_server *srv, const mrk_request *req, double *out_scores, int32_t *out_order, bool &done) {
  int status = 0;
  done = false;
  const int rc = guard([&] { done = serve_fast(*srv, req, out_scores, out_order, status); });
  if (rc != MRK_OK || !done) return rc;
  std::string msg;
  const int code = status_to_code(status, msg);
  if (code != MRK_OK) set_last_error(msg);
  return code;
}

int mrk_serve_rank(mrk_server *srv, const mrk_request *req, double *out_scores, int32_t *out_order) {
  if (!srv || !req) { set_last_error("null argument"); return MRK_ERR_INVALID_ARG; }
  bool done = false;
  const int rc = serve_one(srv, req, out_scores, out_order, done);
  if (done || rc != MRK_OK) return rc;
  srv->n_fallback.fetch_add(1);
  return rank_front(srv->ctx, srv->model, srv->model_name.c_str(), req, out_scores, out_order, nullptr);
}

// mrk_rank's way into the queue: the context's server for (model, model_name), if one was started.  `users` keeps
// mrk_serve_stop from freeing it between the lookup and the slot.
static int rank_through_server(mrk_ctx *ctx, mrk_model *model, const char *model_name, const mrk_request *req, double *out_scores,
                               int32_t *out_order, bool &done) {
  done = false;
  if (req->n_items > QS_TILE_ROWS) return MRK_OK;
  // The context's most recently started queue, without a lock (64 callers a few hundred thousand times a second on one mutex
  // are a queue of their own, and a holder the scheduler parks stops them all).  Readers count themselves into the current
  // epoch's cell BEFORE they load the pointer; mrk_serve_stop withdraws the pointer, turns the epoch and waits for the old
  // cell to drain - whoever could still hold the withdrawn pointer is in it.
  {
    int cell = -1;
    for (;;) {
      const uint32_t e = ctx->hot_epoch.load(std::memory_order_seq_cst);
      ctx->hot_readers[e & 1].fetch_add(1, std::memory_order_seq_cst);
      if (ctx->hot_epoch.load(std::memory_order_seq_cst) == e) { cell = (int)(e & 1); break; }
      ctx->hot_readers[e & 1].fetch_sub(1, std::memory_order_seq_cst);
    }
    mrk_server *hot = (mrk_server *)ctx->hot_server.load(std::memory_order_seq_cst);
    int rc = MRK_OK;
    const bool mine = hot && hot->model == model && hot->model_name == model_name;
    if (mine) {
      rc = serve_one(hot, req, out_scores, out_order, done);
      if (!done && rc == MRK_OK) hot->n_fallback.fetch_add(1);
    }
    ctx->hot_readers[cell].fetch_sub(1, std::memory_order_seq_cst);
    if (mine) return rc;
  }
  mrk_server *srv = nullptr;
  {
    std::lock_guard<std::mutex> lk(ctx->servers_mu);
    for (void *p : ctx->servers) {
      mrk_server *c = (mrk_server *)p;
      if (c->model == model && c->model_name == model_name) { srv = c; break; }
    }
    if (srv) srv->users.fetch_add(1);
  }
  if (!srv) return MRK_OK;
  const int rc = serve_one(srv, req, out_scores, out_order, done);
  if (!done && rc == MRK_OK) srv->n_fallback.fetch_add(1);
  srv->users.fetch_sub(1);
  return rc;
}

int mrk_serve_stats(mrk_server *srv, int64_t *out, int n_out) {
  if (!srv || !out || n_out < 0) return MRK_ERR_INVALID_ARG;
  int64_t v[MRK_SERVE_STATS] = {};
  v[0] = (int64_t)srv->n_queue.load();
  v[1] = (int64_t)srv->n_fallback.load();
  v[2] = (int64_t)srv->n_launches.load();
  for (int k = 0; k < 3; ++k) {
    v[3 + k] = (int64_t)srv->host_ns[k].load();
    v[6 + k] = (int64_t)(srv->dev_ticks[k].load() * 10);   // 100 MHz ticks -> ns
  }
  v[9] = (int64_t)srv->dev_ticks[3].load();
  for (int k = 0; k < n_out && k < MRK_SERVE_STATS; ++k) out[k] = v[k];   // never past the caller's buffer, whatever ABI it was built for
  return MRK_OK;
}

void mrk_serve_stop(mrk_server *srv) {
  if (!srv) return;
  mrk_ctx *ctx = srv->ctx;
  {
    std::unique_lock<std::mutex> lk(srv->mu);
    srv->closing = true;
    srv->cv.wait(lk, [&] { return srv->free_slots.size() + srv->dead_slots == srv->slots.size(); });  // requests in flight finish first
  }
  {
    std::lock_guard<std::mutex> lk(ctx->servers_mu);
    ctx->servers.erase(std::remove(ctx->servers.begin(), ctx->servers.end(), (void *)srv), ctx->servers.end());
    ctx->n_servers.store((int)ctx->servers.size(), std::memory_order_release);
    if (ctx->hot_server.load() == (void *)srv) ctx->hot_server.store(ctx->servers.empty() ? nullptr : ctx->servers.back(), std::memory_order_seq_cst);
  }
  {  // mrk_rank's lock-free readers that may still hold the withdrawn pointer (rank_through_server)
    std::lock_guard<std::mutex> lk(ctx->hot_stop_mu);
    const uint32_t e = ctx->hot_epoch.fetch_add(1, std::memory_order_seq_cst);
    while (ctx->hot_readers[e & 1].load(std::memory_order_seq_cst) != 0) std::this_thread::yield();
  }
  while (srv->users.load() != 0) std::this_thread::yield();   // callers of mrk_rank that found this server before it left the list
  (void)hipSetDevice(ctx->device);
  for (auto &g : srv->gangs) {
    std::lock_guard<std::mutex> gl(g->mu);
    if (!g->dead.load()) (void)drain_gang(*srv, *g, 5);
    if (g->dead.load() && g->running) {
      // its workgroups may still be resident, polling their slots' pinned blocks: stream, blocks and device buffers are leaked
      // on purpose (a hipFree would wait for the resident kernel - for ever, if it hangs)
      g->d_slots.p = nullptr; g->d_slots.cap = 0;
      g->d_clock.p = nullptr; g->d_clock.cap = 0;
      for (int i = g->first; i < g->first + g->n; ++i) { srv->slots[(size_t)i]->d_in.p = nullptr; srv->slots[(size_t)i]->d_in.cap = 0; }
      continue;
    }
    if (g->stream) (void)hipStreamDestroy(g->stream);
    for (int i = g->first; i < g->first + g->n; ++i) {
      ServeSlot &sl = *srv->slots[(size_t)i];
      if (sl.pinned) (void)hipHostFree(sl.pinned);
    }
  }
  mrk_model_free(srv->model);
  delete srv;
  if (resident_gangs().load(std::memory_order_acquire) == 0) flush_deferred_releases();
  ctx_release(ctx);
}

void mrk_batch_free(mrk_batch *batch) {
  if (!batch) return;
  mrk_ctx *ctx = batch->ctx;
  if (ctx) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(batch->s());
    if (batch->stream) (void)hipStreamDestroy(batch->stream);
    delete batch;
    ctx_release(ctx);
  } else {
    delete batch;
  }
}

}  // extern "C"
