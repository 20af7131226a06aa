// C ABI (include/mrk.h): lifecycle, model handles, predictMat replacement, profiling.
// Feature store / rank entry points live in capi_rank.cpp.
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <thread>
#include <utility>

#include "runtime.hpp"

namespace mrk {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }

static Switches read_switches() {
  Switches s;
  auto flag = [](const char *name, bool dflt) { const char *e = getenv(name); return e ? atoi(e) != 0 : dflt; };
  auto num = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
  s.rank_fused = flag("MRK_RANK_FUSED", true);
  s.rank_cells = flag("MRK_RANK_CELLS", true);
  if (const char *e = getenv("MRK_SCORER")) s.scorer_walk = !strcmp(e, "walk");
  if (const char *e = getenv("MRK_FUSED_THREADS")) s.fused_threads = std::min(256, std::max(64, atoi(e) / 64 * 64));
  { const int fs = num("MRK_FUSED_SPLIT", 0); s.fused_split = fs == 1 || fs == 2 || fs == 4 ? fs : 0; }
  s.fused_slices = std::max(0, num("MRK_FUSED_SLICES", 0));
  s.prepass_lds = flag("MRK_PREPASS_LDS", true);
  s.rank_combine = flag("MRK_RANK_COMBINE", true);
  s.rank_one = flag("MRK_RANK_ONE", true);
  s.rank_fused_score = flag("MRK_RANK_FUSED_SCORE", false);
  s.rank_serve = flag("MRK_RANK_SERVE", true);
  s.serve_idle_us = std::max(1, num("MRK_SERVE_IDLE_US", 2000));
  s.serve_spin_callers = std::max(0, num("MRK_SERVE_SPIN_CALLERS", 8));
  s.serve_sleep_extra_us = std::max(0, num("MRK_SERVE_SLEEP_EXTRA_US", 8));
  s.serve_overload_ms = std::max(0, num("MRK_SERVE_OVERLOAD_MS", 200));
  s.serve_poll_us = std::max(0, num("MRK_SERVE_POLL_US", 4));
  s.serve_life_us = std::max(1, num("MRK_SERVE_LIFE_US", 20000));
  s.combine_max = std::max(1, num("MRK_RANK_COMBINE_MAX", 256));
  s.rank_lanes = std::max(1, std::min((int)mrk_ctx::RANK_LANES_MAX, num("MRK_RANK_LANES", 3)));
  s.table_load_pct = std::max(10, std::min(90, num("MRK_TABLE_LOAD_PCT", 75)));
  s.host_threads = std::max(0, std::min(256, num("MRK_HOST_THREADS", 0)));
  if (const char *e = getenv("MRK_RANK_JIT")) s.jit_mode = !strcmp(e, "require") ? 2 : !strcmp(e, "async") ? 3 : !strcmp(e, "auto") ? 4 : atoi(e) != 0 ? 1 : 0;
  s.jit_waves = num("MRK_JIT_WAVES", 0);
  s.jit_record_regs = flag("MRK_JIT_REGS", true);
  s.jit_sig = flag("MRK_JIT_SIG", true);
  s.items_lds = flag("MRK_ITEMS_LDS", true);
  s.items_rt = flag("MRK_ITEMS_RT", true);
  s.rank_one_max = std::max(1, std::min(256, num("MRK_RANK_ONE_MAX", 128)));
  s.split_max_req = std::max(0, std::min(256, num("MRK_SPLIT_MAX_REQ", 64)));
  s.items_rt_threads = num("MRK_ITEMS_RT_THREADS", 0);
  s.jit_shipped = flag("MRK_JIT_SHIPPED", true);
  if (const char *d = getenv("MRK_JIT_DEFINES")) s.jit_defines = d; else s.jit_defines.clear();
  s.thr_stage = flag("MRK_THR_STAGE", true);
  s.fused_lds_min = std::max(0, num("MRK_FUSED_LDS_MIN", 0));
  if (const char *d = getenv("MRK_JIT_CACHE_DIR")) s.jit_cache_dir = d;
  else if (const char *x = getenv("XDG_CACHE_HOME")) s.jit_cache_dir = std::string(x) + "/mrk_jit";
  else if (const char *h = getenv("HOME")) s.jit_cache_dir = std::string(h) + "/.cache/mrk_jit";
  if (s.jit_cache_dir == "off") s.jit_cache_dir.clear();
  s.big_sort_cap = num("MRK_BIG_SORT_CAP", 4096);
  s.big_sort_tile = num("MRK_BIG_SORT_TILE", 0);
  s.big_sort_bucket = num("MRK_BIG_SORT_BUCKET", 0);
  s.jit_prepass = flag("MRK_JIT_PREPASS", true);
  s.qs_split = num("MRK_QS_SPLIT", -1);
  s.qs_kernel = num("MRK_QS_KERNEL", 1);
  s.qs_r = num("MRK_QS_R", 2);
  s.walk_tile = num("MRK_WALK_TILE", 0);
  s.encoder_graph = flag("MRK_ENCODER_GRAPH", false);
  s.encoder_skinny = num("MRK_ENCODER_SKINNY", 15);
  s.encoder_packed = flag("MRK_ENCODER_PACKED", true);
  s.encoder_f32_mfma = flag("MRK_ENCODER_F32_MFMA", true);
  return s;
}
static Switches &switches_storage() {
  static Switches s = read_switches();
  return s;
}
const Switches &switches() { return switches_storage(); }
void reload_switches() { switches_storage() = read_switches(); }

ScopedKernelTimer::ScopedKernelTimer(mrk_ctx *c, const char *n) : ctx(c), name(n) {
  if (!ctx->profile) return;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
    a = b = nullptr;
    return;
  }
  (void)hipEventRecord(a, ctx->launch);
}
ScopedKernelTimer::~ScopedKernelTimer() {
  if (!a || !b) return;
  (void)hipEventRecord(b, ctx->launch);
  ctx->pending_events.emplace_back(name, a, b);
}
void drain_profile_events(mrk_ctx *ctx) {
  for (auto &e : ctx->pending_events) {
    hipEvent_t a = std::get<1>(e), b = std::get<2>(e);
    if (hipEventSynchronize(b) == hipSuccess) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, a, b) == hipSuccess) {
        auto &t = ctx->timers[std::get<0>(e)];
        t.total_ms += ms;
        t.launches += 1;
      }
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
  }
  ctx->pending_events.clear();
}

std::atomic<int> &resident_gangs() {
  static std::atomic<int> n{0};
  return n;
}

namespace {
std::mutex &deferred_mu() { static std::mutex m; return m; }
std::vector<std::pair<void *, bool>> &deferred() { static std::vector<std::pair<void *, bool>> v; return v; }   // (pointer, pinned)
}  // namespace

void release_device(void *p) {
  if (resident_gangs().load(std::memory_order_acquire) > 0) {
    std::lock_guard<std::mutex> lk(deferred_mu());
    deferred().emplace_back(p, false);
    return;
  }
  (void)hipFree(p);
}

void release_pinned(void *p) {
  if (resident_gangs().load(std::memory_order_acquire) > 0) {
    std::lock_guard<std::mutex> lk(deferred_mu());
    deferred().emplace_back(p, true);
    return;
  }
  (void)hipHostFree(p);
}

void flush_deferred_releases() {
  std::vector<std::pair<void *, bool>> mine;
  {
    std::lock_guard<std::mutex> lk(deferred_mu());
    mine.swap(deferred());
  }
  for (auto &e : mine) {
    if (e.second) (void)hipHostFree(e.first);
    else (void)hipFree(e.first);
  }
}

int hw_queue_budget() {
  // Persistent serving workgroups (mrk_serve_*) each hold a stream's hardware queue for as long as they stay; with the runtime's
  // default of 4 queues the fifth slot already waits behind another slot's kernel.  The variable is read when HIP initialises:
  // setting it here works when this library makes the process's first HIP call (a JVM host: always), and never overrides the host.
  static const int budget = [] {
    (void)setenv("GPU_MAX_HW_QUEUES", "24", 0);
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    const int v = e ? atoi(e) : 4;
    return v >= 1 ? v : 4;
  }();
  return budget;
}

void lds_optin(mrk_ctx *ctx, const void *fn, int bytes) {
  static thread_local const void *last_fn = nullptr;
  static thread_local int last_dev = -1;
  if (last_fn == fn && last_dev == ctx->device) return;
  static std::mutex mu;
  static std::set<std::pair<int, const void *>> done;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert({ctx->device, fn}).second) {
      int cur = -1;
      MRK_HIP(hipGetDevice(&cur));
      if (cur != ctx->device) MRK_HIP(hipSetDevice(ctx->device));
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (cur != ctx->device && cur >= 0) (void)hipSetDevice(cur);
      if (e != hipSuccess) {
        done.erase({ctx->device, fn});
        throw StatusError(MRK_ERR_DEVICE, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
      }
    }
  }
  last_fn = fn;
  last_dev = ctx->device;
}

void ctx_retain(mrk_ctx *ctx) { ctx->refs.fetch_add(1); }
void ctx_release(mrk_ctx *ctx) {
  if (ctx->refs.fetch_sub(1) != 1) return;
  (void)hipSetDevice(ctx->device);
  comm_destroy(ctx);
  if (ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
    drain_profile_events(ctx);
    (void)hipStreamDestroy(ctx->stream);
  }
  delete ctx;
}

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
  } catch (const UnsupportedModel &e) {
    set_last_error(e.what());
    return MRK_ERR_UNSUPPORTED;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return MRK_ERR_PARSE;
  }
}

static void upload_model(mrk_ctx *ctx, mrk_model *m) {
  m->packed = pack_forest(m->forest, score_chunk_budget());
  MRK_HIP(hipSetDevice(ctx->device));
  auto up = [&](DevBuf &b, const void *src, size_t bytes) {
    b.reserve(bytes ? bytes : 16);
    if (bytes) MRK_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
  };
  up(m->d_image, m->packed.image.data(), m->packed.image.size());
  up(m->d_trees, m->packed.trees.data(), m->packed.trees.size() * sizeof(TreeRef));
  up(m->d_chunks, m->packed.chunks.data(), m->packed.chunks.size() * sizeof(ChunkRef));
  up(m->d_cat, m->forest.cat_bits.data(), m->forest.cat_bits.size() * 4);
  m->qs = pack_forest_qs(m->forest, m->forest.n_features);
  if (m->qs.ok) {
    up(m->d_qs_nodes, m->qs.nodes.data(), m->qs.nodes.size() * 4);
    up(m->d_qs_leaves, m->qs.leaves.data(), m->qs.leaves.size());
    {  // the assembly kernel stages tables with 1 KiB wave-loads that may run past a table's end: slack after the last one
      std::vector<double> padded(m->qs.thr);
      padded.resize(padded.size() + 2 * QS_STAGE_CHUNK, 0.0);
      // ... and behind the slack the compact tables of the resident-table sinks (score_qs.hip qs_device_view: QsDev::thr_rt)
      padded.insert(padded.end(), m->qs.thr_rt.begin(), m->qs.thr_rt.end());
      up(m->d_qs_thr, padded.data(), padded.size() * 8);
    }
    up(m->d_qs_feats, m->qs.feats.data(), m->qs.feats.size() * sizeof(QsFeature));
    up(m->d_qs_views, m->qs.views.data(), m->qs.views.size() * sizeof(QsView));
    up(m->d_qs_catnodes, m->qs.cat_nodes.data(), m->qs.cat_nodes.size() * sizeof(QsCatNode));
    up(m->d_qs_cat, m->qs.cat_bits.data(), m->qs.cat_bits.size() * 4);
    m->qs_sig = qs_signature(m->qs, qs_stage_cap(m->qs));
  }
}

static mrk_model *make_model(mrk_ctx *ctx, int backend, const uint8_t *bytes, size_t len) {
  if (!ctx) throw StatusError(MRK_ERR_INVALID_ARG, "null context");
  if (!bytes && len) throw StatusError(MRK_ERR_INVALID_ARG, "null model bytes");
  std::unique_ptr<mrk_model> m(new mrk_model());
  m->ctx = ctx;
  if (backend == MRK_BACKEND_LIGHTGBM) m->forest = parse_lightgbm_text((const char *)bytes, len);
  else if (backend == MRK_BACKEND_XGBOOST) m->forest = parse_xgboost(bytes, len);
  else throw StatusError(MRK_ERR_INVALID_ARG, "unsupported booster tag " + std::to_string(backend));
  if (m->forest.average_output)
    throw StatusError(MRK_ERR_UNSUPPORTED, "lightgbm: average_output (random forest) models are not supported");
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (ctx->closed) throw StatusError(MRK_ERR_INVALID_ARG, "context is shut down");
  upload_model(ctx, m.get());
  ctx_retain(ctx);
  return m.release();
}

}  // namespace mrk

using namespace mrk;

mrk_ctx::mrk_ctx() {}
mrk_ctx::~mrk_ctx() { mrk::free_rank_state(this); }

extern "C" {

int mrk_abi_version(void) { return MRK_ABI_VERSION; }
const char *mrk_build_id(void) {
  return
#include "build_id.inc"
      ;
}
/* not part of include/mrk.h: tests / measurement scripts that change an experiment switch inside one process */
void mrk_debug_reload_switches(void) { mrk::reload_switches(); }
const char *mrk_last_error(void) { return g_last_error.c_str(); }

int mrk_device_count(void) {
  (void)hw_queue_budget();
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return count;
}

// one context on one device (streams, flags); mrk_init makes one per listed device
static mrk_ctx *make_context(int device) {
  std::unique_ptr<mrk_ctx> ctx(new mrk_ctx());
  ctx->device = device;
  MRK_HIP(hipSetDevice(ctx->device));
  hipDeviceProp_t prop;
  MRK_HIP(hipGetDeviceProperties(&prop, ctx->device));
  ctx->n_cus = prop.multiProcessorCount;
  ctx->lds_per_block = prop.sharedMemPerBlock;
  MRK_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  ctx->launch = ctx->stream;
  ctx->d_flag.reserve(256);
  ctx->h_flag.reserve(4096);
  MRK_HIP(hipMemset(ctx->d_flag.p, 0, 256));
  return ctx.release();
}

int mrk_init(const int *device_ids, int n_devices, mrk_ctx **out) {
  return guard([&] {
    if (!out) throw StatusError(MRK_ERR_INVALID_ARG, "out is null");
    for (int i = 0; i < std::max(n_devices, 1); ++i) out[i] = nullptr;
    if (n_devices < 1 || !device_ids) throw StatusError(MRK_ERR_INVALID_ARG, "need at least one device id");
    (void)hw_queue_budget();   // before the process's first HIP call
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
      throw StatusError(MRK_ERR_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e));
    for (int i = 0; i < n_devices; ++i)
      if (device_ids[i] < 0 || device_ids[i] >= count) throw StatusError(MRK_ERR_INVALID_ARG, "device id out of range");
    // HipConfig(devices: List[Int]) inside ONE host process (SURVEY 8b touch point 1): a context per listed device, each with its
    // own streams, store replica and models; nothing in the library is per-process, every entry point selects its context's
    // device for the calling thread.  The same ordinal may be listed more than once (independent contexts sharing a GPU).
    std::vector<std::unique_ptr<mrk_ctx>> made;
    try {
      for (int i = 0; i < n_devices; ++i) made.emplace_back(make_context(device_ids[i]));
    } catch (...) {
      for (auto &c : made) mrk_shutdown(c.release());
      throw;
    }
    for (int i = 0; i < n_devices; ++i) out[i] = made[(size_t)i].release();
  });
}

void mrk_shutdown(mrk_ctx *ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->closed) return;
    ctx->closed = true;
  }
  unbind_encoders(ctx);  // bound encoders hold a context reference each
  ctx_release(ctx);  // freed once the last model / batch handle is gone
}

int mrk_model_load(mrk_ctx *ctx, int backend, const uint8_t *bytes, size_t len, mrk_model **out) {
  return guard([&] {
    if (!out) throw StatusError(MRK_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    *out = make_model(ctx, backend, bytes, len);
  });
}

int mrk_model_load_container(mrk_ctx *ctx, const uint8_t *blob, size_t len, const char *const *feature_names,
                             int n_features, mrk_model **out) {
  return guard([&] {
    if (!out) throw StatusError(MRK_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!blob) throw StatusError(MRK_ERR_INVALID_ARG, "cannot load model: not found, maybe you forgot to run train?");
    Container c = parse_container(blob, len);
    if (feature_names) {
      bool same = (int)c.features.size() == n_features;
      for (int i = 0; same && i < n_features; ++i) same = c.features[i] == feature_names[i];
      if (!same) {
        std::string exp, act;
        for (auto &s : c.features) exp += (exp.empty() ? "" : ", ") + s;
        for (int i = 0; i < n_features; ++i) act += (act.empty() ? "" : ", ") + std::string(feature_names[i]);
        throw StatusError(MRK_ERR_FEATURE_MISMATCH, "booster trained with List(" + exp + ") features, but config defines List(" + act +
                                                        ")\nYou may need to retrain the model with the newer config");
      }
    }
    mrk_model *m = make_model(ctx, c.booster_tag, c.inner, c.inner_len);
    m->container_features = c.features;
    m->n_warmup = c.n_warmup;
    if (c.n_warmup > 0) m->warmup_bytes.assign(c.warmup, c.warmup + c.warmup_len);
    *out = m;
  });
}

static void check_predict_args(mrk_model *model, const void *x, int rows, int cols, const void *out) {
  if (!model) throw StatusError(MRK_ERR_INVALID_ARG, "null model");
  if (model->refs.load() <= 0) throw StatusError(MRK_ERR_INVALID_ARG, "model is closed");
  if (rows < 0 || cols < 0) throw StatusError(MRK_ERR_INVALID_ARG, "negative matrix shape");
  if (rows > 0 && (!x || !out)) throw StatusError(MRK_ERR_INVALID_ARG, "null matrix / output");
  int used = 0;
  for (auto &t : model->forest.trees)
    for (auto f : t.feat) used = std::max(used, f + 1);
  if (rows > 0 && cols < used)
    throw StatusError(MRK_ERR_DIM_MISMATCH, "matrix has " + std::to_string(cols) + " columns but the booster splits on feature " +
                                                std::to_string(used - 1));
}

int mrk_model_predict_f64(mrk_model *model, const double *rowmajor, int rows, int cols, double *out_scores) {
  return guard([&] {
    check_predict_args(model, rowmajor, rows, cols, out_scores);
    if (rows == 0) return;
    mrk_ctx *ctx = model->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    MRK_HIP(hipSetDevice(ctx->device));
    const size_t xb = (size_t)rows * cols * sizeof(double);
    ctx->d_x.reserve(xb ? xb : 16);
    ctx->d_out.reserve((size_t)rows * sizeof(double));
    if (xb) MRK_HIP(hipMemcpyAsync(ctx->d_x.p, rowmajor, xb, hipMemcpyHostToDevice, ctx->stream));
    MRK_HIP(hipMemsetAsync(ctx->d_flag.p, 0, sizeof(int), ctx->stream));
    launch_score(ctx, model, ctx->d_x.as<double>(), rows, cols, ctx->d_out.as<double>(), ctx->d_flag.as<int>());
    MRK_HIP(hipMemcpyAsync(out_scores, ctx->d_out.p, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MRK_HIP(hipMemcpyAsync(ctx->h_flag.p, ctx->d_flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MRK_HIP(hipStreamSynchronize(ctx->stream));
    drain_profile_events(ctx);
    if (*ctx->h_flag.as<int>() & 1)
      throw StatusError(MRK_ERR_INVALID_ARG,
                        "Input data contains `inf` or a value too large, while `missing` is not set to `inf`");
  });
}

int mrk_model_predict_device(mrk_model *model, const double *d_rowmajor, int rows, int cols, double *d_out_scores) {
  return guard([&] {
    check_predict_args(model, d_rowmajor, rows, cols, d_out_scores);
    if (rows == 0) return;
    mrk_ctx *ctx = model->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    MRK_HIP(hipSetDevice(ctx->device));
    launch_score(ctx, model, d_rowmajor, rows, cols, d_out_scores, ctx->d_flag.as<int>());
  });
}

int mrk_model_get_info(mrk_model *model, mrk_model_info *out) {
  return guard([&] {
    if (!model || !out) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
    const Forest &f = model->forest;
    out->backend = (int)f.backend;
    out->n_trees = (int)f.trees.size();
    out->max_depth = f.max_depth();
    out->n_features = f.n_features;
    out->is_f64 = f.backend == Backend::LightGBM ? 1 : 0;
    out->n_categorical = (int)f.n_categorical();
    out->n_nodes = f.n_nodes();
    out->n_leaves = f.n_leaves();
    out->device_bytes = (int64_t)(model->packed.image.size() + model->packed.trees.size() * sizeof(TreeRef) +
                                  model->packed.chunks.size() * sizeof(ChunkRef) + f.cat_bits.size() * 4);
    out->base_score = f.base_score;
    out->bitvector = model->qs.ok ? 1 : 0;
    out->tile_columns = model->qs.ok ? (int32_t)model->qs.views.size() : 0;
  });
}

static void importance_checked(const Forest &f, int type, double *out, int n_cols) {
  if (!out) throw StatusError(MRK_ERR_INVALID_ARG, "out is null");
  if (type < MRK_IMPORTANCE_SPLIT || type > MRK_IMPORTANCE_TOTAL_GAIN) throw StatusError(MRK_ERR_INVALID_ARG, "unknown importance type " + std::to_string(type));
  if (n_cols < f.n_features)  // the reference indexes w(offset) for every descriptor column: a shorter array would be an IndexOutOfBounds there
    throw StatusError(MRK_ERR_DIM_MISMATCH, "weights: the model knows " + std::to_string(f.n_features) + " features, the caller's descriptor has " + std::to_string(n_cols) + " columns");
  f.feature_importance(type, out, n_cols);
}

// == Booster.weights() (ltrlib), reference call site ml/rank/LambdaMARTRanker.scala:391-406
int mrk_model_weights(mrk_model *model, int importance_type, double *out, int n_cols) {
  return guard([&] {
    if (!model) throw StatusError(MRK_ERR_INVALID_ARG, "null model");
    importance_checked(model->forest, importance_type, out, n_cols);
  });
}

int mrk_model_inspect_weights(int backend, const uint8_t *bytes, size_t len, int importance_type, double *out, int n_cols) {
  return guard([&] {
    if (!bytes && len) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
    Forest f;
    if (backend == MRK_BACKEND_LIGHTGBM) f = parse_lightgbm_text((const char *)bytes, len);
    else if (backend == MRK_BACKEND_XGBOOST) f = parse_xgboost(bytes, len);
    else throw StatusError(MRK_ERR_INVALID_ARG, "unsupported booster tag " + std::to_string(backend));
    importance_checked(f, importance_type, out, n_cols);
  });
}

// sizeof / offsetof of every struct that crosses the boundary, in the order mrk.h lists them
int mrk_abi_layout(int32_t *out, int cap) {
  const int32_t v[] = {
      MRK_ABI_VERSION,
      (int32_t)sizeof(mrk_field), (int32_t)offsetof(mrk_field, name), (int32_t)offsetof(mrk_field, type), (int32_t)offsetof(mrk_field, n),
      (int32_t)offsetof(mrk_field, num), (int32_t)offsetof(mrk_field, str), (int32_t)offsetof(mrk_field, strs), (int32_t)offsetof(mrk_field, nums),
      (int32_t)sizeof(mrk_request), (int32_t)offsetof(mrk_request, id), (int32_t)offsetof(mrk_request, timestamp_ms), (int32_t)offsetof(mrk_request, user),
      (int32_t)offsetof(mrk_request, session), (int32_t)offsetof(mrk_request, fields), (int32_t)offsetof(mrk_request, n_fields),
      (int32_t)offsetof(mrk_request, n_items), (int32_t)offsetof(mrk_request, item_ids), (int32_t)offsetof(mrk_request, item_field_offsets),
      (int32_t)offsetof(mrk_request, item_fields),
      (int32_t)sizeof(mrk_model_info), (int32_t)offsetof(mrk_model_info, backend), (int32_t)offsetof(mrk_model_info, n_trees),
      (int32_t)offsetof(mrk_model_info, max_depth), (int32_t)offsetof(mrk_model_info, n_features), (int32_t)offsetof(mrk_model_info, is_f64),
      (int32_t)offsetof(mrk_model_info, n_categorical), (int32_t)offsetof(mrk_model_info, n_nodes), (int32_t)offsetof(mrk_model_info, n_leaves),
      (int32_t)offsetof(mrk_model_info, device_bytes), (int32_t)offsetof(mrk_model_info, base_score), (int32_t)offsetof(mrk_model_info, bitvector),
      (int32_t)offsetof(mrk_model_info, tile_columns),
  };
  const int n = (int)(sizeof(v) / sizeof(v[0]));
  if (out)
    for (int i = 0; i < n && i < cap; ++i) out[i] = v[i];
  return n;
}

// Host-only: parse + validate + pack a booster exactly as mrk_model_load does, without a device (a config check, CPU tests)
int mrk_model_inspect(int backend, const uint8_t *bytes, size_t len, mrk_model_info *out) {
  return guard([&] {
    if (!out || (!bytes && len)) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
    Forest f;
    if (backend == MRK_BACKEND_LIGHTGBM) f = parse_lightgbm_text((const char *)bytes, len);
    else if (backend == MRK_BACKEND_XGBOOST) f = parse_xgboost(bytes, len);
    else throw StatusError(MRK_ERR_INVALID_ARG, "unsupported booster tag " + std::to_string(backend));
    if (f.average_output) throw StatusError(MRK_ERR_UNSUPPORTED, "lightgbm: average_output (random forest) models are not supported");
    const PackedForest packed = pack_forest(f, score_chunk_budget());
    const PackedForestQS qs = pack_forest_qs(f, f.n_features);
    out->backend = (int)f.backend;
    out->n_trees = (int)f.trees.size();
    out->max_depth = f.max_depth();
    out->n_features = f.n_features;
    out->is_f64 = f.backend == Backend::LightGBM ? 1 : 0;
    out->n_categorical = (int)f.n_categorical();
    out->n_nodes = f.n_nodes();
    out->n_leaves = f.n_leaves();
    out->device_bytes = (int64_t)(packed.image.size() + packed.trees.size() * sizeof(TreeRef) + packed.chunks.size() * sizeof(ChunkRef) + f.cat_bits.size() * 4);
    out->base_score = f.base_score;
    out->bitvector = qs.ok ? 1 : 0;
    out->tile_columns = qs.ok ? (int32_t)qs.views.size() : 0;
  });
}

void mrk_model_retain(mrk_model *model) {
  if (model) model->refs.fetch_add(1);
}

void mrk_model_free(mrk_model *model) {
  if (!model) return;
  if (model->refs.fetch_sub(1) == 1) {
    mrk_ctx *ctx = model->ctx;
    if (ctx) {
      {
        std::lock_guard<std::mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        delete model;
      }
      ctx_release(ctx);
    } else {
      delete model;
    }
  }
}

int mrk_sync(mrk_ctx *ctx) {
  return guard([&] {
    if (!ctx) throw StatusError(MRK_ERR_INVALID_ARG, "null context");
    MRK_HIP(hipSetDevice(ctx->device));
    MRK_HIP(hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> lk(ctx->mu);
    drain_profile_events(ctx);
  });
}

void *mrk_stream(mrk_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int mrk_profile_enable(mrk_ctx *ctx, int on) {
  return guard([&] {
    if (!ctx) throw StatusError(MRK_ERR_INVALID_ARG, "null context");
    std::lock_guard<std::mutex> lk(ctx->mu);
    drain_profile_events(ctx);
    ctx->profile = on != 0;
    ctx->timers.clear();
  });
}

int mrk_profile_get(mrk_ctx *ctx, const char *kernel, double *total_ms, int64_t *launches) {
  return guard([&] {
    if (!ctx || !kernel) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    drain_profile_events(ctx);
    auto it = ctx->timers.find(kernel);
    if (total_ms) *total_ms = it == ctx->timers.end() ? 0.0 : it->second.total_ms;
    if (launches) *launches = it == ctx->timers.end() ? 0 : it->second.launches;
  });
}

}  // extern "C"
