/*
 * mrk.h — C ABI of libmrk_hip.so, the MI355X-native /rank hot path for Metarank.
 *
 * This is the drop-in boundary.  The Scala host keeps its HTTP API,
 * event schema, feature registry and persistence; it binds these entry points through
 * JNA / Panama and nothing else.  Plain C: opaque handles,
 * caller-owned buffers, no exceptions, no C++/torch types in any signature.
 *
 * Every function returns MRK_OK (0) on success and a negative mrk_status on failure;
 * mrk_last_error() returns a thread-local human readable message for the last failure
 * on the calling thread.  Handles are thread-safe: many request threads may call
 * mrk_model_predict_f64 / mrk_rank on the same handles concurrently while a feedback
 * thread calls mrk_store_put_* (reference threading model: cats-effect compute pool,
 * src/main/scala/ai/metarank/ml/rank/LambdaMARTRanker.scala:348 and
 * src/main/scala/ai/metarank/fstore/cache/CachedModelStore.scala:39-42).
 *
 * Reference interfaces replaced (all paths relative to the reference repo root,
 * M = src/main/scala/ai/metarank):
 *
 *   mrk_model_load*            <- LightGBMBooster(bytes) / XGBoostBooster(bytes)
 *                                 M/ml/rank/LambdaMARTRanker.scala:228-232 and the
 *                                 bitstream reader :192-236
 *   mrk_model_predict_f64      <- ltrlib Booster.predictMat(values, rows, cols)
 *                                 M/ml/rank/LambdaMARTRanker.scala:348
 *   mrk_model_free             <- Booster.close()/isClosed() :361-365
 *   mrk_config_load_json       <- FeatureMapping.fromFeatureSchema + makeDatasetDescriptor
 *                                 M/FeatureMapping.scala:56-99
 *   mrk_store_put_*            <- KVStore[Key,FeatureValue].put  M/fstore/Persistence.scala:85-89
 *                                 fed by FeatureValueSink.write   M/flow/FeatureValueSink.scala:10-14
 *   mrk_rank, mrk_rank_binary, <- Ranker.rerank = makeQuery + predict + sortBy(-score)
 *   mrk_batch_*                   M/ml/Ranker.scala:27-83,97-106
 *   mrk_model_weights          <- ltrlib Booster.weights()  M/ml/rank/LambdaMARTRanker.scala:391-406
 *   mrk_init(devices, n)       <- HipConfig(inner, devices: List[Int]) next to LightGBMConfig / XGBoostConfig
 *                                 M/config/BoosterConfig.scala:96-104 (SURVEY.md 8b touch point 1): one context per device,
 *                                 all inside the one host process (M/main/command/Serve.scala:72-128)
 *   mrk_encoder_*              <- OnnxSession / OnnxBiEncoder / OnnxCrossEncoder  M/ml/onnx/sbert/ (OnnxSession, OnnxBiEncoder, OnnxCrossEncoder .scala)
 *
 * ABI 8 (round 5): mrk_init creates n contexts; mrk_device_count; mrk_comm_init_local; mrk_model_inspect; mrk_serve_stats takes
 * the length of its output array; mrk_store_put_binary_at / mrk_store_expire; mrk_encoder_load is f32 (MRK_ENCODER_AUTO = F32).
 * ABI 9 (round 6): mrk_model_weights / mrk_model_inspect_weights (Booster.weights()), mrk_abi_layout.  Same ABI, new behaviour:
 * the serving queue's slots are launched in gangs (64 slots on 8 streams) and mrk_rank answers through a started queue.
 */
#ifndef MRK_H
#define MRK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRK_ABI_VERSION 9

typedef enum mrk_status {
  MRK_OK = 0,
  MRK_ERR_INVALID_ARG = -1,   /* null handle, bad sizes, unknown enum                    */
  MRK_ERR_PARSE = -2,         /* model / config bytes could not be parsed                 */
  MRK_ERR_DEVICE = -3,        /* HIP runtime failure (message carries hipGetErrorString)  */
  MRK_ERR_DIM_MISMATCH = -4,  /* IllegalStateException "dim mismatch" M/model/ItemValue.scala:47,56,60 */
  MRK_ERR_ARITHMETIC = -5,    /* java.lang.ArithmeticException: / by zero in the normalised rate,
                                 M/feature/RateFeature.scala:346-348 (global top counter == 0)  */
  MRK_ERR_UNSUPPORTED = -6,   /* valid input the device path does not implement (yet)     */
  MRK_ERR_NOT_FOUND = -7,     /* unknown model / feature name                             */
  MRK_ERR_FEATURE_MISMATCH = -8 /* "booster trained with X features, but config defines Y"
                                 M/ml/rank/LambdaMARTRanker.scala:208-213                   */
} mrk_status;

typedef struct mrk_ctx mrk_ctx;       /* one per process per device set           */
typedef struct mrk_model mrk_model;   /* one per loaded booster (ref-counted)     */
typedef struct mrk_batch mrk_batch;   /* a prepared, device-resident request batch */

/* ---------------------------------------------------------------- lifecycle */

int mrk_abi_version(void);
/* identifies the library's SOURCES (hex digest over csrc/ and this header, fixed when the library is built) */
const char *mrk_build_id(void);
const char *mrk_last_error(void);
/* Layout of every struct that crosses the boundary by value or by pointer, as THIS library was compiled: what a JNA
 * @FieldOrder Structure checks at start-up instead of trusting its own copy of this header.  Writes up to
 * `cap` int32 into out (nullable) and returns how many there are:
 *   [0] MRK_ABI_VERSION
 *   [1..8]   mrk_field:      sizeof, then offsetof name, type, n, num, str, strs, nums
 *   [9..19]  mrk_request:    sizeof, then offsetof id, timestamp_ms, user, session, fields, n_fields, n_items, item_ids,
 *                            item_field_offsets, item_fields
 *   [20..32] mrk_model_info: sizeof, then offsetof backend, n_trees, max_depth, n_features, is_f64, n_categorical, n_nodes,
 *                            n_leaves, device_bytes, base_score, bitvector, tile_columns */
int mrk_abi_layout(int32_t *out, int cap);

/* One context per listed HIP ordinal, all in the calling process: out[0 .. n_devices) (HipConfig(inner, devices: List[Int]),
 * SURVEY 8b touch point 1; the reference's host is ONE JVM, M/main/command/Serve.scala:72-128).  A context owns its streams,
 * its replica of the feature store, its models and batches; every entry point selects its context's device for the calling
 * thread, nothing is per-process, so contexts are driven concurrently from different host threads (one thread per context
 * for the serving loop; puts may come from any thread).  The same ordinal may appear more than once (independent contexts
 * sharing a GPU).  n_devices = 1 is the single-device case.  On failure nothing is created.  Contexts of one process are
 * joined into an RCCL communicator by mrk_comm_init_local, contexts of different processes by mrk_comm_init. */
int mrk_init(const int *device_ids, int n_devices, mrk_ctx **out);
int mrk_device_count(void); /* HIP devices visible to this process (0: none / no driver): what HipConfig.devices is validated against */
void mrk_shutdown(mrk_ctx *ctx);

/* ------------------------------------------------- model (replaces Booster) */

enum { MRK_BACKEND_LIGHTGBM = 0, MRK_BACKEND_XGBOOST = 1 }; /* = boosterTag, LambdaMARTRanker.scala:229-230 */

/* Inner booster bytes: LightGBM model string, or XGBoost JSON / UBJSON.  Bytes are copied. */
int mrk_model_load(mrk_ctx *ctx, int backend, const uint8_t *bytes, size_t len, mrk_model **out);

/* Metarank model container (bitstream v2/v3, LambdaMARTRanker.scala:367-389).
 * If feature_names != NULL the stored list must equal it in order, else MRK_ERR_FEATURE_MISMATCH. */
int mrk_model_load_container(mrk_ctx *ctx, const uint8_t *blob, size_t len,
                             const char *const *feature_names, int n_features, mrk_model **out);

/* == predictMat: rowmajor is rows*cols f64 (host memory), out_scores is rows f64 (host memory). */
int mrk_model_predict_f64(mrk_model *model, const double *rowmajor, int rows, int cols,
                          double *out_scores);
/* Same with all three buffers already in device memory of ctx's device (no copies, async on the
 * context stream; call mrk_sync before reading). */
int mrk_model_predict_device(mrk_model *model, const double *d_rowmajor, int rows, int cols,
                             double *d_out_scores);

typedef struct mrk_model_info {
  int32_t backend;       /* MRK_BACKEND_*                                   */
  int32_t n_trees;
  int32_t max_depth;     /* longest root->leaf path in node visits          */
  int32_t n_features;    /* max_feature_idx + 1 / num_feature               */
  int32_t is_f64;        /* 1: LightGBM f64 arithmetic, 0: XGBoost f32       */
  int32_t n_categorical; /* number of categorical split nodes               */
  int64_t n_nodes;       /* internal nodes                                  */
  int64_t n_leaves;
  int64_t device_bytes;  /* packed forest bytes resident in HBM             */
  double base_score;     /* XGBoost base margin (0.0 for LightGBM)          */
  int32_t bitvector;     /* 1: every tree has <= 16 leaves, the bit-vector scorer applies; 0: tree-walk scorer */
  int32_t tile_columns;  /* bitvector: columns ("views") of the scorer's binned tile */
} mrk_model_info;
int mrk_model_get_info(mrk_model *model, mrk_model_info *out);
/* Host-only (no context, no device): reads, validates and packs a booster exactly as mrk_model_load would and reports what it
 * found (device_bytes = what the load would place in HBM) - MRK_ERR_PARSE for malformed bytes, MRK_ERR_UNSUPPORTED for a
 * well-formed model the scorer does not implement (dart, multi-output, vector leaves, num_parallel_tree > 1, a non-identity
 * objective, linear trees, random-forest averaging).  What a host calls when it validates a config. */
int mrk_model_inspect(int backend, const uint8_t *bytes, size_t len, mrk_model_info *out);

/* == ltrlib Booster.weights(): Array[Double], one entry per matrix column (reference call site
 * M/ml/rank/LambdaMARTRanker.scala:391-406: `w(offset)` / `w.slice(offset, offset + size)` per descriptor feature).  out[0 .. n_cols):
 * n_cols = the DatasetDescriptor's dimension; columns past the model's n_features get 0.0; n_cols < n_features is
 * MRK_ERR_DIM_MISMATCH (the JVM would throw IndexOutOfBounds).  Computed on the host from the parsed booster with the owning
 * library's own arithmetic:
 *   LightGBM  (LGBM_BoosterFeatureImportance, GBDT::FeatureImportance, num_iteration 0): SPLIT = number of splits on the feature
 *             with split_gain > 0; GAIN = TOTAL_GAIN = sum of those float gains accumulated in double, tree order then node order.
 *   XGBoost   (Booster.getScore, GBTree::FeatureScore): SPLIT = "weight" = number of splits; TOTAL_GAIN = "total_gain" = sum of
 *             loss_chg accumulated in float; GAIN = "gain" = total_gain / weight in float; a feature never split on is absent
 *             from the JVM's map - 0.0 here.
 * ASSUMPTION (ltrlib 0.2.6 is not in the reference tree): its LightGBMBooster.weights() asks lightgbm4j for
 * FeatureImportanceType.GAIN and its XGBoostBooster.weights() fills an array from getScore("", "gain"): a HipBooster passes
 * MRK_IMPORTANCE_GAIN for both.  Unverifiable here; if ltrlib uses another type the binding changes one constant.  A model file
 * without split gains (no `split_gain=` / "loss_changes") answers GAIN / TOTAL_GAIN with MRK_ERR_UNSUPPORTED. */
enum { MRK_IMPORTANCE_SPLIT = 0, MRK_IMPORTANCE_GAIN = 1, MRK_IMPORTANCE_TOTAL_GAIN = 2 };
int mrk_model_weights(mrk_model *model, int importance_type, double *out, int n_cols);
/* the same host-only, from booster bytes (no context, no device): config validation, CPU tests */
int mrk_model_inspect_weights(int backend, const uint8_t *bytes, size_t len, int importance_type, double *out, int n_cols);

void mrk_model_retain(mrk_model *model);
void mrk_model_free(mrk_model *model); /* drops one reference; idempotent at zero (close()/isClosed()) */

/* ------------------------------------ feature schema (FeatureMapping mirror) */

/* json: {"features":[<Metarank feature schemas>], "models": {"<name>": {"type":"lambdamart",
 * "features":[...]}}} i.e. the `features:` and `models:` sections of Metarank's config.yml
 * rendered as JSON (the Scala host has circe encoders for every schema).  Builds the feature
 * registry, the device store layout and, per lambdamart model, the DatasetDescriptor
 * (column order = models.<name>.features order, M/FeatureMapping.scala:66-72,89-99). */
int mrk_config_load_json(mrk_ctx *ctx, const char *json, size_t len);

/* number of matrix columns (DatasetDescriptor.dim) for a configured model, <0 on error */
int mrk_model_dim(mrk_ctx *ctx, const char *model_name);
/* Host-only (no device, no context): what the library builds the first time a model of this config is ranked - the
 * assembly kernel specialised for the model's feature list (the reference fixes that list per model when the config is
 * loaded, M/FeatureMapping.scala:56-99; here it becomes compile-time constants of the kernel).  f64: scorer precision the
 * kernel bins for (1 LightGBM, 0 XGBoost).  what = 0: the HIP source handed to hiprtc; what = 1: the gfx950 code object -
 * of all specialised kernels; `what | (k << 8)`, k = 1..4: of the one kernel the library would compile by itself (1 the
 * workgroup-per-request kernel of full batches, 2 its op-split / sliced form, 3 its f64-matrix form, 4 the item-parallel kernel).
 * MRK_ERR_INVALID_ARG with *needed set (for what = 1: to an upper bound) when `out` is NULL or `cap` too small; *needed is
 * the exact size on success. */
int mrk_config_specialize(const char *json, size_t len, const char *model_name, int f64, int what, uint8_t *out, size_t cap,
                          size_t *needed);

/* Host-only: writes into directory `dir` the gfx950 code object of every specialised kernel in `kernel_mask` (bit k: 0 the
 * workgroup-per-request kernel, 1 its op-split / sliced form, 2 the f64-matrix form, 3 the item-parallel kernel, 4 the
 * one-launch kernel of mrk_rank, 5 the persistent workgroup of the serving queue, 6 the one-launch kernel of FULL batches
 * behind MRK_RANK_FUSED_SCORE=1, 7 the stand-alone pre-pass of requests too large for one workgroup) for this config's model - what a
 * deployment ships next to libmrk_hip.so (directory `jit_cache`) so that no process ever compiles: the library looks
 * there, then in the user's cache ($MRK_JIT_CACHE_DIR, ~/.cache/mrk_jit), and only then compiles - in the BACKGROUND,
 * ranking with the kernel that interprets the program meanwhile (MRK_RANK_JIT: 0 never specialise, 1 wait for the compiler,
 * require, async, auto = this default).  out_compiled (nullable): how many were not there yet. */
int mrk_config_precompile(const char *json, size_t len, const char *model_name, int f64, unsigned kernel_mask, const char *dir,
                          int *out_compiled);
/* The kernels that write the scorer's binned tile are keyed by the FOREST too - by its view signature: per matrix column
 * which tile columns it fills (NaN / zero / categorical routing kinds) and how many 128-entry chunks its threshold table
 * takes; not the thresholds, not the trees - and hold it as compile-time constants (no descriptor loads, no view loops).
 * Retraining on the same features usually keeps the signature, hence the kernels.  The two entry points above know the
 * config only: they build the signature-less kernels, which serve any model of the program while its own kernels compile.
 * These two do the same for one serialised booster (backend / bytes as for mrk_model_load; the precision follows from the
 * backend): what a deployment runs when it ships a model.  Host-only. */
int mrk_config_specialize_for_model(const char *json, size_t len, const char *model_name, int backend, const uint8_t *model_bytes,
                                    size_t model_len, int what, uint8_t *out, size_t cap, size_t *needed);
int mrk_config_precompile_for_model(const char *json, size_t len, const char *model_name, int backend, const uint8_t *model_bytes,
                                    size_t model_len, unsigned kernel_mask, const char *dir, int *out_compiled);
/* Which specialised kernels of this model are loaded, as text: one line "<kernel name> <key> program|program+forest" each;
 * <key> = the stem of the kernel's cache file (hash and size of its translation unit, compiler version).  With mrk_build_id
 * this is what a measurement records to say WHICH code it measured (bench.py `provenance`).  MRK_ERR_INVALID_ARG with *needed
 * set (bytes incl. the terminating 0) when `out` is NULL or `cap` too small. */
int mrk_config_kernel_keys(mrk_ctx *ctx, const char *model_name, char *out, size_t cap, size_t *needed);
/* Serve.maybeWarmup for the kernels: waits until the background compiles of this model's kernels that are under way have
 * finished (the next launch uses them).  A host calls it before it opens its port; nothing on the request path waits. */
int mrk_config_warmup(mrk_ctx *ctx, const char *model_name);

/* ------------------------- feature store (KVStore[Key, FeatureValue] mirror) */

/* `key` is Key.encode (M/model/Key.scala:9): "<ScopeCodec.encode(scope)>/<feature name>", e.g.
 * "item=42/popularity", "global/ctr_click_norm", "field=genre:drama/ctr_genre_click",
 * "session=s1/profile_interactions", "irf=query:socks:p1/ctr_click".
 * Each put replaces the previous value of that key (KVStore.put semantics).  Puts are staged on
 * the host and become visible to mrk_rank calls that start after mrk_store_flush returns
 * (mrk_rank / mrk_batch_prepare flush implicitly). */
int mrk_store_put_double(mrk_ctx *ctx, const char *key, double v);               /* ScalarValue(SDouble)      */
int mrk_store_put_bool(mrk_ctx *ctx, const char *key, int v);                    /* ScalarValue(SBoolean)     */
int mrk_store_put_string(mrk_ctx *ctx, const char *key, const char *v);          /* ScalarValue(SString)      */
int mrk_store_put_string_list(mrk_ctx *ctx, const char *key, const char *const *v, int n); /* SStringList    */
int mrk_store_put_double_list(mrk_ctx *ctx, const char *key, const double *v, int n);      /* SDoubleList    */
int mrk_store_put_counter(mrk_ctx *ctx, const char *key, int64_t v);             /* CounterValue             */
int mrk_store_put_periodic(mrk_ctx *ctx, const char *key, const int64_t *values, int n); /* PeriodicCounterValue.values(i).value */
int mrk_store_put_bounded_list(mrk_ctx *ctx, const char *key, const char *const *values, int n); /* BoundedListValue of SString */
int mrk_store_delete(mrk_ctx *ctx, const char *key);

/* Bulk load from the reference's binary wire format: `bytes` is any concatenation of
 * FeatureValueCodec records (M/fstore/codec/impl/FeatureValueCodec.scala:40-168, tags 0-13) - what the
 * Redis / RocksDB / MapDB `values` store holds per key.  Every record is applied like the matching
 * mrk_store_put_* (NumStats / Map / Frequency values are parsed and ignored: /rank does not read them).
 * out_records (nullable) receives the number of records decoded. */
int mrk_store_put_binary(mrk_ctx *ctx, const uint8_t *bytes, size_t len, int *out_records);
/* FeatureValue.expire (FeatureValueCodec.scala:42-48,75: every record carries its feature's ttl; the Redis store drops a key that
 * long after its LAST write, RedisKVStore.scala:40 - the other engines never do).  mrk_store_put_binary ignores it.  A host whose
 * system of record expires keys uses mrk_store_put_binary_at instead: the same load, and every applied record is remembered
 * with deadline = now_ms + expire (now_ms: the host's clock, Timestamp.now; pre-ttl encodings count as 90 days); a later write
 * of the same key through ANY put replaces or clears its deadline.  mrk_store_expire(ctx, now_ms, &n) - called from a timer -
 * deletes the values whose deadline has passed (exactly mrk_store_delete; the next flush uploads the change).  Host memory:
 * about 40 bytes per live deadline; nothing when the _at form is never used. */
int mrk_store_put_binary_at(mrk_ctx *ctx, const uint8_t *bytes, size_t len, int64_t now_ms, int *out_records);
int mrk_store_expire(mrk_ctx *ctx, int64_t now_ms, int64_t *out_expired);

/* Write path: instead of a refreshed FeatureValue the host may forward the raw Writes of
 * FeatureValueFlow.commitWrite (M/flow/FeatureValueFlow.scala:44-62); the FeatureValue the read path needs is
 * then derived by the library - always fresh, no `refresh` interval.
 *   mrk_store_increment_periodic  Write.PeriodicIncrement(key, ts, inc): added to the key's bucket ring in HBM
 *                                 (bucket = ts.toStartOfPeriod(bucket)); the window sums of
 *                                 PeriodicCounterFeature.fromMap (M/model/Feature.scala:142-161) are recomputed on
 *                                 the device at the next flush.  A key is fed either by these or by
 *                                 mrk_store_put_periodic, not both (the ring wins).
 *   mrk_store_increment           Write.Increment(key, ts, inc) -> CounterValue
 *   mrk_store_append              Write.Append(key, SString(value), ts) -> BoundedListValue, bounded by the
 *                                 feature's count / duration (M/fstore/memory/MemBoundedList.scala:18-37) */
int mrk_store_increment_periodic(mrk_ctx *ctx, const char *key, int64_t ts_ms, int64_t inc);
/* n PeriodicIncrements in one call (one event usually produces several: item, field-scoped, global keys) */
int mrk_store_increment_periodic_batch(mrk_ctx *ctx, const char *const *keys, const int64_t *ts_ms, const int64_t *inc, int n);
int mrk_store_increment(mrk_ctx *ctx, const char *key, int64_t inc);
int mrk_store_append(mrk_ctx *ctx, const char *key, const char *value, int64_t ts_ms);
int mrk_store_flush(mrk_ctx *ctx);

/* ---------------------------------------------------------------- requests */

enum {
  MRK_FIELD_STRING = 0,      /* Field.StringField     */
  MRK_FIELD_NUMBER = 1,      /* Field.NumberField     */
  MRK_FIELD_BOOL = 2,        /* Field.BooleanField    */
  MRK_FIELD_STRING_LIST = 3, /* Field.StringListField */
  MRK_FIELD_NUMBER_LIST = 4  /* Field.NumberListField */
};

typedef struct mrk_field {
  const char *name;
  int32_t type;               /* MRK_FIELD_*                                  */
  int32_t n;                  /* list length for the *_LIST types              */
  double num;                 /* NUMBER / BOOL (0|1)                           */
  const char *str;            /* STRING                                        */
  const char *const *strs;    /* STRING_LIST                                   */
  const double *nums;         /* NUMBER_LIST                                   */
} mrk_field;

/* RankingEvent (M/model/Event.scala) restricted to what the hot path reads. */
typedef struct mrk_request {
  const char *id;             /* RankingEvent.id                                */
  int64_t timestamp_ms;       /* RankingEvent.timestamp.ts                      */
  const char *user;           /* nullable                                       */
  const char *session;        /* nullable                                       */
  const mrk_field *fields;    /* ranking-level fields                           */
  int32_t n_fields;
  int32_t n_items;
  const char *const *item_ids;        /* n_items item ids, request order        */
  const int32_t *item_field_offsets;  /* nullable; CSR n_items+1 offsets into item_fields */
  const mrk_field *item_fields;       /* per-item fields (relevancy, overrides) */
} mrk_request;

/* Ranker.rerank for one request.  out_scores[n_items]: score of item i (request order);
 * out_order[n_items]: indices into the request in response order (stable sort by -score with
 * java.lang.Double.compare semantics); out_matrix: nullable, n_items*dim row-major f64 = the
 * ClickthroughQuery matrix (explain / parity). */
int mrk_rank(mrk_ctx *ctx, mrk_model *model, const char *model_name, const mrk_request *req,
             double *out_scores, int32_t *out_order, double *out_matrix);

/* The same for a request in the reference's binary RankingEventFormat (M/util/RankingEventFormat.scala:12-62; the
 * encoding of the warm-up requests inside the model container, and a cheaper wire form than JSON for a host that
 * already holds the event).  `event` holds one record; out_n_items receives its item count; out_scores /
 * out_order must have room for `capacity` items (MRK_ERR_INVALID_ARG with out_n_items set if it has more). */
int mrk_rank_binary(mrk_ctx *ctx, mrk_model *model, const char *model_name, const uint8_t *event, size_t len,
                    int *out_n_items, double *out_scores, int32_t *out_order, int capacity);
/* Warm-up (Serve.maybeWarmup, M/main/command/Serve.scala:130-150): a model loaded with mrk_model_load_container keeps
 * the container's warm-up requests; mrk_model_warmup ranks each of them once (results discarded) and returns the
 * number replayed through out_replayed. */
int mrk_model_warmup(mrk_ctx *ctx, mrk_model *model, const char *model_name, int *out_replayed);

/* Batched form: resolve n_req requests once into a device-resident batch, then run it any number
 * of times (the benchmark's device-only timed region is mrk_batch_run only). */
int mrk_batch_prepare(mrk_ctx *ctx, const char *model_name, const mrk_request *reqs, int n_req,
                      mrk_batch **out);

/* ---- the serving loop: reusable batches, flat item ids, results in pinned memory ----------------------------------
 * What a host that decodes `ranking` events off the wire has in its hands is the UTF-8 bytes of the item ids, not an
 * array of C strings - and the first hop of the read path (FeatureValueLoader.fromStateBackend building one
 * Key(ItemScope(id), feature) per candidate, M/fstore/FeatureValueLoader.scala:11-25, M/model/Key.scala:7-10) is an id
 * lookup per candidate.  With flat ids that hop runs on the device: the id bytes are uploaded as they are and a kernel
 * hashes, probes the device mirror of the store's id table and compares (csrc/resolve.hip); the host does O(requests)
 * work, nothing per item.
 *
 *   mrk_batch_create   an empty batch that owns a HIP stream and grow-only buffers;
 *   mrk_batch_load     (re)fills it: waits for the batch's previous work, resolves the request-level part on the host
 *                      (user / session / ranking slots, request constants, table sizing from bounds), uploads, and - with
 *                      `ids` - enqueues the id resolution.  ids == NULL: reqs[r].item_ids are used (host lookups, exact
 *                      table sizes), exactly like mrk_batch_prepare.  With ids, reqs[r].item_ids is ignored: the ids of
 *                      request 0's items, then request 1's, ... lie back to back in ids->bytes, item i of the batch
 *                      being bytes[offsets[i], offsets[i + 1]).  Buffers obtained from mrk_host_alloc are read by the
 *                      copy engine directly; any other memory is staged through the batch's pinned buffer.
 *   mrk_batch_run      as before (asynchronous);
 *   mrk_batch_enqueue_fetch  enqueues the download of scores / order / status into the batch's pinned result buffer
 *                      behind the run (asynchronous);
 *   mrk_batch_host_outputs   waits for the batch and returns pointers INTO that pinned buffer (valid until the next
 *                      mrk_batch_load / mrk_batch_run of this batch): scores[total_items] (request order), order
 *                      [total_items] (request-local indices in response order), status[n_req] (mrk_status per request).
 * Several batches of a context may be loaded, run and fetched from different host threads at the same time; store puts
 * are serialised against them (a put never changes what a batch already in flight reads). */
typedef struct mrk_item_ids {
  const uint8_t *bytes;     /* concatenated UTF-8 ids, no terminators                  */
  const uint32_t *offsets;  /* total_items + 1 byte offsets into `bytes`, ascending    */
  size_t bytes_len;         /* size of `bytes`: offsets[total_items] must not exceed it.  Offsets come off the wire: a
                               batch whose last offset passes bytes_len is refused (MRK_ERR_INVALID_ARG); an item whose
                               offsets descend or pass bytes_len fails ITS request with MRK_ERR_INVALID_ARG (checked by
                               the id-resolution kernel, and by the host wherever it reads an id itself)            */
} mrk_item_ids;
/* LIFETIME of ids->bytes / ids->offsets: pinned buffers (mrk_host_alloc) are read by the copy engine AFTER mrk_batch_load
 * has returned - do not rewrite or free them before the next mrk_batch_sync / mrk_batch_host_outputs / mrk_batch_fetch of
 * this batch.  Pageable memory is copied before mrk_batch_load returns and may be reused at once. */
int mrk_batch_create(mrk_ctx *ctx, mrk_batch **out);
int mrk_batch_load(mrk_batch *batch, const char *model_name, const mrk_request *reqs, int n_req, const mrk_item_ids *ids);
int mrk_batch_enqueue_fetch(mrk_batch *batch);
int mrk_batch_host_outputs(mrk_batch *batch, const double **scores, const int32_t **order, const int32_t **status);
/* pinned (page-locked) host memory the device can read / write without a staging copy; NULL on failure */
void *mrk_host_alloc(size_t bytes);
void mrk_host_free(void *p);
int mrk_batch_total_items(mrk_batch *batch);
/* asynchronous on the context stream */
int mrk_batch_run(mrk_batch *batch, mrk_model *model);
/* Item-sharded execution (the reference has no counterpart - it scores a request on one
 * JVM thread): shard `shard_index` of `shard_count` assembles and scores batch items
 * [index * chunk, (index + 1) * chunk) only, chunk = mrk_batch_shard_chunk() (total / count rounded up to a
 * whole number of 128-item scorer tiles), and does NOT sort.  Request-level reductions (diversity,
 * interacted_with) are computed from the whole request on every shard; a model with a request-normalised column
 * (bi- / cross-encoder `norm`) has its whole matrix assembled and normalised on every shard - only the forest is
 * shared out.  The caller merges the score
 * slices of all shards into the device score buffer (one all-gather of chunk * count f64; the buffer
 * has room for the padded tail) and then calls mrk_batch_sort on the rank(s) that need the order. */
int mrk_batch_shard_chunk(mrk_batch *batch, int shard_count);
/* The shard arithmetic by itself - host-only, no context, no device (a host lays out its buffers with it, the CPU tests
 * of the N > 1 path check it): mrk_shard_chunk = ceil(total_items / shard_count) rounded up to whole MRK_SHARD_TILE-item
 * scorer tiles (negative mrk_status on bad arguments); mrk_shard_range = the items [lo, hi) shard `shard_index` assembles
 * and scores: [index * chunk, (index + 1) * chunk) clipped to total_items (trailing shards may be empty). */
#define MRK_SHARD_TILE 128
int64_t mrk_shard_chunk(int64_t total_items, int shard_count);
int mrk_shard_range(int64_t total_items, int shard_index, int shard_count, int64_t *lo, int64_t *hi);
int mrk_batch_run_shard(mrk_batch *batch, mrk_model *model, int shard_index, int shard_count);
int mrk_batch_sort(mrk_batch *batch);
/* A prepared batch owns a HIP stream: batches of one context may be in flight together (mrk_batch_run is
 * asynchronous), which is how consecutive batches overlap on the device - the assembly kernels wait on memory,
 * the scorer is VALU-bound.  mrk_batch_sync waits for this batch; mrk_batch_fetch / _status synchronise it too.
 * mrk_store_flush (and every put that grows a table) waits for all batches before it rewrites device memory. */
void *mrk_batch_stream(mrk_batch *batch);
int mrk_batch_sync(mrk_batch *batch);
/* device pointers of the batch outputs (valid until mrk_batch_free): scores f64[total_items],
 * order i32[total_items] (request-local indices), matrix f64[total_items*dim].
 * The f64 matrix (ClickthroughQuery's layout) is materialised on demand only: with a LightGBM /
 * XGBoost model of <= 16-leaf trees the assembled values go straight into the scorer's binned
 * tile.  Asking for d_matrix here makes every later mrk_batch_run of this batch write it;
 * mrk_batch_fetch(out_matrix != NULL) and mrk_rank(out_matrix != NULL) materialise it for that call. */
int mrk_batch_device_outputs(mrk_batch *batch, double **d_scores, int32_t **d_order, double **d_matrix);
/* copy results to host (synchronises); any pointer may be NULL */
int mrk_batch_fetch(mrk_batch *batch, double *out_scores, int32_t *out_order, double *out_matrix);
/* per-request outcome of the last run: out_status[n_req] = MRK_OK or the mrk_status the reference's
 * exception for that request maps to (e.g. MRK_ERR_ARITHMETIC); results of failed requests are undefined */
int mrk_batch_status(mrk_batch *batch, int32_t *out_status);
void mrk_batch_free(mrk_batch *batch);

/* ---- the serving queue ---------------------------------------------------------------------------
 * Replaces the request loop around Ranker.rerank - main/command/Serve.scala:130-150 (warm-up, then the port opens) and
 * api/routes/RankApi.scala:25-41 (one rerank per request thread).  mrk_serve_start compiles what the model needs (the
 * warm-up: no request ever waits for a compiler) and prepares n_slots slots, each served by a PERSISTENT workgroup on
 * the device that polls its slot in pinned memory.  mrk_serve_rank is then the whole request path, callable from any
 * number of host threads: the request is resolved on the calling thread and written into a free slot; the workgroup
 * assembles, scores and orders it and writes scores / order / status back into pinned memory - no launch, no copy
 * command, no HIP call.  Results and errors are exactly mrk_rank's.  Requests the one-workgroup path does not take
 * (more than 128 candidates, per-item field overrides, a model with a request-normalised column, all slots busy) go
 * through mrk_rank transparently.  A workgroup that has seen no request for a while (2 ms; MRK_SERVE_IDLE_US) leaves
 * its CU and is relaunched by the next request; store flushes stop the workgroups for their duration.
 * n_slots (1 .. 64): the slots are launched in GANGS - one kernel of up to 8 workgroups per stream, workgroup i serving slot i of
 * its gang - because a resident kernel occupies its stream's hardware queue while it stays and a process has few of those
 * (GPU_MAX_HW_QUEUES; the library sets it to 24 before its first HIP call unless the host has set it): 64 slots take 8 streams.
 * A gang leaves as a whole (idle: no request in any of its slots for MRK_SERVE_IDLE_US; old: MRK_SERVE_LIFE_US; told: a store
 * flush) and is launched again as a whole by the next request that finds its slot left.  Up to MRK_SERVE_SPIN_CALLERS (8)
 * callers wait for their answer spinning; the callers beyond them sleep through the time the device is known to need
 * (+ MRK_SERVE_SLEEP_EXTRA_US) and spin for the rest - 64 spinning threads are 64 busy CPUs, which a host with a CPU quota does
 * not have.  Callers beyond the slots are combined by mrk_rank's front.
 * While a queue is started, mrk_rank itself answers through it (same model handle and model name, no matrix asked for, at most
 * 128 candidates) and sends the rest through its batching front: a host calls mrk_serve_start once at warm-up and keeps calling
 * mrk_rank from its request fibers.
 * mrk_serve_stats: the first min(n_out, MRK_SERVE_STATS) of {requests through the queue, requests through mrk_rank, workgroup launches; then, summed over the
 * queue's requests, in ns: host resolve + pack, host publish -> acknowledgement, host copy-out, device input copy + cache drops,
 * device ranking, device result write-back; last: the SHADER CYCLES of the device ranking summed the same way - cycles / ns = the
 * clock the requests ran at (a lone workgroup on an otherwise idle device does not see the boost clock)}. */
typedef struct mrk_server mrk_server;
int mrk_serve_start(mrk_ctx *ctx, mrk_model *model, const char *model_name, int n_slots, mrk_server **out);
int mrk_serve_rank(mrk_server *srv, const mrk_request *req, double *out_scores, int32_t *out_order);
#define MRK_SERVE_STATS 10
int mrk_serve_stats(mrk_server *srv, int64_t *out, int n_out);
void mrk_serve_stop(mrk_server *srv);

/* ------------------------------------------------ multi-GPU (RCCL over xGMI) */

/* The path shards without a data-path collective except one: merging score slices (the reference has
 * no counterpart - one rerank per JVM thread, M/ml/Ranker.scala:27-83).  One process per GPU, one context each; rank
 * 0 draws an id (ncclGetUniqueId) and hands its 128 bytes to the other ranks over any channel the host has, then
 * every rank calls mrk_comm_init (ncclCommInitRank: collective, blocks until all `world` ranks have called it). */
#define MRK_COMM_ID_BYTES 128
int mrk_comm_unique_id(uint8_t *out_id /* MRK_COMM_ID_BYTES */);
int mrk_comm_init(mrk_ctx *ctx, const uint8_t *id, int rank, int world);
/* The same for ranks that live in ONE process: ctxs[i] becomes rank i of a world of n (ncclCommInitRank per context inside
 * ncclGroupStart / ncclGroupEnd - no id to pass around).  One rank per GPU: two contexts on the same ordinal are
 * MRK_ERR_INVALID_ARG.  Afterwards each context's host thread issues the collective calls below like a rank of a
 * multi-process job. */
int mrk_comm_init_local(mrk_ctx *const *ctxs, int n);
int mrk_comm_rank(mrk_ctx *ctx);   /* 0 without a communicator */
int mrk_comm_world(mrk_ctx *ctx);  /* 1 without a communicator */
/* host-value collectives for drivers: max over the ranks (in place), and a barrier */
int mrk_comm_max_f64(mrk_ctx *ctx, double *value);
int mrk_comm_barrier(mrk_ctx *ctx);
/* Item-sharded rank of a batch over the communicator's ranks (every rank holds the same batch and a replica of the
 * store): this rank's slice is assembled and scored (mrk_batch_run_shard with the communicator's rank / world), ONE
 * in-place ncclAllGather of the score slices on the batch's stream, then the sort - every rank ends up with all scores
 * and the order.  Per-request status words are merged too (an item that fails its request - dim mismatch, an XGBoost inf -
 * sits in ONE rank's slice: the words of all ranks are all-gathered and OR-ed before the sort, so every rank reports what
 * the single-GPU path reports).  Asynchronous like mrk_batch_run.  Without a communicator it is mrk_batch_run.
 * COLLECTIVE ORDER: all ranks must issue the collectives of a communicator (these calls, mrk_comm_max_f64, _barrier) in
 * the same order; the library serialises them per context, so driving several batches from several host threads is safe
 * only if every rank runs them in one agreed order. */
int mrk_batch_run_sharded(mrk_batch *batch, mrk_model *model);
/* the all-gather step alone: after mrk_batch_run_shard(batch, model, mrk_comm_rank, mrk_comm_world) on every rank */
int mrk_batch_allgather_scores(mrk_batch *batch);
/* Request-sharded replicas (every rank ranks its own batch of the same size): the scores of all ranks in one device
 * buffer owned by the batch, rank r's at [r * total_items, (r + 1) * total_items); asynchronous on the batch's stream. */
int mrk_batch_gather_scores(mrk_batch *batch, double **d_all_scores);

/* -------------------------------------------------------------- utilities */

int mrk_sync(mrk_ctx *ctx);            /* hipStreamSynchronize on the context stream */
void *mrk_stream(mrk_ctx *ctx);        /* the hipStream_t all work is enqueued on     */

/* HIP-event timing of the kernels launched by the last mrk_batch_run / predict call on this ctx.
 * names: scorer "score", assembly "assemble", request pre-pass "prepass", sort "sort".
 * Enable with mrk_profile_enable(ctx, 1); returns accumulated ms and launch count since enable. */
int mrk_profile_enable(mrk_ctx *ctx, int on);
int mrk_profile_get(mrk_ctx *ctx, const char *kernel, double *total_ms, int64_t *launches);

/* ---- text encoders (BASELINE config 5) ------------------------------------------------------------
 * Replaces ml/onnx/sbert/OnnxSession.scala:29-57 (load), OnnxBiEncoder.scala:13-60 (embed + masked mean pool) and
 * OnnxCrossEncoder.scala:22-51 (pair logits).  The tokenizer is host code and needs no device. */
typedef struct mrk_tokenizer mrk_tokenizer;
typedef struct mrk_encoder mrk_encoder;

/* HuggingFaceTokenizer.newInstance(tokenizer.json, {padding: true, truncation: true}) -- OnnxSession.scala:42-43.
 * BERT WordPiece pipelines only; other pipelines -> MRK_ERR_UNSUPPORTED. */
int mrk_tokenizer_load(const char *tokenizer_json, size_t len, mrk_tokenizer **out);
/* tokenizer.batchEncode(a) / batchEncode(PairList(a, b)): rows padded to the longest row of the batch.  `b` may be
 * NULL.  ids/type_ids/mask are n x capacity row-major with row stride *seq_len on return; MRK_ERR_INVALID_ARG when
 * capacity < longest row (then *seq_len holds the length needed). */
int mrk_tokenizer_encode_batch(mrk_tokenizer *tok, const char *const *a, const char *const *b, int n, int32_t *ids,
                               int32_t *type_ids, int32_t *mask, int capacity, int *seq_len);
void mrk_tokenizer_free(mrk_tokenizer *tok);

typedef struct mrk_encoder_info {
  int32_t layers, hidden, heads, intermediate, vocab, max_positions, type_vocab;
  int32_t has_classifier; /* pooler + 1-logit classifier present (cross-encoder) */
  int32_t max_length;     /* tokenizer truncation length */
  int64_t device_bytes;
  double layer_norm_eps;
} mrk_encoder_info;

/* env.createSession(modelBytes) + the tokenizer -- OnnxSession.scala:42-56.  `weights` is the model file the reference
 * reads (`pytorch_model.onnx`: the initializers of a BERT-family graph are extracted) or the same checkpoint as
 * `model.safetensors`; detected by content.  `heads` = number of attention heads (0: read it from the graph's
 * reshape constants / the safetensors metadata key "num_attention_heads").  Arithmetic: MRK_ENCODER_F32 (below) - the
 * reference's session is an fp32 onnxruntime session, and scores within 1e-5 / the same order need its arithmetic. */
int mrk_encoder_load(mrk_ctx *ctx, const uint8_t *weights, size_t len, const char *tokenizer_json, size_t tok_len,
                     int heads, mrk_encoder **out);
/* The same with the arithmetic chosen by the caller - a property of the handle, never of the size of a call.
 * MRK_ENCODER_F32 (what mrk_encoder_load uses since ABI 8): f32 weights, every matrix product and the attention on the
 * f32-input matrix instruction (v_mfma_f32_16x16x4_f32: exact f32, bit for bit a chain of fma's), libm erf / exp - the
 * arithmetic of the reference's fp32 ONNX session (onnxruntime on the CPU, OnnxSession.scala:42-56); cosines within 1e-5 of
 * transformers' fp32 output (the reference's own tests accept 1e-3, OnnxBiencoderTest.scala:23-25).  Every f32 kernel walks
 * k in one canonical order from a zero accumulator, so a sequence's embedding is the same BITS alone (mrk_rank), in a packed
 * batch of thousands (mrk_batch_*), padded or packed: a request's scores do not depend on the load it arrived under.
 * About 3.5x the time of the fp16 path per packed batch (52 % of the 157 TFLOP/s f32 matrix peak), the same latency for
 * a single query.
 * MRK_ENCODER_FP16 (opt-in; BASELINE config 5's "fp16"): fp16 weights and operands on the matrix cores, f32 accumulation -
 * pooled cosines within 3e-3 of an fp32 run of the graph; about 1 % of a 500-tree forest's scores move by more than 1e-5 and
 * a third of the requests see a pair of candidates swapped (bench.py's config-5 line reports it per run).
 * MRK_ENCODER_AUTO: accepted for hosts built against ABI <= 7 (where it chose by call size); it is MRK_ENCODER_F32 now. */
enum { MRK_ENCODER_FP16 = 0, MRK_ENCODER_F32 = 1, MRK_ENCODER_AUTO = 2 };
int mrk_encoder_load_ex(mrk_ctx *ctx, const uint8_t *weights, size_t len, const char *tokenizer_json, size_t tok_len, int heads,
                        int precision, mrk_encoder **out);
int mrk_encoder_get_info(mrk_encoder *enc, mrk_encoder_info *info);
/* Host-only view of what mrk_encoder_load reads from a model file: writes a JSON object
 * {"heads": h, "tensors": {name: {"shape": [...], "sum": s, "abs_sum": a}}} (BertModel state_dict names, Linear weights
 * as [out, in]) into `out`; MRK_ERR_INVALID_ARG with *needed set when `cap` is too small. No device needed. */
int mrk_checkpoint_describe(const uint8_t *weights, size_t len, char *out, size_t cap, size_t *needed);
/* OnnxBiEncoder.embed: n texts -> n x hidden f32 (masked mean over tokens, f64 accumulation, no normalisation) */
int mrk_encoder_embed(mrk_encoder *enc, const char *const *texts, int n, float *out);
/* same from token ids (n x seq_len, as mrk_tokenizer_encode_batch returns them) */
int mrk_encoder_embed_ids(mrk_encoder *enc, const int32_t *ids, const int32_t *type_ids, const int32_t *mask, int n,
                          int seq_len, float *out);
/* last_hidden_state (output 0 of the graph) for parity checks: n x seq_len x hidden f32 */
int mrk_encoder_hidden_ids(mrk_encoder *enc, const int32_t *ids, const int32_t *type_ids, const int32_t *mask, int n,
                           int seq_len, float *out);
/* OnnxCrossEncoder.encode: n (a, b) pairs -> n logits; MRK_ERR_UNSUPPORTED without a classifier head */
int mrk_encoder_score_pairs(mrk_encoder *enc, const char *const *a, const char *const *b, int n, float *out);
int mrk_encoder_score_ids(mrk_encoder *enc, const int32_t *ids, const int32_t *type_ids, const int32_t *mask, int n,
                          int seq_len, float *out);
void mrk_encoder_free(mrk_encoder *enc);
/* FieldMatchBiencoderSchema.create / FieldMatchCrossEncoderSchema.create with `method.model` set
 * (feature/FieldMatchBiencoderFeature.scala:118-124, FieldMatchCrossEncoderFeature.scala:131-135): from now on
 * mrk_rank / mrk_batch_prepare encode the request's `rankingField` text with this encoder (queries are cached by
 * text, as EmbeddingCache does) instead of expecting a host-computed embedding. */
int mrk_config_bind_encoder(mrk_ctx *ctx, const char *feature, mrk_encoder *enc);

#ifdef __cplusplus
}
#endif
#endif /* MRK_H */
