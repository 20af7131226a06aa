// Shared host/device structures of the rank pipeline (pre-pass -> assemble -> score -> sort).
#pragma once
#include "device_types.hpp"

namespace mrk {

// One assembly op = one feature of the model, writes `dim` consecutive matrix columns at `dst`
// (DatasetDescriptor offsets, reference FeatureMapping.scala:89-99 / ClickthroughQuery.scala:50-74).
enum OpKind : int32_t {
  OP_SCALAR_DOUBLE = 0,  // number / word_count: SDouble -> v else NaN        (NumberFeature.scala:54-69)
  OP_SCALAR_BOOL = 1,    // boolean: SBoolean -> 1/0 else NaN                  (BooleanFeature.scala:53-67)
  OP_VECTOR = 2,         // vector: SDoubleList copied, else NaN x dim         (NumVectorFeature.scala:58-73)
  OP_STRING_INDEX = 3,   // string encode:index                                (StringFeature.scala:124-137)
  OP_STRING_ONEHOT = 4,  // string encode:onehot                               (OneHotEncoder.scala:11-22)
  OP_COUNTER = 5,        // interaction_count: Long -> double, missing 0.0     (InteractionCountFeature.scala:44-59)
  OP_WINDOW = 6,         // window_count                                       (WindowInteractionCountFeature.scala:50-63)
  OP_RATE = 7,           // rate, plain and normalised                         (RateFeature.scala:290-356)
  OP_INTERACTED = 8,     // interacted_with                                    (InteractedWithFeature.scala:133-164)
  OP_DIVERSITY = 9,      // diversity                                          (DiversityFeature.scala:67-132)
  OP_ITEM_AGE = 10,      // item_age                                           (ItemAgeFeature.scala:73-83)
  OP_CONST = 11,         // request-level constant(s) evaluated by the host part (position, local_time,
                         // ranking-scoped number/word_count/string, ua/referer columns)
  OP_FILL_NAN = 12,      // relevancy / host-computed per-item features: NaN unless an override arrives
  OP_BIENCODER = 13,     // field_match bi-encoder cosine                      (FieldMatchBiencoderFeature.scala:80-109)
};

enum RateMode : int32_t { RATE_ITEM = 0, RATE_ITEM_FIELD = 1, RATE_RANKING_FIELD = 2 };
enum NormMode : int32_t { NORM_NOOP = 0, NORM_MINMAX = 1, NORM_POSITION = 2 };

struct ColRef {
  int32_t tag;  // byte index of the tag in the record, -1 = no such column
  int32_t val;  // byte offset of the value cell
};

struct Op {
  int32_t kind;
  int32_t dst;
  int32_t dim;
  int32_t scope;      // ScopeId of the primary column
  ColRef c0;          // primary column (rate: top | item-field link column)
  ColRef c1;          // rate: bottom
  ColRef c2, c3;      // rate: global top / global bottom (GLOBAL table) ; item-field mode: top / bottom in FIELD table live in c4/c5
  ColRef c4, c5;
  int32_t i0, i1, i2, i3;  // kind specific (see features.cpp)
  double d0;               // rate weight
};

// entries of the per-request pre-pass (cross-item reductions)
enum PrepKind : int32_t { PREP_IW_FIELD = 0, PREP_DIVERSITY = 1 };
struct PrepEntry {
  int32_t kind;
  ColRef item_col;     // IW: item.<name>_<field> ; diversity: item.<name>
  int32_t list_scope;  // IW: SC_SESSION | SC_USER
  ColRef list_col;     // IW: <name>_interactions bounded list
  int32_t top;         // diversity: schema.top
  int32_t pad;
};

enum DivMode : int32_t { DIV_EMPTY = 0, DIV_STRING = 1, DIV_DOUBLE = 2 };
struct PrepOut {
  uint32_t tab_off;    // first entry of this hash table in the arena (host-sized)
  uint32_t tab_cap;    // capacity in entries (> number of tokens the host counted for it)
  double scalar;       // diversity: median (DOUBLE) or sum of counts (STRING)
  int32_t mode;        // DivMode
  int32_t preset;      // 1: mode / scalar were computed by the host (a numeric diversity over more values than the pre-pass sorts in LDS): the pre-pass leaves the entry alone
};

struct ReqDev {
  int32_t item_begin, n_items;
  int32_t user_slot, session_slot, ranking_slot;
  uint32_t arena_begin;  // first hash-table entry of this request in the arena (its tables are contiguous)
  int64_t ts_ms;
};

struct Override {  // per-request dense inputs that win over the store (item.fields overrides, relevancy, ...)
  uint32_t item;   // batch item index
  uint32_t col;    // matrix column
  double value;
};

// per-request status bits (become mrk_status on the host)
enum : int32_t {
  ST_ARITHMETIC = 1,    // / by zero in the normalised rate
  ST_ILLEGAL_ARG = 2,   // FiniteDuration overflow in item_age
  ST_TABLE_FULL = 4,    // a hash table was under-sized (store changed after prepare)
  ST_TOO_MANY = 8,      // diversity over more values than the pre-pass supports
  ST_DIM = 16,          // embedding shorter than the query
  ST_XGB_INF = 32,      // XGBoost: a value that is +-inf after the Double -> Float narrowing
  ST_BAD_IDS = 128,     // flat item ids: an item's offsets descend or pass bytes_len (set by resolve_ids_kernel at load time)
};

struct ProgramDev {
  const Op *ops;
  int32_t n_ops;
  const PrepEntry *prep;
  int32_t n_prep;
  const uint32_t *aux;   // possible-value tokens of string features, interacted_with column refs
  int32_t dim;           // matrix columns
  int32_t n_consts;      // f64 constants per request
};

struct BatchDev {
  const ReqDev *reqs;
  int32_t n_req;
  int32_t total_items;
  int32_t item_lo, item_hi;   // the batch items this launch assembles (item-sharded runs; else 0, total_items)
  const int32_t *item_slot;   // ITEM table slot of every batch item (-1: unknown id)
  const uint32_t *item_req;   // request index of every batch item
  const double *consts;       // n_req * n_consts
  const int32_t *irf;         // n_irf * total_items : IRF table slots (-1 none)
  const Override *overrides;
  int32_t n_overrides;
  PrepOut *prep_out;          // n_req * n_prep
  unsigned long long *arena;  // hash tables
  int32_t *status;            // n_req
  double *matrix;             // total_items * dim, row-major
  double *scores;             // total_items
  int32_t *order;             // total_items (request-local indices)
};

constexpr int PREP_MAX_VALUES = 4096;  // diversity numeric: values sorted in LDS
// where the one-launch kernel (rank_device.hpp rank_one_body) leaves a batch's results: the host-visible (pinned) buffer
struct OneOut {
  double *scores;        // [total_items], request order
  int32_t *order;        // [total_items], request-local indices in response order
  int32_t *status;       // [n_req] status words of this run, [n_req_pad .. n_req_pad + n_req) what the id resolution found at load time
  const int32_t *load_status;   // device copy of the latter (nullptr: zeros)
  int32_t n_req_pad;     // max(n_req, 1)
};

// ---- the serving queue (rank_device.hpp rank_serve_body, capi_rank.cpp mrk_serve_*): one persistent workgroup per slot,
// launched in GANGS - one kernel of up to SERVE_GANG workgroups on one stream, workgroup i serving slot i of the gang - so
// that 64 slots take 8 streams (a resident kernel holds its stream's hardware queue, and a process has few of those).
// ServeCtl lives in pinned host memory: the host writes `seq` / `stop` and the request header, the device `ack` /
// `exited`.  No word is written from both sides.
struct ServeCtl {
  uint32_t seq;       // host: number of the request in the slot's input block (published last, after the block and this header)
  uint32_t stop;      // host: 1 = leave at the next poll
  uint32_t ack;       // device: last request whose results are in the slot's output block
  uint32_t exited;    // device: the launch id of the gang whose workgroup has left this slot (idle, old, or told to stop)
  // the request's header
  uint32_t in_bytes;  // bytes of the input block
  uint32_t o_reqs, o_consts, o_irf, o_prep, o_slot, o_ireq;   // where the arrays of BatchDev start inside it
  uint32_t total_items, tab_entries, vals_cap, mode;
  uint32_t pad[17];   // 128 B
};
struct ServeSlotDev {        // one per slot, in device memory, written once (mrk_serve_start)
  ServeCtl *ctl;             // pinned
  const uint8_t *in_host;    // pinned input block (the host side's packing of one request: build_batch's layout)
  uint8_t *in_dev;           // its device copy, made by the workgroup itself
  OneOut out;                // pinned output block
};
constexpr int SERVE_GANG = 8;
struct ServeGangDev {        // kernel argument of one gang launch
  const ServeSlotDev *slots; // [gridDim.x]
  unsigned long long *clock; // device word: wall_clock64 of the gang's last served request (idleness is the gang's, not a slot's)
  uint32_t launch_id, pad;
  unsigned long long idle_ticks;   // wall_clock64 ticks (100 MHz) without a request in the GANG after which its workgroups leave
  unsigned long long life_ticks;   // ... and the age at which each leaves after the request it is serving, however busy the slot is:
                                   // hipFree / hipHostFree / a device-wide sync on ANY thread wait for every resident kernel
};

constexpr int SORT_MAX_ITEMS = 4096;   // per-request LDS sort

}  // namespace mrk
