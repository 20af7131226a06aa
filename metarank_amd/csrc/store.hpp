// Device-resident feature store: the mirror of KVStore[Key, FeatureValue] (reference:
// fstore/Persistence.scala:85-89, fed by flow/FeatureValueSink.scala:10-14).
//
// Layout in HBM.  One table per scope type (item, user, session, global, ranking, field, irf);
// a table is an array of fixed-stride RECORDS, one per scope instance ("slot"), so that the
// gather of one candidate item touches a single contiguous record (a few 128-B lines) instead of
// one line per feature column.  A record is
//     [tag bytes: one per column][u16 heap_used][8-byte value cells, one per column (P cells for a P-period counter)]
//     [inline heap: the interned tokens of the record's own string lists]
// padded to a whole number of 128-byte lines (64 bytes for tiny records) and starting on one: a candidate costs exactly
// stride / 128 lines, its tokens included.  A string list cell is {u32 offset, u32 length}; offset has LIST_INLINE
// set when the tokens live in the record's own heap (byte offset from the record's start), else it indexes the
// token pool.  Lists that do not fit the heap, double lists and bounded lists (item slots) live in three pools
// addressed by {offset, length} cells; pool ranges come in power-of-two size classes and are recycled through free
// lists when a value is replaced, so a long-running store does not grow with the number of puts.
// Strings are interned host-side (one dictionary for the whole store); the second key hop of the
// item-field-scoped `rate` (item -> "field=<name>:<value>") is resolved at put time into a direct
// slot reference.
#pragma once
#include <cstdint>
#include <climits>
#include <cstring>
#include <memory>
#include <queue>
#include <stdexcept>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "device_types.hpp"
#include "runtime.hpp"

namespace mrk {


enum ColKind : uint8_t { COL_SCALAR = 0, COL_COUNTER = 1, COL_PERIODIC = 2, COL_BOUNDED_LIST = 3 };


struct Column {
  std::string name;
  ColKind kind = COL_SCALAR;
  int periods = 0;           // COL_PERIODIC: number of i64 cells
  int tag_index = 0;         // byte index of the tag inside the record
  int val_off = 0;           // byte offset of the first value cell inside the record
  std::string link_field;    // SString scalar whose value is also a key into the FIELD table ("field=<link_field>:<value>")
  bool expect_list = false;  // the feature that reads it stores string lists here: the record's inline heap is sized for it
  // write path (raw state): PeriodicCounterConfig(period, sumPeriodRanges = periods.map(PeriodRange(_, 0))),
  // model/Feature.scala:196-209; BoundedListConfig(count, duration), model/Feature.scala:100-108
  int64_t period_ms = 0;            // COL_PERIODIC: bucket length; 0 = no write-path config (values arrive by put only)
  std::vector<int32_t> offsets;     // COL_PERIODIC: PeriodRange.startOffset per cell
  int ring_off = -1;                // byte offset of this column's bucket ring inside the table's ring record
  int ring_w = 0;                   // ring entries = max(offsets) + 1
  int64_t list_count = INT64_MAX;   // COL_BOUNDED_LIST
  int64_t list_duration_ms = INT64_MAX;
  // the longest string list ever put into this column (an SString counts 1): upper bound used to size the pre-pass
  // hash tables of a request without reading item records on the host (features.cpp resolve_requests)
  uint32_t max_len = 0;
};

// device descriptor of a periodic column that owns a bucket ring
constexpr int RING_MAX_RANGES = 16;
struct RingColDev {
  uint32_t ring_off, w, val_off, tag_index;
  int64_t period_ms;
  int32_t n_ranges;
  int32_t offsets[RING_MAX_RANGES];
  int32_t pad;
};
struct IncGroup { uint32_t slot, col, begin, end; };   // updates [begin, end) of one (slot, ring column)
struct IncUpdate { int64_t bucket, inc; };             // bucket = start of period (ms)
constexpr unsigned long long RING_EMPTY = 0x8080808080808080ull;  // memset(0x80): "no bucket here"

// id -> slot of one table: open addressing over 64-bit hashes, the ids kept in one arena (a lookup is one probe into
// a flat array plus one compare against the arena; std::unordered_map<std::string, .> costs two dependent cache
// misses and a temporary string per lookup, and id lookups are 70 % of the host's share of a rank batch).
// Slots are assigned densely in insertion order, so slot s's id is the s-th string of the arena.
struct SlotMap {
  typedef IdEntry Entry;          // {hash, slot, off}: hash 0 = empty; off: the id (NUL-terminated) in `ids`
  std::vector<Entry> table;       // power-of-two capacity, load <= 1/2
  std::string ids;
  size_t n = 0;
  // device mirror (ITEM table only, store.cpp flush_ids): entries placed since the last upload; `regrown` = the whole
  // table moved
  bool mirrored = false;
  bool regrown = false;
  std::vector<uint32_t> touched;
  static uint64_t hash(const char *s, size_t len) {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (len * 0xff51afd7ed558ccdull);
    size_t i = 0;
    for (; i + 8 <= len; i += 8) {
      uint64_t w;
      memcpy(&w, s + i, 8);
      h = (h ^ w) * 0x9fb21c651e98df25ull;
      h ^= h >> 29;
    }
    uint64_t w = 0;
    memcpy(&w, s + i, len - i);
    h = (h ^ w) * 0x9fb21c651e98df25ull;
    h ^= h >> 32;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 29;
    return h ? h : 1;
  }
  static constexpr uint32_t NONE = 0xffffffffu;
  // a batch of lookups first asks for the home entries of all its keys (independent cache misses overlap), then probes
  void prefetch(uint64_t h) const {
    if (!table.empty()) __builtin_prefetch(&table[(size_t)h & (table.size() - 1)]);
  }
  uint32_t find_hashed(uint64_t h, const char *s, size_t len) const {
    if (table.empty()) return NONE;
    const size_t mask = table.size() - 1;
    for (size_t i = (size_t)h & mask;; i = (i + 1) & mask) {
      const Entry &e = table[i];
      if (e.hash == 0) return NONE;
      if (e.hash == h) {
        const char *id = ids.data() + e.off;
        if (memcmp(id, s, len) == 0 && id[len] == '\0') return e.slot;
      }
    }
  }
  uint32_t find(const char *s, size_t len) const { return find_hashed(hash(s, len), s, len); }
  void insert(const char *s, size_t len, uint32_t slot) {  // `s` (no NUL inside) must not be present
    if ((n + 1) * 2 > table.size()) grow();
    if (ids.size() + len + 1 > 0xffffffffull) throw std::runtime_error("slot map: more than 4 GB of ids in one table");
    const uint32_t off = (uint32_t)ids.size();
    ids.append(s, len);
    ids.push_back('\0');
    place(Entry{hash(s, len), slot, off});
    ++n;
  }

 private:
  void place(const Entry &e) {
    const size_t mask = table.size() - 1;
    size_t i = (size_t)e.hash & mask;
    while (table[i].hash != 0) i = (i + 1) & mask;
    table[i] = e;
    if (mirrored && !regrown) touched.push_back((uint32_t)i);
  }
  void grow() {
    regrown = true;
    touched.clear();
    std::vector<Entry> old;
    old.swap(table);
    table.assign(old.empty() ? 64 : old.size() * 2, Entry{0, 0, 0});
    for (const Entry &e : old)
      if (e.hash) place(e);
  }
};

struct Table {
  ScopeId scope;
  std::vector<Column> cols;
  std::unordered_map<std::string, int> col_of;
  uint32_t stride = 64;
  uint32_t heap_used_off = 0;     // byte offset of the u16 "inline heap bytes in use" inside the record
  uint32_t heap_off = 0;          // first byte of the inline heap = end of the value cells
  uint32_t heap_cap = 0;          // stride - heap_off
  SlotMap slot_of;
  std::vector<uint8_t> rows;      // host mirror, n_slots * stride
  uint32_t n_slots = 0;
  // device side
  DevBuf d_rows;
  uint32_t d_slots_cap = 0;       // slots allocated on device
  // write path: one ring record per slot = the bucket rings of every periodic column with a config; device only
  uint32_t ring_stride = 0;
  std::vector<RingColDev> ring_cols;      // host copy of the descriptors
  std::vector<int> ring_col_of;           // column index -> index into ring_cols, -1
  DevBuf d_ring, d_ring_cols;
  uint32_t d_ring_slots = 0;
  bool ring_used = false;
  uint32_t dirty_lo = UINT32_MAX, dirty_hi = 0;  // slot range to upload
  bool dirty_all = false;
  // device mirror of slot_of (ITEM table): lets a kernel resolve item ids (resolve.hip)
  DevBuf d_id_table, d_id_arena;
  size_t d_id_entries = 0;        // entries allocated = host capacity at the last full upload
  size_t d_arena_cap = 0, d_arena_uploaded = 0;
  IdTableDev id_table_view() const {
    IdTableDev v{};
    v.table = d_id_entries ? (const IdEntry *)d_id_table.p : nullptr;
    v.arena = (const uint8_t *)d_id_arena.p;
    v.mask = d_id_entries ? (uint32_t)(d_id_entries - 1) : 0u;
    return v;
  }
  void mark(uint32_t slot) {
    if (slot < dirty_lo) dirty_lo = slot;
    if (slot + 1 > dirty_hi) dirty_hi = slot + 1;
  }
};

constexpr uint32_t LIST_INLINE = 0x80000000u;  // list cell: the tokens are in the record's own heap (device_types.hpp)
constexpr uint32_t POOL_MAX = 0x7fffffffu;     // pool offsets keep the top bit free

// A pool of variable-length values.  Ranges are allocated in power-of-two size classes (a range of class c holds 2^c
// elements) and recycled: replacing a value whose class still fits rewrites it in place, anything else returns the old
// range to its class's free list.  Element 0 is never handed out ({0, 0} = the empty list).
template <typename T>
struct Pool {
  std::vector<T> host;
  DevBuf dev;
  size_t dev_cap = 0;                                 // elements allocated on device
  std::vector<std::pair<size_t, size_t>> dirty;      // element ranges [lo, hi) to upload
  std::vector<uint32_t> free_[32];
  static int cls(uint32_t n) { return n <= 1 ? 0 : 32 - __builtin_clz(n - 1); }
  uint32_t align = 1;            // ranges start on a multiple of this many elements
  uint32_t alloc(uint32_t n) {  // n >= 1
    const int c = cls(n);
    if (!free_[c].empty()) { const uint32_t off = free_[c].back(); free_[c].pop_back(); return off; }
    if (host.size() % align) host.resize((host.size() + align - 1) / align * align, T());
    const size_t off = host.size(), len = (size_t)1 << c;
    if (off + len > POOL_MAX) throw std::length_error("feature store: a value pool would exceed 2^31 entries");
    host.resize(off + len, T());
    return (uint32_t)off;
  }
  void release(uint32_t off, uint32_t n) { if (n && off) free_[cls(n)].push_back(off); }
  // the range for a value of n elements replacing one of old_n elements at old_off (0, 0: none)
  uint32_t realloc(uint32_t old_off, uint32_t old_n, uint32_t n) {
    if (n == 0) { release(old_off, old_n); return 0; }
    if (old_n && old_off && cls(old_n) == cls(n)) return old_off;
    release(old_off, old_n);
    return alloc(n);
  }
  void write(uint32_t off, const T *v, uint32_t n) {
    if (!n) return;
    std::copy(v, v + n, host.begin() + off);
    if (!dirty.empty() && dirty.back().second == off) dirty.back().second = off + n;  // consecutive appends coalesce
    else dirty.emplace_back(off, off + n);
  }
};


// Key(scope, feature) (model/Key.scala:7-10) with the scope already told apart: `id` is what follows "<kind>=" in
// ScopeCodec.encode (fstore/codec/impl/ScopeCodec.scala:18-26) - the item / user / session / ranking id, or
// "<field>:<value>" / "<field>:<value>:<item>" for the field scopes; empty for the global scope.  Views: nothing is copied.
struct KeyRef {
  ScopeId scope = SC_GLOBAL;
  std::string_view id, feature;
};

struct Store {
  Table tables[SC_COUNT];
  Pool<uint32_t> tok_pool;   // interned token ids of string lists
  Pool<double> f64_pool;
  Pool<float> f32_pool;      // double lists whose values are all exactly floats (LIST_F32), ranges 16-byte aligned
  Pool<uint32_t> slot_pool;  // item slots of bounded lists
  std::unordered_map<std::string, uint32_t> token_of;  // string -> token id (>= 1)
  struct PendingInc { uint8_t table; uint32_t slot; uint32_t ring_col; int64_t bucket, inc; };
  std::vector<PendingInc> pending;   // staged PeriodicIncrements
  std::unordered_map<std::string, std::vector<std::pair<int64_t, std::string>>> lists;  // raw bounded lists, newest first
  // item texts of cross-encoder columns (FieldMatchCrossEncoderFeature.scala:60-70 stores SString(text) under the item
  // scope): host-only, tokenised lazily by the bound encoder's tokenizer; feature -> item id -> text
  struct ItemText { std::string text; std::vector<int32_t> pieces; bool tokenized = false; };
  std::unordered_map<std::string, std::unordered_map<std::string, ItemText>> texts;
  DevBuf d_groups, d_updates;        // scratch of the apply kernel
  bool frozen = false;       // layout frozen after the first slot is created
  uint64_t version = 0;      // bumped on every put
  std::unordered_map<uint64_t, int64_t> ttl_deadline;   // cell (scope | column | slot) -> deadline, ms since the epoch
  std::priority_queue<std::pair<int64_t, uint64_t>, std::vector<std::pair<int64_t, uint64_t>>, std::greater<std::pair<int64_t, uint64_t>>> ttl_heap;  // (deadline, cell), stale entries skipped
  static uint64_t ttl_cell(ScopeId scope, uint32_t col, uint32_t slot) { return ((uint64_t)scope << 58) | ((uint64_t)col << 32) | slot; }

  Store();
  // layout (called by the registry while loading the config)
  int add_column(ScopeId scope, const std::string &name, ColKind kind, int periods, const std::string &link_field = "",
                 bool expect_list = false);
  void freeze_layout();

  uint32_t intern(const std::string &s);
  uint32_t find_token(const std::string &s) const;  // 0 if never seen
  uint32_t slot(ScopeId scope, const std::string &id, bool create) { return slot(scope, id.data(), id.size(), create); }
  uint32_t slot(ScopeId scope, const char *id, size_t len, bool create);
  static constexpr uint32_t NO_SLOT = 0xffffffffu;

  // Key.encode ("<ScopeCodec.encode(scope)>/<feature>", model/Key.scala:9) -> KeyRef viewing `key`; false if malformed.
  // Key.fromString splits on the FIRST '/', so an id that itself contains '/' cannot travel in this form - the
  // structured KeyRef overloads below (used by the binary loader, codec.cpp) take any bytes.
  static bool parse_key(const char *key, KeyRef &out);

  // puts: return false when the key does not belong to any configured column (ignored)
  bool put_double(const KeyRef &k, double v);
  bool put_bool(const KeyRef &k, bool v);
  bool put_string(const KeyRef &k, std::string_view v);
  bool put_string_list(const KeyRef &k, const std::string_view *v, int n);
  bool put_double_list(const KeyRef &k, const double *v, int n);
  bool put_counter(const KeyRef &k, int64_t v);
  bool put_periodic(const KeyRef &k, const int64_t *v, int n);
  bool put_bounded_list(const KeyRef &k, const std::string_view *v, int n);
  bool erase(const KeyRef &k);
  // ---- FeatureValue.expire (FeatureValueCodec.scala:42-48,75; the Redis store drops a key `expire` after its last write,
  // RedisKVStore.scala:40).  Opt-in - only records applied through mrk_store_put_binary_at are tracked: 8 + 16 bytes of host
  // memory per live deadline.  A later write of the same cell through any put replaces or clears its deadline.
  void ttl_note(const KeyRef &k, int64_t deadline_ms);
  void ttl_note(const char *key, int64_t deadline_ms) { ttl_note(need_key(key), deadline_ms); }
  int64_t ttl_expire(int64_t now_ms);                  // drops every value whose deadline has passed; returns how many
  size_t ttl_tracked() const { return ttl_deadline.size(); }
  // the same for Key.encode strings and C strings (the C ABI)
  bool put_double(const char *key, double v) { return put_double(need_key(key), v); }
  bool put_bool(const char *key, bool v) { return put_bool(need_key(key), v); }
  bool put_string(const char *key, const char *v);
  bool put_string_list(const char *key, const char *const *v, int n);
  bool put_double_list(const char *key, const double *v, int n) { return put_double_list(need_key(key), v, n); }
  bool put_counter(const char *key, int64_t v) { return put_counter(need_key(key), v); }
  bool put_periodic(const char *key, const int64_t *v, int n) { return put_periodic(need_key(key), v, n); }
  bool put_bounded_list(const char *key, const char *const *v, int n);
  bool erase(const char *key) { return erase(need_key(key)); }

  // ---- write path (raw Writes of flow/FeatureValueFlow.scala:44-62; the FeatureValue is derived here) ----
  void set_periodic_config(ScopeId scope, const std::string &name, int64_t period_ms, const std::vector<int32_t> &offsets);
  void set_list_config(ScopeId scope, const std::string &name, int64_t count, int64_t duration_ms);
  // Write.PeriodicIncrement: staged; applied to the device bucket rings (and the window sums recomputed
  // there) at the next flush
  bool increment_periodic(const char *key, int64_t ts_ms, int64_t inc);
  bool increment(const char *key, int64_t inc);                          // Write.Increment
  bool append(const char *key, const char *value, int64_t ts_ms);        // Write.Append(SString)

  // Measurement aid (mrk_debug_clone_items): every ITEM-scope instance that exists now gets `copies` deep copies under
  // the ids "<id>#1" .. "<id>#<copies>" - the way bench.py grows a generated catalogue past the Infinity Cache without
  // generating (and putting) hundreds of millions of values through Python.  Returns the number of items afterwards.
  uint32_t clone_items(int copies);

  // host-side readers (used to size per-request scratch; never to compute features)
  const uint8_t *record(ScopeId scope, uint32_t slot) const {
    return tables[scope].rows.data() + (size_t)slot * tables[scope].stride;
  }

  // copies everything that changed since the last call to the device (async on stream)
  void flush(hipStream_t stream);
  void flush_writes(hipStream_t stream, const uint32_t *uploaded_lo, const uint32_t *uploaded_hi);
  bool dirty() const;   // anything flush() would upload
  void flush_ids(Table &t, hipStream_t stream);
  StoreDev device_view() const;
  size_t device_bytes() const;

 private:
  struct Cell { Table *t; Column *c; uint32_t slot; uint8_t *rec; };
  static KeyRef need_key(const char *key);
  bool locate(const KeyRef &k, Cell &out);
  void set_tag(Cell &c, uint8_t tag) { c.rec[c.c->tag_index] = tag; }
  template <typename T> void set_val(Cell &c, int idx, T v) { memcpy(c.rec + c.c->val_off + idx * 8, &v, sizeof(T)); }
  // returns the pool range / heap bytes a cell's current value holds before the value is replaced
  void drop_value(Cell &c, bool keep_tok_range = false);
  void put_tokens(Cell &c, const uint32_t *toks, uint32_t n);
  bool heap_place(Table &t, uint8_t *rec, const Column *col, const uint32_t *toks, uint32_t n, uint32_t &off_out);
};

// codec.cpp: a `ranking` event decoded from the reference's binary RankingEventFormat; `req` points into this object
struct DecodedRequest {
  mrk_request req;
  std::vector<std::unique_ptr<std::string>> strs;
  std::vector<std::unique_ptr<std::vector<const char *>>> lists;
  std::vector<std::unique_ptr<std::vector<double>>> nums;
  std::vector<mrk_field> fields, item_fields;
  std::vector<const char *> item_ids;
  std::vector<int32_t> item_offsets;
  size_t decode(const uint8_t *bytes, size_t len);  // returns the bytes consumed; throws StatusError(MRK_ERR_PARSE)
};

// writes.hip: bucket rings -> window sums (PeriodicCounterFeature.fromMap, model/Feature.scala:142-161)
void launch_periodic_apply(hipStream_t stream, uint8_t *rows, uint32_t stride, uint8_t *ring, uint32_t ring_stride,
                           const RingColDev *cols, const IncGroup *groups, int n_groups, const IncUpdate *updates);
void launch_periodic_refresh(hipStream_t stream, uint8_t *rows, uint32_t stride, uint8_t *ring, uint32_t ring_stride,
                             const RingColDev *cols, int n_cols, uint32_t slot_lo, uint32_t slot_hi);

}  // namespace mrk
