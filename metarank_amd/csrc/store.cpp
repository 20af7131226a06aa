// Host side of the device feature store (see store.hpp for the HBM layout).
#include "store.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace mrk {

Store::Store() {
  for (int s = 0; s < SC_COUNT; ++s) tables[s].scope = (ScopeId)s;
  tables[SC_ITEM].slot_of.mirrored = true;  // item ids are resolved on the device (resolve.hip)
  tok_pool.host.push_back(0);   // offset 0 is never a valid list start for a non-empty list; keeps {0,0} == empty
  f64_pool.host.push_back(0.0);
  slot_pool.host.push_back(0);
}

int Store::add_column(ScopeId scope, const std::string &name, ColKind kind, int periods, const std::string &link_field) {
  if (frozen) throw StatusError(MRK_ERR_INVALID_ARG, "store layout is frozen");
  Table &t = tables[scope];
  auto it = t.col_of.find(name);
  if (it != t.col_of.end()) {
    Column &c = t.cols[it->second];
    if (c.kind != kind || c.periods != periods)
      throw StatusError(MRK_ERR_INVALID_ARG, "state '" + name + "' is declared twice with different types");
    if (!link_field.empty()) c.link_field = link_field;
    return it->second;
  }
  Column c;
  c.name = name;
  c.kind = kind;
  c.periods = periods;
  c.link_field = link_field;
  t.cols.push_back(c);
  t.col_of[name] = (int)t.cols.size() - 1;
  return (int)t.cols.size() - 1;
}

void Store::freeze_layout() {
  for (int s = 0; s < SC_COUNT; ++s) {
    Table &t = tables[s];
    uint32_t tag_bytes = ((uint32_t)t.cols.size() + 7u) & ~7u;
    uint32_t off = tag_bytes;
    for (size_t i = 0; i < t.cols.size(); ++i) {
      t.cols[i].tag_index = (int)i;
      t.cols[i].val_off = (int)off;
      off += 8u * (uint32_t)(t.cols[i].kind == COL_PERIODIC ? std::max(1, t.cols[i].periods) : 1);
    }
    t.stride = std::max(16u, (off + 15u) & ~15u);
    // bucket rings of the periodic columns that have a write-path config
    t.ring_col_of.assign(t.cols.size(), -1);
    uint32_t roff = 0;
    for (size_t i = 0; i < t.cols.size(); ++i) {
      Column &c = t.cols[i];
      if (c.kind != COL_PERIODIC || c.period_ms <= 0 || c.offsets.empty()) continue;
      int maxo = 0;
      for (int o : c.offsets) maxo = std::max(maxo, o);
      if ((int)c.offsets.size() > RING_MAX_RANGES || maxo > 4095) continue;  // served by puts only
      c.ring_w = maxo + 1;
      c.ring_off = (int)roff;
      RingColDev d{};
      d.ring_off = roff;
      d.w = (uint32_t)c.ring_w;
      d.val_off = (uint32_t)c.val_off;
      d.tag_index = (uint32_t)c.tag_index;
      d.period_ms = c.period_ms;
      d.n_ranges = (int32_t)c.offsets.size();
      for (size_t k = 0; k < c.offsets.size(); ++k) d.offsets[k] = c.offsets[k];
      t.ring_col_of[i] = (int)t.ring_cols.size();
      t.ring_cols.push_back(d);
      roff += (uint32_t)c.ring_w * 16u;
    }
    t.ring_stride = roff;
  }
  frozen = true;
  // the global scope has exactly one instance
  slot(SC_GLOBAL, "", true);
}

uint32_t Store::intern(const std::string &s) {
  auto it = token_of.find(s);
  if (it != token_of.end()) return it->second;
  uint32_t id = (uint32_t)token_of.size() + 1;
  token_of.emplace(s, id);
  return id;
}

uint32_t Store::find_token(const std::string &s) const {
  auto it = token_of.find(s);
  return it == token_of.end() ? 0 : it->second;
}

uint32_t Store::slot(ScopeId scope, const char *id, size_t len, bool create) {
  Table &t = tables[scope];
  const uint32_t found = t.slot_of.find(id, len);
  if (found != SlotMap::NONE) return found;
  if (!create) return NO_SLOT;
  if (!frozen) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_config_load_json must be called before the store is used");
  uint32_t s = t.n_slots++;
  t.slot_of.insert(id, len, s);
  t.rows.resize((size_t)t.n_slots * t.stride, 0);
  t.mark(s);
  return s;
}

bool Store::split_key(const char *key, ScopeId &scope, std::string &id, std::string &feature) {
  if (!key) return false;
  // ScopeCodec.encode never emits '/', feature names may not either; ids may contain '/', so split on the LAST one?
  // Key.fromString (model/Key.scala:14-23) splits on the FIRST '/': mirror that.
  const char *slash = strchr(key, '/');
  if (!slash || slash == key) return false;
  std::string sc(key, slash);
  feature.assign(slash + 1);
  if (sc == "global") { scope = SC_GLOBAL; id.clear(); return true; }
  size_t eq = sc.find('=');
  if (eq == std::string::npos || eq == 0) return false;
  std::string left = sc.substr(0, eq);
  id = sc.substr(eq + 1);
  if (left == "item") scope = SC_ITEM;
  else if (left == "user") scope = SC_USER;
  else if (left == "session") scope = SC_SESSION;
  else if (left == "ranking") scope = SC_RANKING;
  else if (left == "field") scope = SC_FIELD;
  else if (left == "irf") scope = SC_IRF;
  else return false;
  return true;
}

bool Store::locate(const char *key, Cell &out) {
  ScopeId sc;
  std::string id, feature;
  if (!split_key(key, sc, id, feature))
    throw StatusError(MRK_ERR_INVALID_ARG, std::string("malformed key '") + (key ? key : "(null)") + "'");
  Table &t = tables[sc];
  auto it = t.col_of.find(feature);
  if (it == t.col_of.end()) return false;  // state of a feature this config does not use
  uint32_t s = slot(sc, id, true);
  out.t = &t;
  out.c = &t.cols[it->second];
  out.slot = s;
  out.rec = t.rows.data() + (size_t)s * t.stride;
  t.mark(s);
  ++version;
  return true;
}

static void kind_check(const Column *c, ColKind want, const char *key) {
  if (c->kind != want)
    throw StatusError(MRK_ERR_INVALID_ARG, std::string("value type does not match the state '") + c->name + "' (key " + key + ")");
}

bool Store::put_double(const char *key, double v) {
  Cell c;
  if (!locate(key, c)) return false;
  kind_check(c.c, COL_SCALAR, key);
  set_tag(c, TAG_DOUBLE);
  set_val(c, 0, v);
  return true;
}

bool Store::put_bool(const char *key, bool v) {
  Cell c;
  if (!locate(key, c)) return false;
  kind_check(c.c, COL_SCALAR, key);
  set_tag(c, TAG_BOOL);
  set_val(c, 0, v ? 1.0 : 0.0);
  return true;
}

bool Store::put_string(const char *key, const char *v) {
  if (!texts.empty()) {  // a cross-encoder column's item text
    ScopeId sc;
    std::string id, feature;
    if (split_key(key, sc, id, feature) && sc == SC_ITEM) {
      auto t = texts.find(feature);
      if (t != texts.end()) {
        if (!v) throw StatusError(MRK_ERR_INVALID_ARG, "null string value");
        ItemText &e = t->second[id];
        e.text = v;
        e.pieces.clear();
        e.tokenized = false;
        ++version;
        return true;
      }
    }
  }
  Cell c;
  if (!locate(key, c)) return false;
  kind_check(c.c, COL_SCALAR, key);
  if (!v) throw StatusError(MRK_ERR_INVALID_ARG, "null string value");
  uint32_t tok = intern(v);
  uint32_t link = 0;
  if (!c.c->link_field.empty()) {
    // second key hop of the item-field scoped rate, resolved now: ItemFieldScope(field, value)
    const std::string link_field = c.c->link_field;
    Table *t = c.t;
    uint32_t sl = c.slot;
    const Column *col = c.c;
    uint32_t fs = slot(SC_FIELD, link_field + ":" + v, true);  // may grow another table only
    link = fs + 1;
    c.rec = t->rows.data() + (size_t)sl * t->stride;
    c.c = const_cast<Column *>(col);
  }
  set_tag(c, TAG_STRING);
  c.c->max_len = std::max<uint32_t>(c.c->max_len, 1u);
  uint64_t cell = (uint64_t)tok | ((uint64_t)link << 32);
  set_val(c, 0, cell);
  return true;
}

bool Store::put_string_list(const char *key, const char *const *v, int n) {
  Cell c;
  if (!locate(key, c)) return false;
  kind_check(c.c, COL_SCALAR, key);
  if (n < 0 || (n > 0 && !v)) throw StatusError(MRK_ERR_INVALID_ARG, "bad string list");
  uint32_t off = n ? (uint32_t)tok_pool.host.size() : 0;
  for (int i = 0; i < n; ++i) tok_pool.host.push_back(intern(v[i] ? v[i] : ""));
  set_tag(c, TAG_STRING_LIST);
  c.c->max_len = std::max<uint32_t>(c.c->max_len, (uint32_t)n);
  uint64_t cell = (uint64_t)off | ((uint64_t)(uint32_t)n << 32);
  set_val(c, 0, cell);
  return true;
}

bool Store::put_double_list(const char *key, const double *v, int n) {
  Cell c;
  if (!locate(key, c)) return false;
  kind_check(c.c, COL_SCALAR, key);
  if (n < 0 || (n > 0 && !v)) throw StatusError(MRK_ERR_INVALID_ARG, "bad double list");
  uint32_t off = n ? (uint32_t)f64_pool.host.size() : 0;
  f64_pool.host.insert(f64_pool.host.end(), v, v + n);
  set_tag(c, TAG_DOUBLE_LIST);
  uint64_t cell = (uint64_t)off | ((uint64_t)(uint32_t)n << 32);
  set_val(c, 0, cell);
  return true;
}

bool Store::put_counter(const char *key, int64_t v) {
  Cell c;
  if (!locate(key, c)) return false;
  kind_check(c.c, COL_COUNTER, key);
  set_tag(c, TAG_PRESENT);
  set_val(c, 0, v);
  return true;
}

bool Store::put_periodic(const char *key, const int64_t *v, int n) {
  Cell c;
  if (!locate(key, c)) return false;
  kind_check(c.c, COL_PERIODIC, key);
  if (n < 0 || (n > 0 && !v)) throw StatusError(MRK_ERR_INVALID_ARG, "bad periodic counter");
  // the reference emits NaN when values.length != dim; keep the length in the tag, keep at most `periods` cells
  set_tag(c, (uint8_t)(1 + std::min(n, 250)));
  for (int i = 0; i < c.c->periods; ++i) set_val(c, i, i < n ? v[i] : (int64_t)0);
  return true;
}

bool Store::put_bounded_list(const char *key, const char *const *v, int n) {
  Cell c;
  if (!locate(key, c)) return false;
  kind_check(c.c, COL_BOUNDED_LIST, key);
  if (n < 0 || (n > 0 && !v)) throw StatusError(MRK_ERR_INVALID_ARG, "bad bounded list");
  Table *t = c.t;
  uint32_t sl = c.slot;
  Column *col = c.c;
  uint32_t off = n ? (uint32_t)slot_pool.host.size() : 0;
  for (int i = 0; i < n; ++i) slot_pool.host.push_back(slot(SC_ITEM, v[i] ? v[i] : "", true));  // may grow the item table
  c.rec = t->rows.data() + (size_t)sl * t->stride;  // t may BE the item table
  c.c = col;
  set_tag(c, TAG_PRESENT);
  uint64_t cell = (uint64_t)off | ((uint64_t)(uint32_t)n << 32);
  set_val(c, 0, cell);
  return true;
}

bool Store::erase(const char *key) {
  ScopeId sc;
  std::string id, feature;
  if (!split_key(key, sc, id, feature)) throw StatusError(MRK_ERR_INVALID_ARG, "malformed key");
  if (sc == SC_ITEM) {
    auto tx = texts.find(feature);
    if (tx != texts.end()) { ++version; return tx->second.erase(id) > 0; }
  }
  Table &t = tables[sc];
  auto it = t.col_of.find(feature);
  if (it == t.col_of.end()) return false;
  uint32_t s = slot(sc, id, false);
  if (s == NO_SLOT) return false;
  t.rows[(size_t)s * t.stride + t.cols[it->second].tag_index] = TAG_MISSING;
  t.mark(s);
  ++version;
  return true;
}

template <typename T>
static void flush_pool(Pool<T> &p, hipStream_t stream) {
  const size_t n = p.host.size();
  if (n > p.dev_cap) {
    size_t cap = std::max<size_t>(n + n / 2, 1024);
    p.dev.release();
    p.dev.reserve(cap * sizeof(T));
    p.dev_cap = cap;
    p.uploaded = 0;
  }
  if (n > p.uploaded) {
    MRK_HIP(hipMemcpyAsync((T *)p.dev.p + p.uploaded, p.host.data() + p.uploaded, (n - p.uploaded) * sizeof(T),
                           hipMemcpyHostToDevice, stream));
    p.uploaded = n;
  }
}

bool Store::dirty() const {
  bool d = !pending.empty() || tok_pool.host.size() > tok_pool.uploaded || f64_pool.host.size() > f64_pool.uploaded ||
           slot_pool.host.size() > slot_pool.uploaded;
  for (int s = 0; s < SC_COUNT && !d; ++s) {
    const Table &t = tables[s];
    d = t.dirty_hi > t.dirty_lo || t.n_slots > t.d_slots_cap ||
        (t.slot_of.mirrored && (t.slot_of.regrown || !t.slot_of.touched.empty() || t.slot_of.table.size() != t.d_id_entries));
  }
  return d;
}

// device mirror of an id -> slot map: the arena grows at its end; the table is re-sent whole after a rehash, else only
// the entries placed since the last flush
void Store::flush_ids(Table &t, hipStream_t stream) {
  SlotMap &m = t.slot_of;
  if (!m.mirrored) return;
  const size_t arena = m.ids.size();
  if (arena > t.d_arena_cap) {
    const size_t cap = std::max<size_t>(arena + arena / 2, 4096);
    t.d_id_arena.release();
    t.d_id_arena.reserve(cap);
    t.d_arena_cap = cap;
    t.d_arena_uploaded = 0;
  }
  if (arena > t.d_arena_uploaded) {
    MRK_HIP(hipMemcpyAsync((uint8_t *)t.d_id_arena.p + t.d_arena_uploaded, m.ids.data() + t.d_arena_uploaded, arena - t.d_arena_uploaded,
                           hipMemcpyHostToDevice, stream));
    t.d_arena_uploaded = arena;
  }
  const bool whole = m.regrown || m.table.size() != t.d_id_entries || m.touched.size() > 4096;
  if (whole) {
    if (m.table.size() != t.d_id_entries) {
      t.d_id_table.release();
      t.d_id_table.reserve(std::max<size_t>(m.table.size(), 1) * sizeof(IdEntry));
      t.d_id_entries = m.table.size();
    }
    if (!m.table.empty())
      MRK_HIP(hipMemcpyAsync(t.d_id_table.p, m.table.data(), m.table.size() * sizeof(IdEntry), hipMemcpyHostToDevice, stream));
  } else {
    for (uint32_t i : m.touched)
      MRK_HIP(hipMemcpyAsync((IdEntry *)t.d_id_table.p + i, &m.table[i], sizeof(IdEntry), hipMemcpyHostToDevice, stream));
  }
  m.regrown = false;
  m.touched.clear();
}

void Store::flush(hipStream_t stream) {
  // batches run on their own streams: nothing may be reading the tables / pools while they are rewritten or
  // reallocated.  Nothing dirty (the serving steady state between feedback events): no synchronisation at all.
  if (!dirty()) return;
  MRK_HIP(hipDeviceSynchronize());
  for (int s = 0; s < SC_COUNT; ++s) flush_ids(tables[s], stream);
  uint32_t uploaded_lo[SC_COUNT], uploaded_hi[SC_COUNT];
  for (int s = 0; s < SC_COUNT; ++s) {
    Table &t = tables[s];
    uploaded_lo[s] = uploaded_hi[s] = 0;
    if (t.n_slots > t.d_slots_cap) {
      uint32_t cap = std::max<uint32_t>(t.n_slots + t.n_slots / 2, 64);
      t.d_rows.release();
      t.d_rows.reserve((size_t)cap * t.stride);
      t.d_slots_cap = cap;
      t.dirty_lo = 0;
      t.dirty_hi = t.n_slots;
    }
    if (t.dirty_hi > t.dirty_lo) {
      uint32_t hi = std::min(t.dirty_hi, t.n_slots);
      MRK_HIP(hipMemcpyAsync((uint8_t *)t.d_rows.p + (size_t)t.dirty_lo * t.stride,
                             t.rows.data() + (size_t)t.dirty_lo * t.stride, (size_t)(hi - t.dirty_lo) * t.stride,
                             hipMemcpyHostToDevice, stream));
      uploaded_lo[s] = t.dirty_lo;
      uploaded_hi[s] = hi;
    }
    t.dirty_lo = UINT32_MAX;
    t.dirty_hi = 0;
  }
  flush_pool(tok_pool, stream);
  flush_pool(f64_pool, stream);
  flush_pool(slot_pool, stream);
  flush_writes(stream, uploaded_lo, uploaded_hi);
  // the host vectors may be reallocated by later puts: finish the copies before returning
  MRK_HIP(hipStreamSynchronize(stream));
}

// Write path: apply the staged PeriodicIncrements to the device bucket rings and recompute the window sums
// of every touched (slot, column); rows that were just re-uploaded from the host mirror (which knows
// nothing about ring-fed cells) get their ring-fed cells recomputed too.
void Store::flush_writes(hipStream_t stream, const uint32_t *uploaded_lo, const uint32_t *uploaded_hi) {
  std::sort(pending.begin(), pending.end(), [](const PendingInc &a, const PendingInc &b) {
    if (a.table != b.table) return a.table < b.table;
    if (a.slot != b.slot) return a.slot < b.slot;
    if (a.ring_col != b.ring_col) return a.ring_col < b.ring_col;
    return a.bucket < b.bucket;
  });
  size_t p = 0;
  for (int s = 0; s < SC_COUNT; ++s) {
    Table &t = tables[s];
    if (!t.ring_used || t.ring_stride == 0) continue;
    // grow the ring storage with the table (device-to-device copy keeps the buckets)
    if (t.d_slots_cap > t.d_ring_slots) {
      DevBuf nb;
      nb.reserve((size_t)t.d_slots_cap * t.ring_stride);
      MRK_HIP(hipMemsetAsync(nb.p, 0x80, (size_t)t.d_slots_cap * t.ring_stride, stream));
      if (t.d_ring_slots)
        MRK_HIP(hipMemcpyAsync(nb.p, t.d_ring.p, (size_t)t.d_ring_slots * t.ring_stride, hipMemcpyDeviceToDevice, stream));
      MRK_HIP(hipStreamSynchronize(stream));
      t.d_ring = std::move(nb);
      t.d_ring_slots = t.d_slots_cap;
      t.d_ring_cols.reserve(t.ring_cols.size() * sizeof(RingColDev));
      MRK_HIP(hipMemcpyAsync(t.d_ring_cols.p, t.ring_cols.data(), t.ring_cols.size() * sizeof(RingColDev), hipMemcpyHostToDevice, stream));
    }
    if (uploaded_hi[s] > uploaded_lo[s])
      launch_periodic_refresh(stream, (uint8_t *)t.d_rows.p, t.stride, (uint8_t *)t.d_ring.p, t.ring_stride,
                              (const RingColDev *)t.d_ring_cols.p, (int)t.ring_cols.size(), uploaded_lo[s], uploaded_hi[s]);
    std::vector<IncGroup> groups;
    std::vector<IncUpdate> ups;
    while (p < pending.size() && pending[p].table < s) ++p;
    while (p < pending.size() && pending[p].table == s) {
      IncGroup g{pending[p].slot, pending[p].ring_col, (uint32_t)ups.size(), 0};
      while (p < pending.size() && pending[p].table == s && pending[p].slot == g.slot && pending[p].ring_col == g.col) {
        if (!ups.empty() && ups.size() > g.begin && ups.back().bucket == pending[p].bucket) ups.back().inc += pending[p].inc;
        else ups.push_back(IncUpdate{pending[p].bucket, pending[p].inc});
        ++p;
      }
      g.end = (uint32_t)ups.size();
      groups.push_back(g);
    }
    if (!groups.empty()) {
      d_groups.reserve(groups.size() * sizeof(IncGroup));
      d_updates.reserve(ups.size() * sizeof(IncUpdate));
      MRK_HIP(hipMemcpyAsync(d_groups.p, groups.data(), groups.size() * sizeof(IncGroup), hipMemcpyHostToDevice, stream));
      MRK_HIP(hipMemcpyAsync(d_updates.p, ups.data(), ups.size() * sizeof(IncUpdate), hipMemcpyHostToDevice, stream));
      launch_periodic_apply(stream, (uint8_t *)t.d_rows.p, t.stride, (uint8_t *)t.d_ring.p, t.ring_stride,
                            (const RingColDev *)t.d_ring_cols.p, groups.data() ? (const IncGroup *)d_groups.p : nullptr, (int)groups.size(),
                            (const IncUpdate *)d_updates.p);
      MRK_HIP(hipStreamSynchronize(stream));  // groups / ups are reused by the next table
    }
  }
  pending.clear();
}

void Store::set_periodic_config(ScopeId scope, const std::string &name, int64_t period_ms, const std::vector<int32_t> &offsets) {
  Table &t = tables[scope];
  auto it = t.col_of.find(name);
  if (it == t.col_of.end()) return;
  t.cols[it->second].period_ms = period_ms;
  t.cols[it->second].offsets = offsets;
}

void Store::set_list_config(ScopeId scope, const std::string &name, int64_t count, int64_t duration_ms) {
  Table &t = tables[scope];
  auto it = t.col_of.find(name);
  if (it == t.col_of.end()) return;
  t.cols[it->second].list_count = count;
  t.cols[it->second].list_duration_ms = duration_ms;
}

bool Store::increment_periodic(const char *key, int64_t ts_ms, int64_t inc) {
  Cell c;
  if (!locate(key, c)) return false;
  kind_check(c.c, COL_PERIODIC, key);
  const int col = (int)(c.c - c.t->cols.data());
  const int rc = c.t->ring_col_of.empty() ? -1 : c.t->ring_col_of[col];
  if (rc < 0) throw StatusError(MRK_ERR_UNSUPPORTED, std::string("state '") + c.c->name + "' has no bucket / periods the device write path supports");
  // Timestamp.toStartOfPeriod (model/Timestamp.scala:18-21): floor(ts.toDouble / period.toMillis).toLong * period.toMillis
  const int64_t bucket = (int64_t)std::floor((double)ts_ms / (double)c.c->period_ms) * c.c->period_ms;
  pending.push_back(PendingInc{(uint8_t)c.t->scope, c.slot, (uint32_t)rc, bucket, inc});
  c.t->ring_used = true;
  return true;
}

bool Store::increment(const char *key, int64_t inc) {
  Cell c;
  if (!locate(key, c)) return false;
  kind_check(c.c, COL_COUNTER, key);
  // MemCounter.put (fstore/memory/MemCounter.scala): existing + inc, else inc
  int64_t cur = 0;
  if (c.rec[c.c->tag_index] != TAG_MISSING) memcpy(&cur, c.rec + c.c->val_off, 8);
  set_tag(c, TAG_PRESENT);
  set_val(c, 0, (int64_t)((uint64_t)cur + (uint64_t)inc));
  return true;
}

bool Store::append(const char *key, const char *value, int64_t ts_ms) {
  {
    Cell c;
    if (!locate(key, c)) return false;
    kind_check(c.c, COL_BOUNDED_LIST, key);
    if (!value) throw StatusError(MRK_ERR_INVALID_ARG, "null list element");
    // MemBoundedList.put (fstore/memory/MemBoundedList.scala:18-37): the first element is stored as is; later ones
    // are prepended, then everything older than (this ts - duration) is dropped and `count` elements are kept
    auto &lst = lists[key];
    const int64_t count = c.c->list_count, dur = c.c->list_duration_ms;
    if (lst.empty()) {
      lst.emplace_back(ts_ms, value);
    } else {
      lst.insert(lst.begin(), std::make_pair(ts_ms, std::string(value)));
      const int64_t cutoff = dur == INT64_MAX ? INT64_MIN : ts_ms - dur;
      std::vector<std::pair<int64_t, std::string>> kept;
      for (auto &e : lst)
        if (e.first >= cutoff && (int64_t)kept.size() < count) kept.push_back(std::move(e));
      lst.swap(kept);
    }
  }
  auto &lst = lists[key];
  std::vector<const char *> ptrs;
  for (auto &e : lst) ptrs.push_back(e.second.c_str());
  return put_bounded_list(key, ptrs.data(), (int)ptrs.size());
}

StoreDev Store::device_view() const {
  StoreDev d{};
  for (int s = 0; s < SC_COUNT; ++s) {
    d.tab[s].rows = (const uint8_t *)tables[s].d_rows.p;
    d.tab[s].stride = tables[s].stride;
    d.tab[s].n_slots = tables[s].n_slots;
  }
  d.tok_pool = (const uint32_t *)tok_pool.dev.p;
  d.f64_pool = (const double *)f64_pool.dev.p;
  d.slot_pool = (const uint32_t *)slot_pool.dev.p;
  return d;
}

size_t Store::device_bytes() const {
  size_t b = 0;
  for (int s = 0; s < SC_COUNT; ++s) b += (size_t)tables[s].n_slots * tables[s].stride;
  return b + tok_pool.host.size() * 4 + f64_pool.host.size() * 8 + slot_pool.host.size() * 4;
}

}  // namespace mrk
