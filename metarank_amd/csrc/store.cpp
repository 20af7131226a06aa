// Host side of the device feature store (see store.hpp for the HBM layout).
#include "store.hpp"

#include <algorithm>
#include <cstring>

namespace mrk {

Store::Store() {
  for (int s = 0; s < SC_COUNT; ++s) tables[s].scope = (ScopeId)s;
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

uint32_t Store::slot(ScopeId scope, const std::string &id, bool create) {
  Table &t = tables[scope];
  auto it = t.slot_of.find(id);
  if (it != t.slot_of.end()) return it->second;
  if (!create) return NO_SLOT;
  if (!frozen) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_config_load_json must be called before the store is used");
  uint32_t s = t.n_slots++;
  t.slot_of.emplace(id, s);
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

void Store::flush(hipStream_t stream) {
  for (int s = 0; s < SC_COUNT; ++s) {
    Table &t = tables[s];
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
    }
    t.dirty_lo = UINT32_MAX;
    t.dirty_hi = 0;
  }
  flush_pool(tok_pool, stream);
  flush_pool(f64_pool, stream);
  flush_pool(slot_pool, stream);
  // the host vectors may be reallocated by later puts: finish the copies before returning
  MRK_HIP(hipStreamSynchronize(stream));
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
