// Host side of the device feature store (see store.hpp for the HBM layout).
#include "store.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace mrk {

Store::Store() {
  for (int s = 0; s < SC_COUNT; ++s) tables[s].scope = (ScopeId)s;
  tables[SC_ITEM].slot_of.mirrored = true;  // item ids are resolved on the device (resolve.hip)
  tok_pool.host.push_back(0);   // offset 0 is never handed out: {0, 0} == the empty list
  f64_pool.host.push_back(0.0);
  slot_pool.host.push_back(0);
  f32_pool.host.assign(4, 0.f);
  f32_pool.align = 4;
  f32_pool.dirty.emplace_back(0, 4);
  tok_pool.dirty.emplace_back(0, 1);
  f64_pool.dirty.emplace_back(0, 1);
  slot_pool.dirty.emplace_back(0, 1);
}

int Store::add_column(ScopeId scope, const std::string &name, ColKind kind, int periods, const std::string &link_field, bool expect_list) {
  if (frozen) throw StatusError(MRK_ERR_INVALID_ARG, "store layout is frozen");
  Table &t = tables[scope];
  auto it = t.col_of.find(name);
  if (it != t.col_of.end()) {
    Column &c = t.cols[it->second];
    if (c.kind != kind || c.periods != periods)
      throw StatusError(MRK_ERR_INVALID_ARG, "state '" + name + "' is declared twice with different types");
    if (!link_field.empty()) c.link_field = link_field;
    c.expect_list = c.expect_list || expect_list;
    return it->second;
  }
  Column c;
  c.name = name;
  c.kind = kind;
  c.periods = periods;
  c.link_field = link_field;
  c.expect_list = expect_list;
  t.cols.push_back(c);
  t.col_of[name] = (int)t.cols.size() - 1;
  return (int)t.cols.size() - 1;
}

void Store::freeze_layout() {
  for (int s = 0; s < SC_COUNT; ++s) {
    Table &t = tables[s];
    // [tags][u16 heap_used] padded to 8, then the value cells, then the inline heap up to the end of the last line
    t.heap_used_off = (uint32_t)t.cols.size();
    t.heap_used_off = (t.heap_used_off + 1u) & ~1u;
    uint32_t off = (t.heap_used_off + 2u + 7u) & ~7u;
    uint32_t list_cols = 0;
    for (size_t i = 0; i < t.cols.size(); ++i) {
      t.cols[i].tag_index = (int)i;
      t.cols[i].val_off = (int)off;
      off += 8u * (uint32_t)(t.cols[i].kind == COL_PERIODIC ? std::max(1, t.cols[i].periods) : 1);
      if (t.cols[i].expect_list) ++list_cols;
    }
    // records start on a 128-byte line and end on one (64 for records that fit half a line); the heap gets what the
    // declared list columns are expected to need (12 bytes each = 3 tokens on average) plus the padding
    const uint32_t want = off + 12u * list_cols;
    t.stride = want <= 64 ? 64u : (want + 127u) & ~127u;
    t.heap_off = off;
    t.heap_cap = std::min<uint32_t>(t.stride - off, 0xfff0u);
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
  if (memchr(id, 0, len)) throw StatusError(MRK_ERR_INVALID_ARG, "ids containing U+0000 are not supported");
  uint32_t s = t.n_slots++;
  t.slot_of.insert(id, len, s);
  t.rows.resize((size_t)t.n_slots * t.stride, 0);
  t.mark(s);
  return s;
}

bool Store::parse_key(const char *key, KeyRef &out) {
  if (!key) return false;
  // Key.fromString (model/Key.scala:14-23) splits on the FIRST '/': mirror that
  const char *slash = strchr(key, '/');
  if (!slash || slash == key) return false;
  const std::string_view sc(key, (size_t)(slash - key));
  out.feature = std::string_view(slash + 1);
  if (sc == "global") { out.scope = SC_GLOBAL; out.id = std::string_view(); return true; }
  const size_t eq = sc.find('=');
  if (eq == std::string_view::npos || eq == 0) return false;
  const std::string_view left = sc.substr(0, eq);
  out.id = sc.substr(eq + 1);
  if (left == "item") out.scope = SC_ITEM;
  else if (left == "user") out.scope = SC_USER;
  else if (left == "session") out.scope = SC_SESSION;
  else if (left == "ranking") out.scope = SC_RANKING;
  else if (left == "field") out.scope = SC_FIELD;
  else if (left == "irf") out.scope = SC_IRF;
  else return false;
  return true;
}

KeyRef Store::need_key(const char *key) {
  KeyRef k;
  if (!parse_key(key, k)) throw StatusError(MRK_ERR_INVALID_ARG, std::string("malformed key '") + (key ? key : "(null)") + "'");
  return k;
}

bool Store::locate(const KeyRef &k, Cell &out) {
  Table &t = tables[k.scope];
  auto it = t.col_of.find(std::string(k.feature));
  if (it == t.col_of.end()) return false;  // state of a feature this config does not use
  uint32_t s = slot(k.scope, k.id.data() ? k.id.data() : "", k.id.size(), true);
  out.t = &t;
  out.c = &t.cols[it->second];
  out.slot = s;
  out.rec = t.rows.data() + (size_t)s * t.stride;
  t.mark(s);
  ++version;
  if (!ttl_deadline.empty()) ttl_deadline.erase(ttl_cell(k.scope, (uint32_t)it->second, s));   // a fresh write: the old deadline is void (ttl_note sets the new one)
  return true;
}

static void kind_check(const Column *c, ColKind want, const KeyRef &k) {
  if (c->kind != want)
    throw StatusError(MRK_ERR_INVALID_ARG, std::string("value type does not match the state '") + c->name + "' (scope id " + std::string(k.id) + ")");
}

// the pool range a cell's CURRENT value holds goes back to its free list (inline heap bytes are reclaimed by the next
// re-pack of the record); keep_tok_range: the caller is about to store a new string list and handles the old one itself
void Store::drop_value(Cell &c, bool keep_tok_range) {
  const uint8_t tag = c.rec[c.c->tag_index];
  if (tag == TAG_MISSING) return;
  uint64_t cell;
  memcpy(&cell, c.rec + c.c->val_off, 8);
  const uint32_t off = (uint32_t)cell, len = (uint32_t)(cell >> 32);
  if (c.c->kind == COL_SCALAR) {
    if (tag == TAG_STRING_LIST && !keep_tok_range && !(off & LIST_INLINE)) tok_pool.release(off, len);
    else if (tag == TAG_DOUBLE_LIST) { if (off & LIST_F32_BIT) f32_pool.release(off & ~LIST_F32_BIT, len); else f64_pool.release(off, len); }
  } else if (c.c->kind == COL_BOUNDED_LIST) {
    slot_pool.release(off, len);
  }
}

bool Store::put_double(const KeyRef &k, double v) {
  Cell c;
  if (!locate(k, c)) return false;
  kind_check(c.c, COL_SCALAR, k);
  drop_value(c);
  set_tag(c, TAG_DOUBLE);
  set_val(c, 0, v);
  return true;
}

bool Store::put_bool(const KeyRef &k, bool v) {
  Cell c;
  if (!locate(k, c)) return false;
  kind_check(c.c, COL_SCALAR, k);
  drop_value(c);
  set_tag(c, TAG_BOOL);
  set_val(c, 0, v ? 1.0 : 0.0);
  return true;
}

bool Store::put_string(const char *key, const char *v) {
  if (!v) throw StatusError(MRK_ERR_INVALID_ARG, "null string value");
  return put_string(need_key(key), std::string_view(v));
}

bool Store::put_string(const KeyRef &k, std::string_view v) {
  if (!texts.empty() && k.scope == SC_ITEM) {  // a cross-encoder column's item text
    auto t = texts.find(std::string(k.feature));
    if (t != texts.end()) {
      ItemText &e = t->second[std::string(k.id)];
      e.text = std::string(v);
      e.pieces.clear();
      e.tokenized = false;
      ++version;
      return true;
    }
  }
  Cell c;
  if (!locate(k, c)) return false;
  kind_check(c.c, COL_SCALAR, k);
  uint32_t tok = intern(std::string(v));
  uint32_t link = 0;
  if (!c.c->link_field.empty()) {
    // second key hop of the item-field scoped rate, resolved now: ItemFieldScope(field, value)
    Table *t = c.t;
    const uint32_t sl = c.slot;
    const std::string fid = c.c->link_field + ":" + std::string(v);
    uint32_t fs = slot(SC_FIELD, fid, true);  // may grow another table only
    link = fs + 1;
    c.rec = t->rows.data() + (size_t)sl * t->stride;
  }
  drop_value(c);
  set_tag(c, TAG_STRING);
  c.c->max_len = std::max<uint32_t>(c.c->max_len, 1u);
  uint64_t cell = (uint64_t)tok | ((uint64_t)link << 32);
  set_val(c, 0, cell);
  return true;
}

// Places `n` tokens of column `col` in the record's inline heap; false when they do not fit even after the heap has
// been re-packed (then the caller uses the pool).  The heap is a bump allocator: an in-place rewrite when the column's
// old inline list was at least as long, else the tokens go to the end; when the end is reached the live lists are
// compacted (in column order) once.
bool Store::heap_place(Table &t, uint8_t *rec, const Column *col, const uint32_t *toks, uint32_t n, uint32_t &off_out) {
  const uint32_t need = n * 4u;
  if (need > t.heap_cap) return false;
  uint16_t used;
  memcpy(&used, rec + t.heap_used_off, 2);
  uint64_t cur;
  memcpy(&cur, rec + col->val_off, 8);
  const bool cur_inline = rec[col->tag_index] == TAG_STRING_LIST && ((uint32_t)cur & LIST_INLINE);
  if (cur_inline && (uint32_t)(cur >> 32) >= n) {  // rewrite in place
    off_out = (uint32_t)cur & ~LIST_INLINE;
    memcpy(rec + off_out, toks, need);
    return true;
  }
  if ((uint32_t)used + need > t.heap_cap) {
    // compact: every OTHER live inline list, in column order (this column's old list is dropped)
    std::vector<uint32_t> tmp;
    std::vector<std::pair<const Column *, std::pair<uint32_t, uint32_t>>> live;
    for (const Column &c2 : t.cols) {
      if (c2.kind != COL_SCALAR || &c2 == col || rec[c2.tag_index] != TAG_STRING_LIST) continue;
      uint64_t cell;
      memcpy(&cell, rec + c2.val_off, 8);
      if (!((uint32_t)cell & LIST_INLINE)) continue;
      const uint32_t o = (uint32_t)cell & ~LIST_INLINE, l = (uint32_t)(cell >> 32);
      live.push_back({&c2, {(uint32_t)tmp.size(), l}});
      tmp.resize(tmp.size() + l);
      memcpy(tmp.data() + tmp.size() - l, rec + o, (size_t)l * 4);
    }
    uint32_t at = 0;
    for (auto &e : live) {
      const uint32_t l = e.second.second;
      memcpy(rec + t.heap_off + at, tmp.data() + e.second.first, (size_t)l * 4);
      const uint64_t cell = (uint64_t)((t.heap_off + at) | LIST_INLINE) | ((uint64_t)l << 32);
      memcpy(rec + e.first->val_off, &cell, 8);
      at += l * 4u;
    }
    used = (uint16_t)at;
    if ((uint32_t)used + need > t.heap_cap) {
      memcpy(rec + t.heap_used_off, &used, 2);
      return false;
    }
  }
  off_out = t.heap_off + used;
  memcpy(rec + off_out, toks, need);
  used = (uint16_t)(used + need);
  memcpy(rec + t.heap_used_off, &used, 2);
  return true;
}

// stores a string list (already interned) into the cell: inline when the record's heap has room, else in the pool
void Store::put_tokens(Cell &c, const uint32_t *toks, uint32_t n) {
  uint64_t old = 0;
  const bool had_list = c.rec[c.c->tag_index] == TAG_STRING_LIST;
  if (had_list) memcpy(&old, c.rec + c.c->val_off, 8);
  else drop_value(c);
  const uint32_t old_off = (uint32_t)old, old_len = (uint32_t)(old >> 32);
  const bool old_pool = had_list && !(old_off & LIST_INLINE);
  uint32_t off = 0;
  if (n == 0) {
    if (old_pool) tok_pool.release(old_off, old_len);
  } else if (heap_place(*c.t, c.rec, c.c, toks, n, off)) {
    if (old_pool) tok_pool.release(old_off, old_len);
    off |= LIST_INLINE;
  } else {
    off = tok_pool.realloc(old_pool ? old_off : 0, old_pool ? old_len : 0, n);
    tok_pool.write(off, toks, n);
  }
  set_tag(c, TAG_STRING_LIST);
  c.c->max_len = std::max<uint32_t>(c.c->max_len, n);
  const uint64_t cell = (uint64_t)off | ((uint64_t)n << 32);
  set_val(c, 0, cell);
}

bool Store::put_string_list(const char *key, const char *const *v, int n) {
  if (n < 0 || (n > 0 && !v)) throw StatusError(MRK_ERR_INVALID_ARG, "bad string list");
  std::vector<std::string_view> sv((size_t)n);
  for (int i = 0; i < n; ++i) sv[(size_t)i] = v[i] ? std::string_view(v[i]) : std::string_view("");
  return put_string_list(need_key(key), sv.data(), n);
}

bool Store::put_string_list(const KeyRef &k, const std::string_view *v, int n) {
  Cell c;
  if (!locate(k, c)) return false;
  kind_check(c.c, COL_SCALAR, k);
  if (n < 0 || (n > 0 && !v)) throw StatusError(MRK_ERR_INVALID_ARG, "bad string list");
  std::vector<uint32_t> toks((size_t)n);
  for (int i = 0; i < n; ++i) toks[(size_t)i] = intern(std::string(v[i]));
  put_tokens(c, toks.data(), (uint32_t)n);
  return true;
}

bool Store::put_double_list(const KeyRef &k, const double *v, int n) {
  Cell c;
  if (!locate(k, c)) return false;
  kind_check(c.c, COL_SCALAR, k);
  if (n < 0 || (n > 0 && !v)) throw StatusError(MRK_ERR_INVALID_ARG, "bad double list");
  // long lists whose values are all exactly floats (embeddings: f32 widened at ingest) are kept as f32
  bool as_f32 = (uint32_t)n >= LIST_F32_MIN;
  for (int i = 0; i < n && as_f32; ++i) as_f32 = (double)(float)v[i] == v[i] || v[i] != v[i];
  for (int i = 0; i < n && as_f32; ++i) as_f32 = v[i] == v[i];   // (a NaN's payload would not survive: keep such lists f64)
  uint64_t old = 0;
  const bool had = c.rec[c.c->tag_index] == TAG_DOUBLE_LIST;
  if (had) memcpy(&old, c.rec + c.c->val_off, 8);
  else drop_value(c);
  const uint32_t old_off = (uint32_t)old, old_len = (uint32_t)(old >> 32);
  const bool old_f32 = had && (old_off & LIST_F32_BIT);
  uint32_t off;
  if (as_f32) {
    if (had && !old_f32) f64_pool.release(old_off, old_len);
    off = f32_pool.realloc(old_f32 ? old_off & ~LIST_F32_BIT : 0, old_f32 ? old_len : 0, (uint32_t)n);
    std::vector<float> f((size_t)n);
    for (int i = 0; i < n; ++i) f[(size_t)i] = (float)v[i];
    f32_pool.write(off, f.data(), (uint32_t)n);
    off |= LIST_F32_BIT;
  } else {
    if (old_f32) f32_pool.release(old_off & ~LIST_F32_BIT, old_len);
    off = f64_pool.realloc(had && !old_f32 ? old_off : 0, had && !old_f32 ? old_len : 0, (uint32_t)n);
    f64_pool.write(off, v, (uint32_t)n);
  }
  set_tag(c, TAG_DOUBLE_LIST);
  uint64_t cell = (uint64_t)off | ((uint64_t)(uint32_t)n << 32);
  set_val(c, 0, cell);
  return true;
}

bool Store::put_counter(const KeyRef &k, int64_t v) {
  Cell c;
  if (!locate(k, c)) return false;
  kind_check(c.c, COL_COUNTER, k);
  set_tag(c, TAG_PRESENT);
  set_val(c, 0, v);
  return true;
}

bool Store::put_periodic(const KeyRef &k, const int64_t *v, int n) {
  Cell c;
  if (!locate(k, c)) return false;
  kind_check(c.c, COL_PERIODIC, k);
  if (n < 0 || (n > 0 && !v)) throw StatusError(MRK_ERR_INVALID_ARG, "bad periodic counter");
  // the reference emits NaN when values.length != dim; keep the length in the tag, keep at most `periods` cells
  set_tag(c, (uint8_t)(1 + std::min(n, 250)));
  for (int i = 0; i < c.c->periods; ++i) set_val(c, i, i < n ? v[i] : (int64_t)0);
  return true;
}

bool Store::put_bounded_list(const char *key, const char *const *v, int n) {
  if (n < 0 || (n > 0 && !v)) throw StatusError(MRK_ERR_INVALID_ARG, "bad bounded list");
  std::vector<std::string_view> sv((size_t)n);
  for (int i = 0; i < n; ++i) sv[(size_t)i] = v[i] ? std::string_view(v[i]) : std::string_view("");
  return put_bounded_list(need_key(key), sv.data(), n);
}

bool Store::put_bounded_list(const KeyRef &k, const std::string_view *v, int n) {
  Cell c;
  if (!locate(k, c)) return false;
  kind_check(c.c, COL_BOUNDED_LIST, k);
  if (n < 0 || (n > 0 && !v)) throw StatusError(MRK_ERR_INVALID_ARG, "bad bounded list");
  Table *t = c.t;
  const uint32_t sl = c.slot;
  std::vector<uint32_t> slots((size_t)n);
  for (int i = 0; i < n; ++i) slots[(size_t)i] = slot(SC_ITEM, v[i].data() ? v[i].data() : "", v[i].size(), true);  // may grow the item table
  c.rec = t->rows.data() + (size_t)sl * t->stride;  // t may BE the item table
  uint64_t old = 0;
  if (c.rec[c.c->tag_index] != TAG_MISSING) memcpy(&old, c.rec + c.c->val_off, 8);
  const uint32_t off = slot_pool.realloc((uint32_t)old, (uint32_t)(old >> 32), (uint32_t)n);
  slot_pool.write(off, slots.data(), (uint32_t)n);
  set_tag(c, TAG_PRESENT);
  uint64_t cell = (uint64_t)off | ((uint64_t)(uint32_t)n << 32);
  set_val(c, 0, cell);
  return true;
}

bool Store::erase(const KeyRef &k) {
  if (k.scope == SC_ITEM) {
    auto tx = texts.find(std::string(k.feature));
    if (tx != texts.end()) { ++version; return tx->second.erase(std::string(k.id)) > 0; }
  }
  Table &t = tables[k.scope];
  auto it = t.col_of.find(std::string(k.feature));
  if (it == t.col_of.end()) return false;
  uint32_t s = slot(k.scope, k.id.data() ? k.id.data() : "", k.id.size(), false);
  if (s == NO_SLOT) return false;
  Cell c{&t, &t.cols[it->second], s, t.rows.data() + (size_t)s * t.stride};
  drop_value(c);
  set_tag(c, TAG_MISSING);
  t.mark(s);
  ++version;
  return true;
}

void Store::ttl_note(const KeyRef &k, int64_t deadline_ms) {
  Table &t = tables[k.scope];
  auto it = t.col_of.find(std::string(k.feature));
  if (it == t.col_of.end()) return;                    // a feature this config does not use: nothing was stored
  const uint32_t s = slot(k.scope, k.id.data() ? k.id.data() : "", k.id.size(), false);
  if (s == NO_SLOT) return;
  const uint64_t cell = ttl_cell(k.scope, (uint32_t)it->second, s);
  ttl_deadline[cell] = deadline_ms;
  ttl_heap.emplace(deadline_ms, cell);
  // a rewrite leaves the cell's previous pair in the heap until ITS deadline passes (90 days by default): a hot cell rewritten
  // a hundred times a second would grow the heap by 8 * 10^8 pairs.  Rebuild from the live deadlines once stale pairs outnumber them.
  if (ttl_heap.size() > 2 * ttl_deadline.size() + 1024) {
    std::vector<std::pair<int64_t, uint64_t>> live;
    live.reserve(ttl_deadline.size());
    for (const auto &kv : ttl_deadline) live.emplace_back(kv.second, kv.first);
    ttl_heap = decltype(ttl_heap)(std::greater<std::pair<int64_t, uint64_t>>(), std::move(live));
  }
}

int64_t Store::ttl_expire(int64_t now_ms) {
  int64_t n = 0;
  while (!ttl_heap.empty() && ttl_heap.top().first <= now_ms) {
    const auto [deadline, cell] = ttl_heap.top();
    ttl_heap.pop();
    auto it = ttl_deadline.find(cell);
    if (it == ttl_deadline.end() || it->second != deadline) continue;   // rewritten since (a later deadline is in the heap) or cleared
    ttl_deadline.erase(it);
    const ScopeId scope = (ScopeId)(cell >> 58);
    const uint32_t col = (uint32_t)((cell >> 32) & 0x3ffffffu), s = (uint32_t)cell;
    Table &t = tables[scope];
    if (col >= t.cols.size() || s >= t.n_slots) continue;
    Cell c{&t, &t.cols[col], s, t.rows.data() + (size_t)s * t.stride};
    if (c.rec[c.c->tag_index] == TAG_MISSING) continue;
    drop_value(c);
    set_tag(c, TAG_MISSING);
    t.mark(s);
    ++version;
    ++n;
  }
  return n;
}

uint32_t Store::clone_items(int copies) {
  Table &t = tables[SC_ITEM];
  const uint32_t n0 = t.n_slots;
  if (copies <= 0 || n0 == 0) return n0;
  if ((uint64_t)n0 * (uint64_t)(copies + 1) > 0x7fffffffull) throw StatusError(MRK_ERR_INVALID_ARG, "too many clones");
  // the original ids, in slot order (slot s's id is the s-th string of the arena)
  std::vector<std::string> ids;
  ids.reserve(n0);
  for (size_t at = 0; ids.size() < n0;) {
    const char *p = t.slot_of.ids.data() + at;
    ids.emplace_back(p);
    at += ids.back().size() + 1;
  }
  std::vector<uint8_t> rec(t.stride);
  for (int k = 1; k <= copies; ++k) {
    const std::string suffix = "#" + std::to_string(k);
    for (uint32_t s0 = 0; s0 < n0; ++s0) {
      const std::string id = ids[s0] + suffix;
      const uint32_t s = slot(SC_ITEM, id, true);
      memcpy(rec.data(), t.rows.data() + (size_t)s0 * t.stride, t.stride);
      for (const Column &c : t.cols) {  // values that live in a pool get their own range
        const uint8_t tag = rec[c.tag_index];
        if (tag == TAG_MISSING) continue;
        uint64_t cell;
        memcpy(&cell, rec.data() + c.val_off, 8);
        const uint32_t off = (uint32_t)cell, len = (uint32_t)(cell >> 32);
        uint32_t noff = off;
        if (c.kind == COL_SCALAR && tag == TAG_STRING_LIST && !(off & LIST_INLINE) && len) {
          noff = tok_pool.alloc(len);
          std::vector<uint32_t> v(tok_pool.host.begin() + off, tok_pool.host.begin() + off + len);
          tok_pool.write(noff, v.data(), len);
        } else if (c.kind == COL_SCALAR && tag == TAG_DOUBLE_LIST && len && (off & LIST_F32_BIT)) {
          const uint32_t o = off & ~LIST_F32_BIT;
          noff = f32_pool.alloc(len);
          std::vector<float> v(f32_pool.host.begin() + o, f32_pool.host.begin() + o + len);
          f32_pool.write(noff, v.data(), len);
          noff |= LIST_F32_BIT;
        } else if (c.kind == COL_SCALAR && tag == TAG_DOUBLE_LIST && len) {
          noff = f64_pool.alloc(len);
          std::vector<double> v(f64_pool.host.begin() + off, f64_pool.host.begin() + off + len);
          f64_pool.write(noff, v.data(), len);
        } else if (c.kind == COL_BOUNDED_LIST && len) {
          noff = slot_pool.alloc(len);
          std::vector<uint32_t> v(slot_pool.host.begin() + off, slot_pool.host.begin() + off + len);
          slot_pool.write(noff, v.data(), len);
        } else {
          continue;
        }
        cell = (uint64_t)noff | ((uint64_t)len << 32);
        memcpy(rec.data() + c.val_off, &cell, 8);
      }
      memcpy(t.rows.data() + (size_t)s * t.stride, rec.data(), t.stride);
    }
  }
  ++version;
  return t.n_slots;
}

template <typename T>
static void flush_pool(Pool<T> &p, hipStream_t stream) {
  const size_t n = p.host.size();
  if (n > p.dev_cap) {
    size_t cap = std::max<size_t>(n + n / 2, 1024);
    p.dev.release();
    p.dev.reserve(cap * sizeof(T));
    p.dev_cap = cap;
    p.dirty.clear();
    p.dirty.emplace_back(0, n);
  }
  if (p.dirty.size() > 256) {  // many scattered rewrites: one copy of their hull
    size_t lo = n, hi = 0;
    for (auto &d : p.dirty) { lo = std::min(lo, d.first); hi = std::max(hi, d.second); }
    p.dirty.clear();
    p.dirty.emplace_back(lo, hi);
  }
  for (auto &d : p.dirty)
    if (d.second > d.first)
      MRK_HIP(hipMemcpyAsync((T *)p.dev.p + d.first, p.host.data() + d.first, (d.second - d.first) * sizeof(T), hipMemcpyHostToDevice, stream));
  p.dirty.clear();
}

bool Store::dirty() const {
  bool d = !pending.empty() || !tok_pool.dirty.empty() || !f64_pool.dirty.empty() || !slot_pool.dirty.empty() || !f32_pool.dirty.empty();
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
  flush_pool(f32_pool, stream);
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
  const KeyRef k = need_key(key);
  if (!locate(k, c)) return false;
  kind_check(c.c, COL_PERIODIC, k);
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
  const KeyRef k = need_key(key);
  if (!locate(k, c)) return false;
  kind_check(c.c, COL_COUNTER, k);
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
    const KeyRef k = need_key(key);
    if (!locate(k, c)) return false;
    kind_check(c.c, COL_BOUNDED_LIST, k);
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
  d.f32_pool = (const float *)f32_pool.dev.p;
  return d;
}

size_t Store::device_bytes() const {
  size_t b = 0;
  for (int s = 0; s < SC_COUNT; ++s) b += (size_t)tables[s].n_slots * tables[s].stride;
  return b + tok_pool.host.size() * 4 + f64_pool.host.size() * 8 + slot_pool.host.size() * 4 + f32_pool.host.size() * 4;
}

}  // namespace mrk
