// Feature registry + host half of a request.  See features.hpp.
#include "features.hpp"
#include "tzif.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <exception>
#include <set>
#include <string_view>
#include <thread>

#include "encoder.hpp"
#include "jit.hpp"
#include "json.hpp"

namespace mrk {

namespace {

[[noreturn]] void bad(const std::string &msg) { throw StatusError(MRK_ERR_PARSE, msg); }

const double kNaN = std::numeric_limits<double>::quiet_NaN();

// ---- FieldName (model/FieldName.scala:38-58) and ScopeType (model/ScopeType.scala:99-110) decoders
struct FieldName {
  std::string event;  // item | user | ranking | * | interaction:<type>
  std::string field;
};

bool is_ident(const std::string &s, bool dash) {
  if (s.empty()) return false;
  for (char c : s)
    if (!(isalnum((unsigned char)c) || c == '_' || (dash && c == '-'))) return false;
  return true;
}

FieldName parse_field_name(const std::string &s) {
  size_t dot = s.rfind('.');
  if (dot == std::string::npos || dot == 0) bad("cannot decode source field '" + s + "': it should have a format of <type>.<name>, like item.title, but the delimiter was not found.");
  std::string src = s.substr(0, dot), field = s.substr(dot + 1);
  if (!is_ident(field, false)) bad("cannot decode source field '" + s + "'");
  FieldName out;
  out.field = field;
  if (src.rfind("interaction:", 0) == 0 && is_ident(src.substr(12), false)) out.event = src;
  else if (src == "metadata" || src == "item") out.event = "item";
  else if (src == "user" || src == "ranking" || src == "*") out.event = src;
  else bad("cannot decode source field " + src);
  return out;
}

void parse_scope(const std::string &s, ScopeId &scope, std::string &field) {
  field.clear();
  if (s == "global") scope = SC_GLOBAL;
  else if (s == "item") scope = SC_ITEM;
  else if (s == "user") scope = SC_USER;
  else if (s == "session") scope = SC_SESSION;
  else if (s == "ranking") scope = SC_RANKING;
  else if (s.rfind("item.", 0) == 0 && is_ident(s.substr(5), true)) { scope = SC_FIELD; field = s.substr(5); }
  else if (s.rfind("ranking.", 0) == 0 && is_ident(s.substr(8), true)) { scope = SC_IRF; field = s.substr(8); }
  else bad("scope type " + s + " not supported");
}

const json::Value &need(const json::Value &o, const char *key, const std::string &fname) {
  const json::Value *v = o.find(key);
  if (!v || v->is_null()) bad("feature '" + fname + "': missing '" + key + "'");
  return *v;
}

std::string source_field(const json::Value &o, const std::string &fname, const char *k1, const char *k2, bool &is_ranking,
                         std::string *event_out = nullptr) {
  const json::Value *v = o.find(k1);
  if ((!v || v->is_null()) && k2) v = o.find(k2);
  if (!v || v->is_null()) bad("feature '" + fname + "': missing source field");
  FieldName fn = parse_field_name(v->as_string());
  is_ranking = fn.event == "ranking";
  if (event_out) *event_out = fn.event;
  return fn.field;
}

int vector_dim(const json::Value *reduce) {
  if (!reduce || reduce->is_null()) return 4;  // [min, max, size, avg] (NumVectorFeature.scala:28)
  int d = 0;
  for (auto &r : reduce->arr) {
    const std::string &s = r.as_string();
    static const char *one[] = {"first", "last", "min", "max", "avg", "random", "sum", "size", "euclidean_distance"};
    bool ok = false;
    for (auto o : one) ok = ok || s == o;
    if (ok) { d += 1; continue; }
    if (s.rfind("vector", 0) == 0 && s.size() > 6 && s.size() <= 6 + 6 && std::all_of(s.begin() + 6, s.end(), ::isdigit)) { d += atoi(s.c_str() + 6); continue; }
    bad("reducer " + s + " is not supported");
  }
  return d;
}

// util/DurationJson.scala:9-13: ([0-9]+)([smhd])
static bool parse_duration_ms(const std::string &s, int64_t &out) {
  if (s.size() < 2) return false;
  int64_t n = 0;
  for (size_t i = 0; i + 1 < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    n = n * 10 + (s[i] - '0');
    if (n > 100000000000LL) return false;  // (FiniteDuration is bounded too: ~292 years; 1e11 days is far outside)
  }
  switch (s.back()) {
    case 's': out = n * 1000; return true;
    case 'm': out = n * 60 * 1000; return true;
    case 'h': out = n * 3600 * 1000; return true;
    case 'd': out = n * 86400 * 1000; return true;
    default: return false;
  }
}

// window_count / rate: `bucket` and `periods` (PeriodicCounterConfig(period = bucket, sumPeriodRanges = periods.map(PeriodRange(_, 0))),
// feature/WindowInteractionCountFeature.scala:25-32, feature/RateFeature.scala:50-90)
static void parse_window(const json::Value &o, FeatureDef &f) {
  if (const json::Value *b = o.find("bucket"))
    if (!b->is_null() && !parse_duration_ms(b->as_string(), f.bucket_ms))
      throw StatusError(MRK_ERR_PARSE, "duration is in wrong format: " + b->as_string());
  if (const json::Value *p = o.find("periods"))
    if (!p->is_null())
      for (auto &v : p->arr) f.periods.push_back((int32_t)v.as_int());
}

// Iteration order of a scala.collection.immutable.Map[String, _] of more than 4 entries (Scala 2.13, build.sbt:6) - what
// `for ((fieldName, feature) <- fields)` in InteractedWithFeature.scala:56-65,152-162 walks, i.e. the order of the feature's
// columns.  Up to 4 entries a Map keeps insertion order (Map1..Map4); beyond, `toMap` builds a HashMap: a compressed
// hash-array-mapped prefix tree over improve(key.hashCode), 5 bits per level from the LOW end, whose iterator yields a
// node's own entries by ascending 5-bit index first and then its sub-nodes by ascending index, depth first
// (immutable/HashMap.scala, ChampCommon.scala `ChampBaseIterator`; hashing: collection/Hashing.scala `improve`).  The
// structure is canonical - it does not depend on insertion order - except for keys whose 32-bit hashes are EQUAL, which
// share a collision node in insertion order.  Known answer: Map(a,b,c,d,e) iterates e, a, b, c, d.
static uint32_t java_string_hash(const std::string &utf8) {   // String.hashCode over UTF-16 code units
  uint32_t h = 0;
  for (size_t i = 0; i < utf8.size();) {
    const unsigned char c = (unsigned char)utf8[i];
    uint32_t cp;
    int n;
    if (c < 0x80) { cp = c; n = 1; }
    else if ((c >> 5) == 6) { cp = c & 31; n = 2; }
    else if ((c >> 4) == 14) { cp = c & 15; n = 3; }
    else { cp = c & 7; n = 4; }
    for (int k = 1; k < n && i + k < utf8.size(); ++k) cp = (cp << 6) | ((unsigned char)utf8[i + k] & 63);
    i += n;
    if (cp >= 0x10000) {
      cp -= 0x10000;
      h = 31 * h + (0xD800 + (cp >> 10));
      h = 31 * h + (0xDC00 + (cp & 0x3ff));
    } else {
      h = 31 * h + cp;
    }
  }
  return h;
}

static uint32_t scala_improve(uint32_t hcode) {
  uint32_t h = hcode + ~(hcode << 9);
  h ^= h >> 14;
  h += h << 4;
  return h ^ (h >> 10);
}

static void champ_order(const std::vector<std::pair<uint32_t, int>> &node, int shift, std::vector<int> &out) {
  if (shift >= 32) {   // equal hashes: a collision node, insertion order
    for (auto &e : node) out.push_back(e.second);
    return;
  }
  std::vector<std::pair<uint32_t, int>> at[32];
  for (auto &e : node) at[(e.first >> shift) & 31].push_back(e);
  for (int m = 0; m < 32; ++m)
    if (at[m].size() == 1) out.push_back(at[m][0].second);
  for (int m = 0; m < 32; ++m)
    if (at[m].size() > 1) champ_order(at[m], shift + 5, out);
}

}  // namespace

std::vector<std::string> scala_map_key_order(const std::vector<std::string> &keys) {
  if (keys.size() <= 4) return keys;
  std::vector<std::pair<uint32_t, int>> root;
  for (size_t i = 0; i < keys.size(); ++i) root.emplace_back(scala_improve(java_string_hash(keys[i])), (int)i);
  std::vector<int> order;
  champ_order(root, 0, order);
  std::vector<std::string> out;
  for (int i : order) out.push_back(keys[(size_t)i]);
  return out;
}

namespace {

std::unique_ptr<FeatureDef> parse_feature(const json::Value &o) {
  std::unique_ptr<FeatureDef> f(new FeatureDef());
  const std::string type = o.at("type").as_string();
  f->name = o.at("name").as_string();
  const std::string &nm = f->name;
  auto scope_required = [&]() { parse_scope(need(o, "scope", nm).as_string(), f->scope, f->scope_field); };
  auto plain_scope = [&]() {
    scope_required();
    if (f->scope == SC_FIELD || f->scope == SC_IRF) {
      // readKey gives None for field scopes (BaseFeature.scala:33-34): such a feature is always missing;
      // keep it working by pointing at a scope that never resolves
    }
  };
  if (type == "number") {
    f->type = FType::Number;
    f->field = source_field(o, nm, "source", "field", f->field_is_ranking);
    plain_scope();
  } else if (type == "boolean") {
    f->type = FType::Boolean;
    f->field = source_field(o, nm, "field", "source", f->field_is_ranking);
    plain_scope();
  } else if (type == "word_count") {
    f->type = FType::WordCount;
    f->field = source_field(o, nm, "source", nullptr, f->field_is_ranking);
    plain_scope();
  } else if (type == "vector") {
    f->type = FType::Vector;
    f->field = source_field(o, nm, "source", nullptr, f->field_is_ranking);
    plain_scope();
    f->dim = vector_dim(o.find("reduce"));
  } else if (type == "string") {
    f->type = FType::String;
    f->field = source_field(o, nm, "source", "field", f->field_is_ranking);
    plain_scope();
    for (auto &v : need(o, "values", nm).arr) f->values.push_back(v.as_string());
    if (f->values.empty()) bad("feature '" + nm + "': values must not be empty");
    const json::Value *enc = o.find("encode");
    if (enc && !enc->is_null()) {
      if (enc->as_string() == "index") f->index_encode = true;
      else if (enc->as_string() != "onehot") bad("string encoding method " + enc->as_string() + " is not supported");
    }
    f->dim = f->index_encode ? 1 : (int)f->values.size();
  } else if (type == "interaction_count") {
    f->type = FType::InteractionCount;
    plain_scope();
  } else if (type == "window_count") {
    f->type = FType::WindowCount;
    plain_scope();
    f->dim = (int)need(o, "periods", nm).arr.size();
    parse_window(o, *f);
  } else if (type == "rate") {
    f->type = FType::Rate;
    const json::Value *sc = o.find("scope");
    if (sc && !sc->is_null()) {
      parse_scope(sc->as_string(), f->scope, f->scope_field);
      if (f->scope != SC_ITEM && f->scope != SC_FIELD && f->scope != SC_IRF)
        bad("scope " + sc->as_string() + " is not supported for rate feature " + nm);
    }
    f->top = need(o, "top", nm).as_string();
    f->bottom = need(o, "bottom", nm).as_string();
    f->dim = (int)need(o, "periods", nm).arr.size();
    parse_window(o, *f);
    const json::Value *norm = o.find("normalize");
    if (norm && !norm->is_null()) {
      f->normalize = true;
      f->weight = norm->at("weight").as_double();
    }
  } else if (type == "interacted_with") {
    f->type = FType::InteractedWith;
    if (const json::Value *c = o.find("count")) if (!c->is_null()) f->list_count = c->as_int();  // InteractedWithFeature.scala:47-55
    if (const json::Value *d = o.find("duration")) if (!d->is_null() && !parse_duration_ms(d->as_string(), f->list_duration_ms))
      bad("duration is in wrong format: " + d->as_string());
    scope_required();
    if (f->scope != SC_SESSION && f->scope != SC_USER) bad("feature '" + nm + "': can only be scoped to user/session");
    const json::Value &fl = need(o, "field", nm);
    std::vector<std::string> fields;
    if (fl.is_string()) fields.push_back(fl.as_string());
    else for (auto &v : fl.arr) fields.push_back(v.as_string());
    std::set<std::string> seen;
    for (auto &s : fields) {
      FieldName fn = parse_field_name(s);
      if (fn.event != "item") bad("feature '" + nm + "': can only be applied to item fields");
      if (!seen.insert(fn.field).second) bad("feature '" + nm + "': field '" + fn.field + "' is listed twice (dim mismatch in the reference)");
      f->values.push_back(fn.field);
    }
    // the reference keeps the fields in a Scala immutable Map: insertion order up to 4 entries, hash
    // order beyond (InteractedWithFeature.scala:56-65,152-162): computed here (scala_map_key_order); a host that
    // knows better - another Scala version - passes the order it sees as "field_order"
    if (const json::Value *ord = o.find("field_order")) {
      std::vector<std::string> order;
      for (auto &v : ord->arr) order.push_back(v.as_string());
      std::vector<std::string> a = order, b2 = f->values;
      std::sort(a.begin(), a.end());
      std::sort(b2.begin(), b2.end());
      if (a != b2) bad("feature '" + nm + "': field_order must be a permutation of the fields");
      f->values = order;
    } else {
      f->values = scala_map_key_order(f->values);
    }
    f->dim = (int)f->values.size();
  } else if (type == "diversity") {
    f->type = FType::Diversity;
    std::string ev;
    f->field = source_field(o, nm, "source", nullptr, f->field_is_ranking, &ev);
    if (ev != "item") bad("diversity feature '" + nm + "' can only accept item fields, but got '" + ev + "'");
    const json::Value *top = o.find("top");
    f->div_top = (top && !top->is_null()) ? (int)top->as_int() : 20;  // DiversityFeature.scala:164
    if (f->div_top < 0) f->div_top = 0;
  } else if (type == "item_age") {
    f->type = FType::ItemAge;
    std::string ev;
    f->field = source_field(o, nm, "source", nullptr, f->field_is_ranking, &ev);
    if (ev != "item") bad("feature '" + nm + "': can only work with fields from metadata events");
  } else if (type == "local_time") {
    f->type = FType::LocalTime;
    f->field = source_field(o, nm, "source", nullptr, f->field_is_ranking);
    if (!f->field_is_ranking) bad("feature '" + nm + "': can only work with ranking event fields");
    const std::string p = need(o, "parse", nm).as_string();
    static const char *names[] = {"time_of_day", "day_of_week", "month_of_year", "year", "second"};
    f->mapper = -1;
    for (int i = 0; i < 5; ++i) if (p == names[i]) f->mapper = i;
    if (f->mapper < 0) bad("parsing method " + p + " is not supported");
  } else if (type == "position") {
    f->type = FType::Position;
    f->position = (double)need(o, "position", nm).as_int();
  } else if (type == "relevancy") {
    f->type = FType::Relevancy;
  } else if (type == "field_match" && o.find("method") && o.at("method").find("type") &&
             o.at("method").at("type").as_string() == "bi-encoder") {
    f->type = FType::Biencoder;
    f->ext_field = "__embedding:" + nm;
    if (const json::Value *rf = o.find("rankingField"))  // "ranking.<field>": the query text (FieldMatchBiencoderFeature.scala:89-92)
      if (rf->is_string()) {
        const std::string &s = rf->as_string();
        f->field = s.compare(0, 8, "ranking.") == 0 ? s.substr(8) : s;
      }
    const json::Value *d = o.at("method").find("dim");
    if (!d || d->is_null()) bad("feature '" + nm + "': method.dim (embedding size) is required");
    f->qdim = (int)d->as_int();
    if (const json::Value *n = o.find("norm")) {
      if (!n->is_null()) {
        const std::string &s = n->as_string();
        if (s == "noop") f->norm = NORM_NOOP;
        else if (s == "linear") f->norm = NORM_MINMAX;
        else if (s == "position") f->norm = NORM_POSITION;
        else bad("normalizer " + s + " is not supported");
      }
    }
    if (const json::Value *dist = o.find("distance"))
      if (!dist->is_null()) {
        const std::string &s = dist->as_string();
        if (!(s == "cos" || s == "Cos" || s == "cosine" || s == "Cosine")) throw StatusError(MRK_ERR_UNSUPPORTED, "distance '" + s + "' is not supported");
      }
  } else if (type == "ua" || type == "referer") {
    // request-level one-hot columns that need JVM-only parsers (uap-java / referer lists): the
    // host computes them and sends them as NumberListField "__ext:<name>"
    f->type = FType::ExternalRanking;
    f->ext_field = "__ext:" + nm;
    f->dim = (int)need(o, "dim", nm).as_int();
  } else if (type == "field_match" && o.find("method") && o.at("method").find("type") &&
             o.at("method").at("type").as_string() == "cross-encoder") {
    // FieldMatchCrossEncoderFeature: logits of (rankingField text, stored item text) pairs.  Without a bound encoder
    // it behaves like the other host-computed columns: values arrive as item field "__ext:<name>" (ScoreCache hits).
    f->type = FType::ExternalItem;
    f->cross = true;
    f->ext_field = "__ext:" + nm;
    f->dim = 1;
    if (const json::Value *rf = o.find("rankingField"))
      if (rf->is_string()) {
        const std::string &s = rf->as_string();
        f->field = s.compare(0, 8, "ranking.") == 0 ? s.substr(8) : s;
      }
    if (const json::Value *n = o.find("norm"))
      if (!n->is_null()) {
        const std::string &s = n->as_string();
        if (s == "noop") f->norm = NORM_NOOP;
        else if (s == "linear") f->norm = NORM_MINMAX;
        else if (s == "position") f->norm = NORM_POSITION;
        else bad("normalizer " + s + " is not supported");
      }
  } else if (type == "field_match" || type == "random") {
    // Lucene analyzers / ONNX cross-encoder / RNG stay on the JVM: per-item values arrive as
    // item field "__ext:<name>"
    f->type = FType::ExternalItem;
    f->ext_field = "__ext:" + nm;
    const json::Value *d = o.find("dim");
    f->dim = (d && !d->is_null()) ? (int)d->as_int() : 1;
  } else {
    bad("feature type " + type + " is not supported");
  }
  if (f->dim < 1) bad("feature '" + nm + "': dimension must be positive");
  return f;
}

ColRef col_ref(const Store &st, ScopeId scope, const std::string &name) {
  const Table &t = st.tables[scope];
  auto it = t.col_of.find(name);
  if (it == t.col_of.end()) return ColRef{-1, 0};
  return ColRef{t.cols[it->second].tag_index, t.cols[it->second].val_off};
}

ScopeId plain_table(const FeatureDef &f) {
  // features read through BaseFeature.readKey: field scopes resolve to None => never present
  return (f.scope == SC_FIELD || f.scope == SC_IRF) ? SC_COUNT : f.scope;
}

void declare_columns(const FeatureDef &f, Store &st) {
  const ScopeId t = plain_table(f);
  switch (f.type) {
    case FType::Number: case FType::WordCount:
      if (f.scope != SC_RANKING || f.type == FType::Number) { if (t != SC_COUNT) st.add_column(t, f.name, COL_SCALAR, 0); }
      break;
    case FType::Boolean: case FType::Vector:
      if (t != SC_COUNT) st.add_column(t, f.name, COL_SCALAR, 0);
      break;
    case FType::String:
      if (t != SC_COUNT) st.add_column(t, f.name, COL_SCALAR, 0, "", /*expect_list=*/true);
      break;
    case FType::InteractionCount:
      if (t != SC_COUNT) st.add_column(t, f.name, COL_COUNTER, 0);
      break;
    case FType::WindowCount:
      if (t != SC_COUNT) {
        st.add_column(t, f.name, COL_PERIODIC, f.dim);
        st.set_periodic_config(t, f.name, f.bucket_ms, f.periods);
      }
      break;
    case FType::Rate: {
      const std::string top = f.name + "_" + f.top, bot = f.name + "_" + f.bottom;
      ScopeId target = f.scope == SC_ITEM ? SC_ITEM : (f.scope == SC_FIELD ? SC_FIELD : SC_IRF);
      st.add_column(target, top, COL_PERIODIC, f.dim);
      st.add_column(target, bot, COL_PERIODIC, f.dim);
      st.add_column(SC_GLOBAL, top + "_norm", COL_PERIODIC, f.dim);
      st.add_column(SC_GLOBAL, bot + "_norm", COL_PERIODIC, f.dim);
      st.set_periodic_config(target, top, f.bucket_ms, f.periods);
      st.set_periodic_config(target, bot, f.bucket_ms, f.periods);
      st.set_periodic_config(SC_GLOBAL, top + "_norm", f.bucket_ms, f.periods);
      st.set_periodic_config(SC_GLOBAL, bot + "_norm", f.bucket_ms, f.periods);
      if (f.scope == SC_FIELD) st.add_column(SC_ITEM, f.name + "_field", COL_SCALAR, 0, f.scope_field);
      break;
    }
    case FType::InteractedWith:
      st.add_column(f.scope, f.name + "_interactions", COL_BOUNDED_LIST, 0);
      st.set_list_config(f.scope, f.name + "_interactions", f.list_count, f.list_duration_ms);
      for (auto &fld : f.values) st.add_column(SC_ITEM, f.name + "_" + fld, COL_SCALAR, 0, "", /*expect_list=*/true);
      break;
    case FType::Diversity:  // string lists or numbers: the heap is sized as if every one held a list
      st.add_column(SC_ITEM, f.name, COL_SCALAR, 0, "", /*expect_list=*/true);
      break;
    case FType::ItemAge: case FType::Biencoder:
      st.add_column(SC_ITEM, f.name, COL_SCALAR, 0);
      break;
    default: break;
  }
}

void build_program(Program &p, const std::vector<const FeatureDef *> &feats, const Store &st) {
  int dst = 0;
  p.item_fixed = (int)st.tables[SC_ITEM].heap_off;
  for (const FeatureDef *f : feats) {
    Op op{};
    HostOp ho;
    ho.def = f;
    ho.dst = dst;
    op.dst = dst;
    op.dim = f->dim;
    op.scope = f->scope;
    op.c0 = op.c1 = op.c2 = op.c3 = op.c4 = op.c5 = ColRef{-1, 0};
    auto as_const = [&](int n) {
      op.kind = OP_CONST;
      ho.const_idx = p.n_consts;
      op.i0 = p.n_consts;
      p.n_consts += n;
    };
    const ScopeId t = plain_table(*f);
    auto scoped_col = [&]() {
      if (t == SC_COUNT) { op.scope = SC_ITEM; op.c0 = ColRef{-1, 0}; }  // never resolves
      else { op.scope = t; op.c0 = col_ref(st, t, f->name); }
    };
    switch (f->type) {
      case FType::Number:
        if (f->scope == SC_RANKING) as_const(1);
        else { op.kind = OP_SCALAR_DOUBLE; scoped_col(); }
        break;
      case FType::WordCount:
        if (f->scope == SC_RANKING) as_const(1);
        else { op.kind = OP_SCALAR_DOUBLE; scoped_col(); }
        break;
      case FType::Boolean: op.kind = OP_SCALAR_BOOL; scoped_col(); break;
      case FType::Vector: op.kind = OP_VECTOR; scoped_col(); break;
      case FType::String:
        if (f->field_is_ranking) { as_const(f->dim); break; }
        op.kind = f->index_encode ? OP_STRING_INDEX : OP_STRING_ONEHOT;
        scoped_col();
        op.i0 = (int)p.aux.size();
        op.i1 = (int)f->values.size();
        for (auto &v : f->values) p.aux.push_back(const_cast<Store &>(st).intern(v));
        break;
      case FType::InteractionCount: op.kind = OP_COUNTER; scoped_col(); break;
      case FType::WindowCount: op.kind = OP_WINDOW; scoped_col(); break;
      case FType::Rate: {
        op.kind = OP_RATE;
        const std::string top = f->name + "_" + f->top, bot = f->name + "_" + f->bottom;
        op.c2 = col_ref(st, SC_GLOBAL, top + "_norm");
        op.c3 = col_ref(st, SC_GLOBAL, bot + "_norm");
        op.i3 = f->normalize ? 1 : 0;
        op.d0 = f->weight;
        if (f->scope == SC_ITEM) {
          op.i0 = RATE_ITEM;
          op.c0 = col_ref(st, SC_ITEM, top);
          op.c1 = col_ref(st, SC_ITEM, bot);
        } else if (f->scope == SC_FIELD) {
          op.i0 = RATE_ITEM_FIELD;
          op.c0 = col_ref(st, SC_ITEM, f->name + "_field");
          op.c4 = col_ref(st, SC_FIELD, top);
          op.c5 = col_ref(st, SC_FIELD, bot);
        } else {
          op.i0 = RATE_RANKING_FIELD;
          op.c4 = col_ref(st, SC_IRF, top);
          op.c5 = col_ref(st, SC_IRF, bot);
          ho.irf_id = p.n_irf;
          op.i2 = p.n_irf++;
        }
        break;
      }
      case FType::InteractedWith: {
        op.kind = OP_INTERACTED;
        op.i0 = (int)p.aux.size();
        op.i1 = (int)p.prep.size();
        ho.prep_base = op.i1;
        for (auto &fld : f->values) {
          ColRef c = col_ref(st, SC_ITEM, f->name + "_" + fld);
          p.aux.push_back((uint32_t)c.tag);
          p.aux.push_back((uint32_t)c.val);
          PrepEntry pe{};
          pe.kind = PREP_IW_FIELD;
          pe.item_col = c;
          pe.list_scope = f->scope;
          pe.list_col = col_ref(st, f->scope, f->name + "_interactions");
          p.prep.push_back(pe);
        }
        break;
      }
      case FType::Diversity: {
        op.kind = OP_DIVERSITY;
        op.scope = SC_ITEM;
        op.c0 = col_ref(st, SC_ITEM, f->name);
        op.i1 = (int)p.prep.size();
        ho.prep_base = op.i1;
        PrepEntry pe{};
        pe.kind = PREP_DIVERSITY;
        pe.item_col = op.c0;
        pe.top = f->div_top;
        p.prep.push_back(pe);
        break;
      }
      case FType::ItemAge:
        op.kind = OP_ITEM_AGE;
        op.scope = SC_ITEM;
        op.c0 = col_ref(st, SC_ITEM, f->name);
        break;
      case FType::LocalTime: case FType::Position: as_const(1); break;
      case FType::ExternalRanking: as_const(f->dim); break;
      case FType::Relevancy: case FType::ExternalItem:
        op.kind = OP_FILL_NAN;
        if (f->cross && f->norm != NORM_NOOP) p.norm_cols.push_back({dst, f->norm, f});  // FieldMatchCrossEncoderFeature.scala:111
        break;
      case FType::Biencoder:
        if (f->norm != NORM_NOOP) p.norm_cols.push_back({dst, f->norm, nullptr});  // schema.norm.scale(raw), FieldMatchBiencoderFeature.scala:107
        op.kind = OP_BIENCODER;
        op.scope = SC_ITEM;
        op.c0 = col_ref(st, SC_ITEM, f->name);
        ho.const_idx = p.n_consts;
        op.i0 = p.n_consts;
        p.n_consts += 1 + f->qdim;
        break;
    }
    dst += f->dim;
    p.ops.push_back(op);
    p.host_ops.push_back(ho);
  }
  p.dim = dst;
}

void upload(Program &p) {
  if (p.prep.size() > 32)  // FUSED_MAX_PREP (rank.hip): per-request reductions the pre-pass keeps scratch for
    throw StatusError(MRK_ERR_UNSUPPORTED, "model '" + p.model + "' needs " + std::to_string(p.prep.size()) +
                                               " per-request reductions (interacted_with fields + diversity features); 32 are supported");
  auto up = [](DevBuf &b, const void *src, size_t bytes) {
    b.reserve(bytes ? bytes : 16);
    if (bytes) MRK_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
  };
  up(p.d_ops, p.ops.data(), p.ops.size() * sizeof(Op));
  up(p.d_prep, p.prep.data(), p.prep.size() * sizeof(PrepEntry));
  up(p.d_aux, p.aux.data(), p.aux.size() * sizeof(uint32_t));
}

}  // namespace

Program::~Program() { jit_release(*this); }

ProgramDev Program::device_view() const {
  ProgramDev d{};
  d.ops = (const Op *)d_ops.p;
  d.n_ops = (int)ops.size();
  d.prep = (const PrepEntry *)d_prep.p;
  d.n_prep = (int)prep.size();
  d.aux = (const uint32_t *)d_aux.p;
  d.dim = dim;
  d.n_consts = n_consts;
  return d;
}

Registry::~Registry() {
  for (auto &f : features)
    if (f->encoder) encoder_release(f->encoder);
}

void unbind_encoders(mrk_ctx *ctx) {
  std::vector<mrk_encoder *> drop;
  {
    StoreWriteLock lk(ctx);
    if (ctx->registry)
      for (auto &f : ctx->registry->features)
        if (f->encoder) { drop.push_back(f->encoder); f->encoder = nullptr; }
  }
  for (mrk_encoder *e : drop) encoder_release(e);
}

void bind_encoder(mrk_ctx *ctx, const char *feature, mrk_encoder *enc) {
  StoreWriteLock lk(ctx);  // resolve_requests reads FeatureDef::encoder under the shared lock
  if (!ctx->registry) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_config_bind_encoder: load a config first");
  for (auto &f : ctx->registry->features)
    if (f->name == feature) {
      if (f->type != FType::Biencoder && !f->cross)
        throw StatusError(MRK_ERR_UNSUPPORTED, std::string("feature ") + feature + " is not a bi-encoder / cross-encoder field_match");
      if (f->cross && !enc->dev.shape.classifier)
        throw StatusError(MRK_ERR_UNSUPPORTED, std::string("feature ") + feature + " needs a cross-encoder (pooler + classifier head)");
      encoder_retain(enc);
      if (f->encoder) encoder_release(f->encoder);
      f->encoder = enc;
      return;
    }
  throw StatusError(MRK_ERR_NOT_FOUND, std::string("feature ") + feature + " is not configured");
}

const Program *Registry::program(const std::string &model) const {
  auto it = programs.find(model);
  return it == programs.end() ? nullptr : it->second.get();
}

std::unique_ptr<Registry> load_config(const char *json_text, size_t len, Store &store, bool do_upload) {
  json::Value root = json::parse(json_text, len);
  std::unique_ptr<Registry> reg(new Registry());
  const json::Value *feats = root.find("features");
  if (!feats || !feats->is_array()) bad("config: 'features' must be a list");
  std::set<std::string> names;
  for (auto &fo : feats->arr) {
    auto f = parse_feature(fo);
    if (!names.insert(f->name).second) bad("config: feature '" + f->name + "' is defined twice");
    reg->features.push_back(std::move(f));
  }
  for (auto &f : reg->features) declare_columns(*f, store);
  for (auto &f : reg->features)
    if (f->cross) store.texts[f->name];  // item texts of this column are kept on the host (Store::put_string)
  store.freeze_layout();
  const json::Value *models = root.find("models");
  if (models && models->is_object()) {
    for (auto &kv : models->obj) {
      const json::Value *tp = kv.second.find("type");
      if (!tp || tp->as_string() != "lambdamart") continue;
      std::unique_ptr<Program> p(new Program());
      p->model = kv.first;
      std::vector<const FeatureDef *> ordered;
      for (auto &fn : kv.second.at("features").arr) {
        p->feature_names.push_back(fn.as_string());
        // FeatureMapping.scala:66-71: names without a definition are silently dropped
        for (auto &f : reg->features)
          if (f->name == fn.as_string()) { ordered.push_back(f.get()); break; }
      }
      build_program(*p, ordered, store);
      if (do_upload) upload(*p);
      reg->programs[kv.first] = std::move(p);
    }
  }
  return reg;
}

// =====================================================================================================
// host half of a request

namespace {

const mrk_field *fields_map_get(const mrk_request &r, const std::string &name) {  // fieldsMap: last wins
  const mrk_field *hit = nullptr;
  for (int i = 0; i < r.n_fields; ++i)
    if (r.fields[i].name && name == r.fields[i].name) hit = &r.fields[i];
  return hit;
}
const mrk_field *fields_find(const mrk_request &r, const std::string &name) {  // List.find: first wins
  for (int i = 0; i < r.n_fields; ++i)
    if (r.fields[i].name && name == r.fields[i].name) return &r.fields[i];
  return nullptr;
}

// "\\s+".r.split(s).length (WordCountFeature.scala:73-76), java.util.regex.Pattern.split semantics
int token_count(const char *s) {
  auto ws = [](unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); };
  const size_t n = strlen(s);
  if (n == 0) return 1;
  int pieces = 0, last_nonempty = 0;
  size_t i = 0;
  bool matched = false;
  size_t start = 0;
  while (i < n) {
    if (ws((unsigned char)s[i])) {
      size_t j = i;
      while (j < n && ws((unsigned char)s[j])) ++j;
      matched = true;
      ++pieces;
      if (i > start) last_nonempty = pieces;
      start = j;
      i = j;
    } else {
      ++i;
    }
  }
  if (!matched) return 1;
  ++pieces;
  if (n > start) last_nonempty = pieces;
  return last_nonempty;  // trailing empty pieces are dropped
}

void encode_string(const FeatureDef &f, const char *const *vals, int n, double *out) {
  if (f.index_encode) {
    double idx = 0;
    if (n > 0)
      for (size_t k = 0; k < f.values.size(); ++k)
        if (f.values[k] == vals[0]) idx = (double)(k + 1);
    out[0] = idx;
  } else {
    for (int k = 0; k < f.dim; ++k) out[k] = 0.0;
    for (int j = 0; j < n; ++j)
      for (size_t k = 0; k < f.values.size(); ++k)
        if (f.values[k] == vals[j]) { out[k] = 1.0; break; }
  }
}

// ---- java.time pieces of LocalDateTimeFeature.scala:31-93
int64_t fdiv(int64_t a, int64_t b) { int64_t q = a / b, r = a % b; return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q; }

struct Civil { int64_t year; int month; int dow; int64_t sod; int64_t epoch; };

Civil civil_of(int64_t epoch_second, int64_t offset) {
  const int64_t local = epoch_second + offset;
  const int64_t day = fdiv(local, 86400);
  Civil c;
  c.sod = local - day * 86400;
  c.epoch = epoch_second;
  c.dow = (int)(((day % 7) + 7 + 3) % 7) + 1;  // 1970-01-01 = Thursday
  // Howard Hinnant's days -> civil
  int64_t z = day + 719468;
  const int64_t era = fdiv(z, 146097);
  const uint64_t doe = (uint64_t)(z - era * 146097);
  const uint64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const uint64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const uint64_t mp = (5 * doy + 2) / 153;
  c.month = (int)(mp < 10 ? mp + 3 : mp - 9);
  c.year = (int64_t)yoe + era * 400 + (c.month <= 2 ? 1 : 0);
  return c;
}

bool two_digits(const char *&p, int &out) {
  if (!isdigit((unsigned char)p[0]) || !isdigit((unsigned char)p[1])) return false;
  out = (p[0] - '0') * 10 + (p[1] - '0');
  p += 2;
  return true;
}

// The zone id inside "[...]" of an ISO_DATE_TIME string (ZoneId.of): "Z", a plain offset, "UTC" / "GMT" / "UT" with an optional
// offset - fixed - or a region of the tz database, whose offset AT THE INSTANT comes from the host's zoneinfo (tzif.cpp).
// false = not a zone id java.time knows (the feature's value is then missing, LocalDateTimeFeature.scala:47-49); a host without
// any zoneinfo directory is an error (MRK_ERR_UNSUPPORTED), not a silently missing value.
bool zone_offset_at(const std::string &zone, int64_t instant, int64_t &off) {
  auto fixed = [&](const char *p, bool need_sign_only) -> bool {   // [+-]h[h][:mm[:ss]] | [+-]hhmm[ss]
    if (*p != '+' && *p != '-') return false;
    const int sign = *p == '-' ? -1 : 1;
    ++p;
    int v[3] = {0, 0, 0}, n = 0;
    const size_t len = strlen(p);
    const bool colon = strchr(p, ':') != nullptr;
    if (!colon && (len == 4 || len == 6) && !need_sign_only) {   // +hhmm, +hhmmss
      for (size_t i = 0; i < len; ++i) if (!isdigit((unsigned char)p[i])) return false;
      for (size_t i = 0; i < len / 2; ++i) v[i] = (p[2 * i] - '0') * 10 + (p[2 * i + 1] - '0');
      n = (int)(len / 2);
    } else {
      while (n < 3) {
        if (!isdigit((unsigned char)*p)) return false;
        int x = *p++ - '0';
        if (isdigit((unsigned char)*p)) x = x * 10 + (*p++ - '0');
        v[n++] = x;
        if (*p == ':') { ++p; continue; }
        break;
      }
      if (*p != 0) return false;
    }
    if (v[0] > 18 || v[1] > 59 || v[2] > 59) return false;
    off = sign * (v[0] * 3600 + v[1] * 60 + v[2]);
    return off >= -18 * 3600 && off <= 18 * 3600;
  };
  if (zone == "Z") { off = 0; return true; }
  if (zone[0] == '+' || zone[0] == '-') return fixed(zone.c_str(), false);
  for (const char *pre : {"UTC", "GMT", "UT"}) {
    const size_t n = strlen(pre);
    if (zone.compare(0, n, pre) == 0 && (zone.size() == n || zone[n] == '+' || zone[n] == '-')) {
      if (zone.size() == n) { off = 0; return true; }
      return fixed(zone.c_str() + n, false);
    }
  }
  const TzRules *rules = nullptr;
  switch (tz_lookup(zone, &rules)) {
    case TzLookup::Ok: off = rules->offset_at(instant); return true;
    case TzLookup::UnknownRegion: return false;
    case TzLookup::NoTzdata:
      throw StatusError(MRK_ERR_UNSUPPORTED, "a date-time names the time zone '" + zone + "' but this host has no zoneinfo directory (set MRK_TZDIR or TZDIR; "
                                             "java.time would use the JVM's own tzdb)");
  }
  return false;
}

// ZonedDateTime.parse(_, ISO_DATE_TIME): ISO_LOCAL_DATE_TIME + offset [+ '[' zone ']'].  java.time resolves the INSTANT from the
// written offset (Parsed.resolveInstant) and the local date-time from the zone's rules at that instant (ZonedDateTime.from ->
// create(epochSecond, nano, zone)); without brackets the zone IS the offset.
bool parse_iso_datetime(const char *s, Civil &out) {
  const char *p = s;
  bool neg = false;
  if (*p == '-' || *p == '+') { neg = *p == '-'; ++p; }
  int64_t year = 0;
  int nd = 0;
  while (isdigit((unsigned char)*p)) { year = year * 10 + (*p - '0'); ++p; ++nd; }
  if (nd < 4) return false;
  if (neg) year = -year;
  int mo, d, h, mi, sec = 0;
  if (*p++ != '-' || !two_digits(p, mo) || *p++ != '-' || !two_digits(p, d)) return false;
  if (*p++ != 'T' || !two_digits(p, h) || *p++ != ':' || !two_digits(p, mi)) return false;
  if (*p == ':') { ++p; if (!two_digits(p, sec)) return false; }
  if (*p == '.') { ++p; if (!isdigit((unsigned char)*p)) return false; while (isdigit((unsigned char)*p)) ++p; }
  int64_t off = 0;
  if (*p == 'Z') { ++p; }
  else if (*p == '+' || *p == '-') {
    const int sign = *p == '-' ? -1 : 1;
    ++p;
    int oh, om = 0, os = 0;
    if (!two_digits(p, oh)) return false;
    if (*p == ':') { ++p; if (!two_digits(p, om)) return false; if (*p == ':') { ++p; if (!two_digits(p, os)) return false; } }
    off = sign * (oh * 3600 + om * 60 + os);
  } else return false;
  std::string zone;
  if (*p == '[') {
    const char *e = strchr(p, ']');
    if (!e || e[1] != 0 || e == p + 1) return false;
    zone.assign(p + 1, e);
    p = e + 1;
  }
  if (*p != 0) return false;
  if (mo < 1 || mo > 12 || d < 1 || d > 31 || h > 23 || mi > 59 || sec > 59) return false;
  // civil -> days
  int64_t y = year - (mo <= 2);
  const int64_t era = fdiv(y, 400);
  const uint64_t yoe = (uint64_t)(y - era * 400);
  const uint64_t doy = (153 * (uint64_t)(mo > 2 ? mo - 3 : mo + 9) + 2) / 5 + (uint64_t)d - 1;
  const uint64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  const int64_t days = era * 146097 + (int64_t)doe - 719468;
  const int64_t local = days * 86400 + h * 3600 + mi * 60 + sec;
  int64_t zoff = off;
  if (!zone.empty() && !zone_offset_at(zone, local - off, zoff)) return false;
  out = civil_of(local - off, zoff);
  return true;
}

double map_datetime(int mapper, const Civil &c) {
  switch (mapper) {
    case 0: return (double)c.sod / 3600.0;
    case 1: return (double)c.dow;
    case 2: return (double)c.month;
    case 3: return (double)c.year;
    default: return (double)c.epoch;
  }
}

// open-addressing capacity for a table that receives `tokens` insertions (distinct keys <= tokens): load factor
// <= 0.75 in the worst case, usually far lower (tokens repeat).  The tables of a request live in LDS and their
// size sets how many requests a CU works on at once; measured on the Ranklens workload (MRK_TABLE_LOAD_PCT
// sweep): 75 % -> 0.42 ms per 3840 requests, 50 % -> 0.45, 33 % -> 0.52, 25 % -> 0.60: shorter probe chains do
// not pay for the lost occupancy.  Even => the 8-byte entries of consecutive tables stay 16-byte aligned.
static uint64_t table_load_pct() { return (uint64_t)switches().table_load_pct; }  // MRK_TABLE_LOAD_PCT (experiments)
static uint32_t table_capacity(uint64_t tokens, uint64_t pct) {
  // tokens * 100 / pct without the division (nine of them per request are a third of the host's share of a small
  // request): a 16-bit fixed-point reciprocal rounded UP, so the capacity never falls below the exact quotient
  const uint64_t inv = (100ull * 65536ull + pct - 1) / pct;  // pct is a per-process constant: hoisted by the compiler's inliner
  const uint64_t cap = (tokens < (1ull << 40) ? (tokens * inv) >> 16 : tokens * 100 / pct) + 2;
  return (uint32_t)std::max<uint64_t>((cap + 1) & ~1ull, 8);  // >= one probe window (rank_device.hpp PROBE_W): a window never laps its table
}

// Host threads of one resolve_requests call: MRK_HOST_THREADS, else min(8, hardware threads).  The per-request work
// (id -> slot lookups, request constants, table sizing) only READS the store, so requests are independent.
static int host_threads() {
  if (switches().host_threads > 0) return switches().host_threads;
  const unsigned hw = std::thread::hardware_concurrency();
  return (int)std::max(1u, std::min(8u, hw ? hw : 1u));
}

// fn(lo, hi, worker) over [0, n) in contiguous ranges, worker w taking the w-th range; the first exception in range
// order is rethrown (what a serial loop would have thrown first)
template <typename Fn>
static void parallel_ranges(int64_t n, int workers, const Fn &fn) {
  if (workers <= 1 || n <= 1) { fn((int64_t)0, n, 0); return; }
  workers = (int)std::min<int64_t>(workers, n);
  std::vector<std::exception_ptr> errs((size_t)workers);
  std::vector<std::thread> th;
  th.reserve((size_t)workers - 1);
  auto run = [&](int w) {
    const int64_t lo = n * w / workers, hi = n * (w + 1) / workers;
    try { fn(lo, hi, w); } catch (...) { errs[(size_t)w] = std::current_exception(); }
  };
  for (int w = 1; w < workers; ++w) th.emplace_back(run, w);
  run(0);
  for (auto &t : th) t.join();
  for (auto &e : errs)
    if (e) std::rethrow_exception(e);
}


struct HostCell { uint8_t tag; uint64_t bits; };
HostCell host_cell(const Store &st, ScopeId scope, int32_t slot, ColRef c) {
  if (slot < 0 || c.tag < 0) return {TAG_MISSING, 0};
  const uint8_t *rec = st.record(scope, (uint32_t)slot);
  HostCell h;
  h.tag = rec[c.tag];
  memcpy(&h.bits, rec + c.val, 8);
  return h;
}

}  // namespace

// the request's rankingField as one string (StringField as is, StringListField joined by " ")
static bool query_text(const mrk_request &rq, const FeatureDef &f, std::string &out) {
  if (f.field.empty()) return false;
  const mrk_field *fl = fields_map_get(rq, f.field);
  if (!fl) return false;
  if (fl->type == MRK_FIELD_STRING && fl->str) { out = fl->str; return true; }
  if (fl->type == MRK_FIELD_STRING_LIST) {
    out.clear();
    for (int k = 0; k < fl->n; ++k) { if (k) out.push_back(' '); if (fl->strs && fl->strs[k]) out += fl->strs[k]; }
    return true;
  }
  return false;
}

void resolve_requests(const Program &prog, Store &store, const mrk_request *reqs, int n_req, const mrk_item_ids *ids, HostBatch &hb) {
  // `hb` may be a batch's own (grow-only) scratch: everything is re-assigned below, nothing is freed
  hb.item_slot.clear();
  hb.item_req.clear();
  hb.irf.clear();
  hb.overrides.clear();
  hb.arena_entries = hb.max_req_entries = 0;
  hb.max_doubles = hb.max_items = hb.total_items = 0;
  // first batch item of every request, and where an item's id bytes are
  std::vector<int> begins((size_t)n_req + 1, 0);
  for (int r = 0; r < n_req; ++r) {
    if (reqs[r].n_items < 0 || (reqs[r].n_items > 0 && !ids && !reqs[r].item_ids)) throw StatusError(MRK_ERR_INVALID_ARG, "bad item list");
    if (reqs[r].n_items > (1 << 27)) throw StatusError(MRK_ERR_UNSUPPORTED, "requests with more than 2^27 items are not supported");
    if ((long long)begins[r] + reqs[r].n_items > INT32_MAX) throw StatusError(MRK_ERR_UNSUPPORTED, "batches of more than 2^31 items are not supported");
    begins[r + 1] = begins[r] + reqs[r].n_items;
  }
  if (ids && begins[n_req] > 0 && (!ids->bytes || !ids->offsets)) throw StatusError(MRK_ERR_INVALID_ARG, "null id bytes / offsets");
  auto id_of = [&](int r, int i) -> std::string_view {
    if (ids) {
      const uint32_t o0 = ids->offsets[begins[r] + i], o1 = ids->offsets[begins[r] + i + 1];
      if (o1 < o0 || o1 > ids->bytes_len) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_item_ids: offsets descend or pass bytes_len");
      return std::string_view((const char *)ids->bytes + o0, o1 - o0);
    }
    const char *s = reqs[r].item_ids[i];
    return s ? std::string_view(s) : std::string_view("");
  };
  // Flat ids: the device resolves the slots and the host sizes the pre-pass tables from bounds.  One exception: a numeric
  // `diversity` over more candidates than the pre-pass sorts needs the exact count of present values, i.e. the slots.
  bool device_ids = ids != nullptr;
  if (device_ids)
    for (const HostOp &ho : prog.host_ops)
      if (ho.def->type == FType::Diversity)
        for (int r = 0; r < n_req && device_ids; ++r)
          if (std::min(ho.def->div_top, reqs[r].n_items) > PREP_MAX_VALUES) device_ids = false;
  hb.device_ids = device_ids;
  // queries of encoder-backed bi-encoder columns: every distinct uncached text of the batch goes through the device
  // encoder in one call, before the per-request loop reads them back
  std::map<std::pair<mrk_encoder *, std::string>, std::vector<float>> queries;
  for (const HostOp &ho : prog.host_ops) {
    const FeatureDef &f = *ho.def;
    if (f.type != FType::Biencoder || !f.encoder) continue;
    std::vector<std::string> texts;
    std::string text;
    for (int r = 0; r < n_req; ++r) {
      const mrk_field *fl = fields_map_get(reqs[r], f.ext_field);
      if (fl && fl->type == MRK_FIELD_NUMBER_LIST) continue;
      if (query_text(reqs[r], f, text) && queries.emplace(std::make_pair(f.encoder, text), std::vector<float>()).second) texts.push_back(text);
    }
    if (!texts.empty()) {
      std::vector<std::vector<float>> out;
      encoder_embed_cached(f.encoder, texts, out);
      for (size_t i = 0; i < texts.size(); ++i) queries[std::make_pair(f.encoder, texts[i])] = std::move(out[i]);
    }
  }
  // cross-encoder columns: every (query, item text) pair of the batch goes through the device encoder; the logits
  // come back as per-item overrides of the (NaN-filled) column, FieldMatchCrossEncoderFeature.scala:80-111
  struct CrossValue { int req, item, dst; double v; };
  std::vector<CrossValue> cross_values;
  for (const HostOp &ho : prog.host_ops) {
    const FeatureDef &f = *ho.def;
    if (!f.cross || !f.encoder) continue;
    auto &texts = store.texts[f.name];
    const Tokenizer &tk = f.encoder->tok;
    std::vector<Encoding> rows;
    std::vector<std::pair<int, int>> where;
    std::string text;
    for (int r = 0; r < n_req; ++r) {
      if (!query_text(reqs[r], f, text)) continue;
      const std::vector<int32_t> q = tk.pieces(text);
      for (int i = 0; i < reqs[r].n_items; ++i) {
        auto it = texts.find(std::string(id_of(r, i)));
        if (it == texts.end()) continue;
        if (reqs[r].item_field_offsets && reqs[r].item_fields) {  // a score the caller already has (ScoreCache hit) wins
          bool given = false;
          for (const mrk_field *p = reqs[r].item_fields + reqs[r].item_field_offsets[i], *pe = reqs[r].item_fields + reqs[r].item_field_offsets[i + 1]; p != pe; ++p)
            if (p->name && f.ext_field == p->name) { given = true; break; }
          if (given) continue;
        }
        Store::ItemText &e = it->second;
        if (!e.tokenized) { e.pieces = tk.pieces(e.text); e.tokenized = true; }
        rows.push_back(tk.assemble(q, &e.pieces));
        where.emplace_back(r, i);
      }
    }
    if (rows.empty()) continue;
    std::vector<float> logits(rows.size());
    encoder_score_rows(f.encoder, rows, logits.data());
    // raw logits (SingleValue(name, score: Float) widens); schema.norm.scale runs on the device over the request's whole
    // column - these, the scores the caller already had, NaN for the rest (FieldMatchCrossEncoderFeature.scala:104-111)
    for (size_t k = 0; k < where.size(); ++k) cross_values.push_back({where[k].first, where[k].second, ho.dst, (double)logits[k]});
  }
  const int total = begins[n_req];
  hb.total_items = total;
  hb.reqs.resize(n_req);
  if (!device_ids) {
    hb.item_slot.resize(total);
    hb.item_req.resize(total);
  }
  hb.consts.assign((size_t)n_req * prog.n_consts, kNaN);
  hb.irf.assign((size_t)prog.n_irf * total, -1);
  hb.prep_out.assign((size_t)n_req * prog.prep.size(), PrepOut{0, 0, 0.0, 0, 0});
  for (int r = 0; r < n_req; ++r) hb.reqs[r].item_begin = begins[r];
  const uint64_t load_pct = table_load_pct();
  // small batches (one request of mrk_rank) are not worth a thread; neither is a batch whose ids the device resolves (~70 ns of
  // host work per request: round 4 tried one worker per 1 024 requests of a 3 840-request batch - the threads cost more than
  // they saved, mrk_batch_load 0.285 -> 0.371 ms, profiles/r04_i_ab.txt)
  const int n_workers = (device_ids ? n_req >= 16384 : total >= 16384) ? host_threads() : 1;
  // the ops that ask something of the REQUEST (constants, table sizes); plain per-item columns do not
  std::vector<const HostOp *> request_ops;
  for (const HostOp &ho : prog.host_ops) {
    const FeatureDef &f = *ho.def;
    const bool per_request = (f.type == FType::Number && f.scope == SC_RANKING) || (f.type == FType::WordCount && f.scope == SC_RANKING) ||
                             (f.type == FType::String && f.field_is_ranking) || f.type == FType::LocalTime || f.type == FType::Position ||
                             f.type == FType::ExternalRanking || f.type == FType::Biencoder || (f.type == FType::Rate && f.scope == SC_IRF) ||
                             f.type == FType::InteractedWith || f.type == FType::Diversity;
    if (per_request) request_ops.push_back(&ho);
  }
  const Table &item_table = store.tables[SC_ITEM];
  auto col_max_len = [&](ColRef c) -> uint64_t { return c.tag >= 0 && c.tag < (int)item_table.cols.size() ? item_table.cols[(size_t)c.tag].max_len : 0u; };
  struct Worker { std::vector<Override> overrides; int max_items = 0, max_doubles = 0; std::vector<double> enc; };
  std::vector<Worker> workers((size_t)std::max(1, n_workers));
  std::vector<uint64_t> req_entries((size_t)n_req, 0);  // hash-table entries of every request (its tables are contiguous)
  // one request: everything but the position of its tables in the arena
  auto resolve_one = [&](int r, Worker &wk, int item_workers) {
    const mrk_request &rq = reqs[r];
    ReqDev &rd = hb.reqs[r];
    const int begin = rd.item_begin;
    std::vector<double> &enc = wk.enc;
    uint64_t arena = 0;  // relative to the request's first table
    rd.n_items = rq.n_items;
    // (user / session / ranking slots: resolved for the whole batch before this loop, scope_slots below)
    wk.max_items = std::max(wk.max_items, rq.n_items);
    rd.ts_ms = rq.timestamp_ms;
    // (one request with very many candidates - C4 - spreads its id lookups over the threads instead)
    if (!device_ids) parallel_ranges(rq.n_items, rq.n_items >= 16384 ? item_workers : 1, [&](int64_t lo, int64_t hi, int) {
      // id -> slot, a block at a time: hash every id and ask for its home entry, then probe (the misses overlap)
      const SlotMap &map = store.tables[SC_ITEM].slot_of;
      constexpr int BLOCK = 64;
      uint64_t hs[BLOCK];
      uint32_t lens[BLOCK];
      for (int64_t i0 = lo; i0 < hi; i0 += BLOCK) {
        const int nb = (int)std::min<int64_t>(BLOCK, hi - i0);
        for (int k = 0; k < nb; ++k) {
          const std::string_view id = id_of(r, (int)(i0 + k));
          lens[k] = (uint32_t)id.size();
          hs[k] = SlotMap::hash(id.data(), lens[k]);
          map.prefetch(hs[k]);
        }
        for (int k = 0; k < nb; ++k) {
          const std::string_view idv = id_of(r, (int)(i0 + k));
          const char *id = idv.data();
          const uint32_t s = map.find_hashed(hs[k], id, lens[k]);
          hb.item_slot[begin + i0 + k] = s == SlotMap::NONE ? -1 : (int32_t)s;
          hb.item_req[begin + i0 + k] = (uint32_t)r;
        }
      }
    });
    double *cs = prog.n_consts ? &hb.consts[(size_t)r * prog.n_consts] : nullptr;
    // the table-sizing walks below read the host mirror of a few dozen item records (the head of the candidate list for
    // diversity, the session's interacted items for interacted_with): ask for their lines now, all at once
    auto prefetch_record = [&](int32_t slot) {
      if (slot < 0) return;
      const Table &t = store.tables[SC_ITEM];
      const uint8_t *rec = t.rows.data() + (size_t)slot * t.stride;
      for (uint32_t o = 0; o < t.stride; o += 64) __builtin_prefetch(rec + o);
    };
    if (!prog.prep.empty() && !device_ids) {
      for (int i = 0; i < std::min(rq.n_items, 48); ++i) prefetch_record(hb.item_slot[begin + i]);
      for (const HostOp &ho : prog.host_ops) {
        if (ho.def->type != FType::InteractedWith) continue;
        const HostCell lc = host_cell(store, ho.def->scope, ho.def->scope == SC_SESSION ? rd.session_slot : rd.user_slot, prog.prep[ho.prep_base].list_col);
        if (lc.tag == TAG_MISSING) continue;
        const uint32_t off = (uint32_t)lc.bits, len = (uint32_t)(lc.bits >> 32);
        for (uint32_t k = 0; k < len; ++k) prefetch_record((int32_t)store.slot_pool.host[off + k]);
      }
    }
    for (const HostOp *hop : request_ops) {
      const HostOp &ho = *hop;
      const FeatureDef &f = *ho.def;
      switch (f.type) {
        case FType::Number:
          if (f.scope == SC_RANKING) {  // NumberFeature.scala:77-82
            const mrk_field *fl = fields_map_get(rq, f.field);
            cs[ho.const_idx] = (fl && fl->type == MRK_FIELD_NUMBER) ? fl->num : kNaN;
          }
          break;
        case FType::WordCount:
          if (f.scope == SC_RANKING) {  // WordCountFeature.scala:58-63
            const mrk_field *fl = fields_map_get(rq, f.field);
            cs[ho.const_idx] = (fl && fl->type == MRK_FIELD_STRING && fl->str) ? (double)token_count(fl->str) : kNaN;
          }
          break;
        case FType::String:
          if (f.field_is_ranking) {  // StringFeature.scala:87-93
            const mrk_field *fl = fields_find(rq, f.field);
            if (fl && fl->type == MRK_FIELD_STRING && fl->str) { const char *one[1] = {fl->str}; encode_string(f, one, 1, cs + ho.const_idx); }
            else if (fl && fl->type == MRK_FIELD_STRING_LIST) encode_string(f, fl->strs, fl->n, cs + ho.const_idx);
            else encode_string(f, nullptr, 0, cs + ho.const_idx);
          }
          break;
        case FType::LocalTime: {
          double v = kNaN;
          if (f.field == "timestamp") {  // LocalDateTimeFeature.scala:36-39
            v = map_datetime(f.mapper, civil_of(fdiv(rq.timestamp_ms, 1000), 0));
          } else {
            const mrk_field *fl = fields_map_get(rq, f.field);
            Civil c;
            if (fl && fl->type == MRK_FIELD_STRING && fl->str && parse_iso_datetime(fl->str, c)) v = map_datetime(f.mapper, c);
          }
          cs[ho.const_idx] = v;
          break;
        }
        case FType::Position: cs[ho.const_idx] = f.position; break;  // PositionFeature.scala:32
        case FType::ExternalRanking: {
          const mrk_field *fl = fields_map_get(rq, f.ext_field);
          if (fl && fl->type == MRK_FIELD_NUMBER && f.dim == 1) cs[ho.const_idx] = fl->num;
          else if (fl && fl->type == MRK_FIELD_NUMBER_LIST) {
            if (fl->n != f.dim) throw StatusError(MRK_ERR_DIM_MISMATCH, "for " + f.name + " dim mismatch: " + std::to_string(f.dim) + " != " + std::to_string(fl->n));
            for (int k = 0; k < f.dim; ++k) cs[ho.const_idx + k] = fl->nums[k];
          }
          break;
        }
        case FType::Biencoder: {
          const mrk_field *fl = fields_map_get(rq, f.ext_field);
          cs[ho.const_idx] = -1.0;
          if (fl && fl->type == MRK_FIELD_NUMBER_LIST) {  // host-side cache hit / host-computed embedding
            if (fl->n > f.qdim) throw StatusError(MRK_ERR_DIM_MISMATCH, "query embedding of " + f.name + " is longer than method.dim");
            cs[ho.const_idx] = (double)fl->n;
            for (int k = 0; k < fl->n; ++k) cs[ho.const_idx + 1 + k] = (double)(float)fl->nums[k];
          } else if (f.encoder) {  // rankingCache miss -> encoder.embed(Array(queryString)), FieldMatchBiencoderFeature.scala:93-99
            std::string text;
            if (query_text(rq, f, text)) {
              const std::vector<float> &q = queries.at(std::make_pair(f.encoder, text));
              if ((int)q.size() > f.qdim) throw StatusError(MRK_ERR_DIM_MISMATCH, "encoder of " + f.name + " returns more than method.dim values");
              cs[ho.const_idx] = (double)q.size();
              for (size_t k = 0; k < q.size(); ++k) cs[ho.const_idx + 1 + k] = (double)q[k];
            }
          }
          break;
        }
        case FType::Rate:
          if (f.scope == SC_IRF) {  // RateFeature.scala:302-310
            const mrk_field *fl = fields_map_get(rq, f.scope_field);
            if (fl && fl->type == MRK_FIELD_STRING && fl->str) {
              const std::string prefix = f.scope_field + ":" + fl->str + ":";
              for (int i = 0; i < rq.n_items; ++i) {
                uint32_t s = store.slot(SC_IRF, prefix + std::string(id_of(r, i)), false);
                hb.irf[(size_t)ho.irf_id * total + begin + i] = s == Store::NO_SLOT ? -1 : (int32_t)s;
              }
            }
          }
          break;
        default: break;
      }
      // pre-pass table sizes from the host mirror (counts only — the histograms are built on the device)
      if (f.type == FType::InteractedWith) {
        const ScopeId ls = f.scope;
        const int32_t vslot = ls == SC_SESSION ? rd.session_slot : rd.user_slot;
        for (size_t fi = 0; fi < f.values.size(); ++fi) {
          const PrepEntry &pe = prog.prep[ho.prep_base + fi];
          uint64_t count = 0;
          HostCell lc = host_cell(store, ls, vslot, pe.list_col);
          if (lc.tag != TAG_MISSING) {
            const uint32_t off = (uint32_t)lc.bits, len = (uint32_t)(lc.bits >> 32);
            if (device_ids) {
              count = (uint64_t)len * col_max_len(pe.item_col);  // bound: every interacted item holds the longest list ever put
            } else {
              for (uint32_t k = 0; k < len; ++k) {
                HostCell ic = host_cell(store, SC_ITEM, (int32_t)store.slot_pool.host[off + k], pe.item_col);
                if (ic.tag == TAG_STRING_LIST) count += (uint32_t)(ic.bits >> 32);
              }
            }
          }
          PrepOut &po = hb.prep_out[(size_t)r * prog.prep.size() + ho.prep_base + fi];
          const uint32_t cap = table_capacity(count, load_pct);
          po.tab_off = (uint32_t)arena;
          po.tab_cap = cap;
          arena += cap;
        }
      } else if (f.type == FType::Diversity) {
        const PrepEntry &pe = prog.prep[ho.prep_base];
        uint64_t tokens = 0;
        int taken = 0, doubles = 0;
        int mode = -1;  // -1 undecided, 0 other, 1 string, 2 double
        if (device_ids) {  // bounds: the first `top` candidates all present, each with the longest list the column ever held
          doubles = std::min(f.div_top, rq.n_items);
          tokens = (uint64_t)doubles * col_max_len(pe.item_col);
        }
        for (int i = 0; !device_ids && i < rq.n_items && taken < f.div_top; ++i) {
          HostCell c = host_cell(store, SC_ITEM, hb.item_slot[begin + i], pe.item_col);
          if (c.tag == TAG_MISSING) continue;
          if (mode < 0) mode = (c.tag == TAG_STRING || c.tag == TAG_STRING_LIST) ? 1 : (c.tag == TAG_DOUBLE ? 2 : 0);
          if (mode == 0) break;
          if (mode == 1 && c.tag == TAG_STRING) { tokens += 1; ++taken; }
          else if (mode == 1 && c.tag == TAG_STRING_LIST) { tokens += (uint32_t)(c.bits >> 32); ++taken; }
          else if (mode == 2 && c.tag == TAG_DOUBLE) { ++doubles; ++taken; }
        }
        PrepOut &po = hb.prep_out[(size_t)r * prog.prep.size() + ho.prep_base];
        if (doubles > PREP_MAX_VALUES) {
          // More values than the device pre-pass sorts in LDS (`top` above 4 096 over that many present candidates): the host
          // holds the same values in its mirror (this path runs with host-resolved slots) and takes the median itself -
          // commons-math Percentile(50), LEGACY estimation, NaN removed (DiversityFeature.scala:113-126), the arithmetic of
          // rank_device.hpp median_of - and hands it over as a finished pre-pass result.
          std::vector<double> vals;
          vals.reserve((size_t)doubles);
          int took = 0;
          for (int i = 0; i < rq.n_items && took < f.div_top; ++i) {
            HostCell c = host_cell(store, SC_ITEM, hb.item_slot[begin + i], pe.item_col);
            if (c.tag != TAG_DOUBLE) continue;
            double v;
            memcpy(&v, &c.bits, 8);
            ++took;
            if (v == v) vals.push_back(v);
          }
          std::sort(vals.begin(), vals.end(), [](double a, double b) { return a < b || (a == b && std::signbit(a) && !std::signbit(b)); });  // Arrays.sort: -0.0 before +0.0
          double med = std::numeric_limits<double>::quiet_NaN();
          const size_t m = vals.size();
          if (m == 1) med = vals[0];
          else if (m > 1) {
            const double pos = 0.5 * (double)(m + 1), fpos = std::floor(pos), dif = pos - fpos;
            const size_t ipos = (size_t)fpos;
            if (pos < 1.0) med = vals[0];
            else if (pos >= (double)m) med = vals[m - 1];
            else med = vals[ipos - 1] + dif * (vals[ipos] - vals[ipos - 1]);
          }
          po.mode = DIV_DOUBLE;
          po.scalar = med;
          po.preset = 1;
          doubles = 0;
        }
        wk.max_doubles = std::max(wk.max_doubles, doubles);
        const uint32_t cap = table_capacity(tokens, load_pct);
        po.tab_off = (uint32_t)arena;
        po.tab_cap = cap;
        arena += cap;
      }
    }
    // per-item inputs that win over / replace the store
    if (rq.item_field_offsets && rq.item_fields) {
      for (int i = 0; i < rq.n_items; ++i) {
        const mrk_field *fb = rq.item_fields + rq.item_field_offsets[i], *fe = rq.item_fields + rq.item_field_offsets[i + 1];
        if (fb == fe) continue;
        for (const HostOp &ho : prog.host_ops) {
          const FeatureDef &f = *ho.def;
          const uint32_t gi = (uint32_t)(begin + i);
          if (f.type == FType::Number && f.scope != SC_RANKING) {  // NumberFeature.scala:86-92
            for (const mrk_field *p = fb; p != fe; ++p)
              if (p->type == MRK_FIELD_NUMBER && p->name && f.field == p->name) { wk.overrides.push_back({gi, (uint32_t)ho.dst, p->num}); break; }
          } else if (f.type == FType::String && !f.field_is_ranking) {  // StringFeature.scala:96-99
            for (const mrk_field *p = fb; p != fe; ++p) {
              if (!p->name || f.field != p->name) continue;
              if (p->type != MRK_FIELD_STRING && p->type != MRK_FIELD_STRING_LIST) continue;
              enc.assign(f.dim, 0.0);
              if (p->type == MRK_FIELD_STRING) { const char *one[1] = {p->str ? p->str : ""}; encode_string(f, one, 1, enc.data()); }
              else encode_string(f, p->strs, p->n, enc.data());
              for (int k = 0; k < f.dim; ++k) wk.overrides.push_back({gi, (uint32_t)(ho.dst + k), enc[k]});
              break;
            }
          } else if (f.type == FType::Relevancy) {  // RelevancyFeature.scala:41-48: the first field called "relevancy" decides
            for (const mrk_field *p = fb; p != fe; ++p)
              if (p->name && !strcmp(p->name, "relevancy")) {
                if (p->type == MRK_FIELD_NUMBER) wk.overrides.push_back({gi, (uint32_t)ho.dst, p->num});
                break;
              }
          } else if (f.type == FType::ExternalItem) {
            for (const mrk_field *p = fb; p != fe; ++p)
              if (p->name && f.ext_field == p->name) {
                if (p->type == MRK_FIELD_NUMBER && f.dim == 1) wk.overrides.push_back({gi, (uint32_t)ho.dst, p->num});
                else if (p->type == MRK_FIELD_NUMBER_LIST) {
                  if (p->n != f.dim) throw StatusError(MRK_ERR_DIM_MISMATCH, "for " + f.name + " dim mismatch: " + std::to_string(f.dim) + " != " + std::to_string(p->n));
                  for (int k = 0; k < f.dim; ++k) wk.overrides.push_back({gi, (uint32_t)(ho.dst + k), p->nums[k]});
                }
                break;
              }
          }
        }
      }
    }
    req_entries[(size_t)r] = arena;
  };
  // user / session / ranking ids -> slots, a block of requests at a time: hash every id and ask for its home entry, probe
  // (the cache misses of a block overlap), then ask for the records the per-request pass reads (the bounded list cell
  // of interacted_with lives in the session / user record)
  {
    constexpr int BLOCK = 32;
    uint64_t hs[3][BLOCK];
    uint32_t lens[3][BLOCK];
    const ScopeId scs[3] = {SC_USER, SC_SESSION, SC_RANKING};
    for (int r0 = 0; r0 < n_req; r0 += BLOCK) {
      const int nb = std::min(BLOCK, n_req - r0);
      for (int k = 0; k < nb; ++k) {
        const mrk_request &rq = reqs[r0 + k];
        const char *idp[3] = {rq.user, rq.session, rq.id ? rq.id : ""};
        for (int j = 0; j < 3; ++j) {
          if (store.tables[scs[j]].slot_of.n == 0) idp[j] = nullptr;  // a scope nothing was ever put under: no hashing
          lens[j][k] = idp[j] ? (uint32_t)strlen(idp[j]) : 0u;
          hs[j][k] = idp[j] ? SlotMap::hash(idp[j], lens[j][k]) : 0;
          if (idp[j]) store.tables[scs[j]].slot_of.prefetch(hs[j][k]);
        }
      }
      for (int k = 0; k < nb; ++k) {
        const mrk_request &rq = reqs[r0 + k];
        const char *idp[3] = {rq.user, rq.session, rq.id ? rq.id : ""};
        int32_t out[3];
        for (int j = 0; j < 3; ++j)
          if (store.tables[scs[j]].slot_of.n == 0) idp[j] = nullptr;
        for (int j = 0; j < 3; ++j) {
          const Table &t = store.tables[scs[j]];
          const uint32_t sl = idp[j] ? t.slot_of.find_hashed(hs[j][k], idp[j], lens[j][k]) : SlotMap::NONE;
          out[j] = sl == SlotMap::NONE ? -1 : (int32_t)sl;
          if (out[j] >= 0) __builtin_prefetch(t.rows.data() + (size_t)out[j] * t.stride);
        }
        ReqDev &rd = hb.reqs[r0 + k];
        rd.user_slot = out[0];
        rd.session_slot = out[1];
        rd.ranking_slot = out[2];
      }
    }
  }
  if (n_req == 1) {
    resolve_one(0, workers[0], n_workers);
  } else {
    parallel_ranges(n_req, n_workers, [&](int64_t lo, int64_t hi, int w) {
      for (int64_t r = lo; r < hi; ++r) resolve_one((int)r, workers[(size_t)w], 1);
    });
  }
  // the arena: request after request (serial: a running sum), then every table offset becomes absolute
  uint64_t arena = 0;
  for (int r = 0; r < n_req; ++r) {
    if (arena + req_entries[(size_t)r] > 0xffffffffull) throw StatusError(MRK_ERR_UNSUPPORTED, "batch needs more than 2^32 hash-table entries");
    hb.reqs[r].arena_begin = (uint32_t)arena;
    for (size_t e = 0; e < prog.prep.size(); ++e) hb.prep_out[(size_t)r * prog.prep.size() + e].tab_off += (uint32_t)arena;
    hb.max_req_entries = std::max(hb.max_req_entries, req_entries[(size_t)r]);
    arena += req_entries[(size_t)r];
  }
  for (Worker &wk : workers) {  // workers own increasing request ranges: concatenation keeps request order
    hb.overrides.insert(hb.overrides.end(), wk.overrides.begin(), wk.overrides.end());
    hb.max_items = std::max(hb.max_items, wk.max_items);
    hb.max_doubles = std::max(hb.max_doubles, wk.max_doubles);
  }
  for (const CrossValue &c : cross_values)
    if (c.v == c.v) hb.overrides.push_back({(uint32_t)(hb.reqs[c.req].item_begin + c.item), (uint32_t)c.dst, c.v});
  hb.arena_entries = arena;
}

// tests (not part of include/mrk.h): LocalDateTimeFeature's parse + mapper on the host - 1 = a value, 0 = missing (java.time would
// not parse the string), < 0 = an mrk_status
int debug_local_time(const char *iso, int mapper, double *out) {
  try {
    Civil c;
    if (!iso || !out || !parse_iso_datetime(iso, c)) return 0;
    *out = map_datetime(mapper, c);
    return 1;
  } catch (const StatusError &e) {
    set_last_error(e.what());
    return e.status;
  }
}

}  // namespace mrk

extern "C" {
int mrk_debug_local_time(const char *iso, int mapper, double *out) { return mrk::debug_local_time(iso, mapper, out); }
void mrk_debug_tz_reset(void) { mrk::tz_debug_reset(); }
}
