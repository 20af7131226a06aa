// TEST INFRASTRUCTURE — CPU oracle for the feature-assembly half of the /rank hot path plus the
// final ordering.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library; the product (metarank_amd/) never links, imports or executes it.
//
// It restates, function by function, the reference's per-request read path over a plain
// in-memory Map[Key, FeatureValue] (what MemPersistence holds):
//   M = /root/reference/src/main/scala/ai/metarank
//   M/ml/Ranker.scala:27-83,97-106          rerank / makeQuery / sortBy(-score)
//   M/fstore/FeatureValueLoader.scala:11-25 two-round key fan-out (keys are derived and looked up
//                                           directly in the store here: same result, less work)
//   M/model/ItemValue.scala:25-71           per-feature columns, dim checks
//   M/flow/ClickthroughQuery.scala:32-74    scatter into the dense row-major Array[Double]
//   M/FeatureMapping.scala:89-99            column order = model feature order
//   M/feature/*.scala                       one function per extractor, cited below
// Pinning: every known-answer test the reference holds for this path is
// transcribed in tests/test_oracle_assembly.py and passes against this file.
//
// Style: deliberately the reference's own shape — string keys ("item=42/popularity",
// M/model/Key.scala:9 + M/fstore/codec/impl/ScopeCodec.scala:18-26), hash-map probes, one value
// at a time.  No SoA, no batching: that is the product's job.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

const double NaN = std::numeric_limits<double>::quiet_NaN();

// ---- M/model/Scalar.scala:8-33, M/model/FeatureValue.scala:19-50 ------------------------------
enum ScalarTag { S_STRING = 0, S_DOUBLE = 1, S_BOOLEAN = 2, S_STRING_LIST = 3, S_DOUBLE_LIST = 4 };
struct Scalar {
  int tag = S_DOUBLE;
  double d = 0;
  bool b = false;
  std::string s;
  std::vector<std::string> sl;
  std::vector<double> dl;
};
enum ValueKind { V_SCALAR = 0, V_COUNTER = 1, V_PERIODIC = 2, V_BOUNDED_LIST = 3 };
struct FeatureValue {
  int kind = V_SCALAR;
  Scalar scalar;
  int64_t counter = 0;
  std::vector<int64_t> periodic;  // PeriodicCounterValue.values(i).value
  std::vector<Scalar> list;       // BoundedListValue.values(i).value
};
struct Store {
  std::unordered_map<std::string, FeatureValue> kv;  // MemKVStore: Key.encode -> FeatureValue
  const FeatureValue *get(const std::string &key) const {
    auto it = kv.find(key);
    return it == kv.end() ? nullptr : &it->second;
  }
};

// ---- request (M/model/Event.scala RankingEvent / RankItem, M/model/Field.scala) ---------------
enum FieldType { F_STRING = 0, F_NUMBER = 1, F_BOOL = 2, F_STRING_LIST = 3, F_NUMBER_LIST = 4 };
struct CField {  // same memory layout as mrk_field in include/mrk.h (the tests build one request for both)
  const char *name;
  int32_t type;
  int32_t n;
  double num;
  const char *str;
  const char *const *strs;
  const double *nums;
};
struct CRequest {  // same memory layout as mrk_request
  const char *id;
  int64_t timestamp_ms;
  const char *user;
  const char *session;
  const CField *fields;
  int32_t n_fields;
  int32_t n_items;
  const char *const *item_ids;
  const int32_t *item_field_offsets;
  const CField *item_fields;
};

// event.fieldsMap = fields.map(f => f.name -> f).toMap  => the LAST field of a name wins
const CField *fields_map_get(const CRequest &r, const std::string &name) {
  const CField *hit = nullptr;
  for (int i = 0; i < r.n_fields; ++i)
    if (name == r.fields[i].name) hit = &r.fields[i];
  return hit;
}
// request.fields.find(_.name == field) => the FIRST field of a name wins
const CField *fields_find(const CRequest &r, const std::string &name) {
  for (int i = 0; i < r.n_fields; ++i)
    if (name == r.fields[i].name) return &r.fields[i];
  return nullptr;
}
struct ItemFields {
  const CField *begin = nullptr, *end = nullptr;
};
ItemFields item_fields(const CRequest &r, int i) {
  if (!r.item_field_offsets || !r.item_fields) return {};
  return {r.item_fields + r.item_field_offsets[i], r.item_fields + r.item_field_offsets[i + 1]};
}

// ---- scopes (M/model/ScopeType.scala, M/fstore/codec/impl/ScopeCodec.scala:18-26) -----------
enum ScopeT { SC_GLOBAL = 0, SC_ITEM = 1, SC_USER = 2, SC_SESSION = 3, SC_RANKING = 4, SC_ITEM_FIELD = 5, SC_RANKING_FIELD = 6 };

std::string key_item(const std::string &item, const std::string &feature) { return "item=" + item + "/" + feature; }
std::string key_global(const std::string &feature) { return "global/" + feature; }

// BaseFeature.readKey, M/feature/BaseFeature.scala:28-36.  false => None
bool read_key(const CRequest &r, int scope, const std::string &feature, const std::string &item, std::string &out) {
  switch (scope) {
    case SC_GLOBAL: out = key_global(feature); return true;
    case SC_ITEM: out = key_item(item, feature); return true;
    case SC_USER: if (!r.user) return false; out = std::string("user=") + r.user + "/" + feature; return true;
    case SC_SESSION: if (!r.session) return false; out = std::string("session=") + r.session + "/" + feature; return true;
    case SC_RANKING: out = std::string("ranking=") + (r.id ? r.id : "") + "/" + feature; return true;
    default: return false;  // ItemFieldScopeType / RankingFieldScopeType => None
  }
}

// ---- feature plan -------------------------------------------------------------------------------
enum Kind {
  K_NUMBER = 0, K_BOOLEAN, K_WORD_COUNT, K_VECTOR, K_STRING_INDEX, K_STRING_ONEHOT, K_INTERACTION_COUNT,
  K_WINDOW_COUNT, K_RATE, K_INTERACTED_WITH, K_DIVERSITY, K_ITEM_AGE, K_POSITION, K_RELEVANCY,
  K_CONST /* ranking-level value supplied by the host: local_time, ua, referer, ... */,
  K_ITEM_EXTERNAL /* per-item value supplied by the host in item.fields */, K_BIENCODER, K_LOCAL_TIME
};
enum Mapper { MAP_TIME_OF_DAY = 0, MAP_DAY_OF_WEEK = 1, MAP_MONTH_OF_YEAR = 2, MAP_YEAR = 3, MAP_SECOND = 4 };
enum NormKind { N_NOOP = 0, N_MINMAX = 1, N_POSITION = 2 };

struct Feature {
  int kind = 0;
  std::string name;
  int scope = SC_ITEM;
  std::string scope_field;       // item.<field> / ranking.<field> scopes
  std::string field;             // source field name
  bool field_is_ranking = false; // FieldName(Ranking, _)
  int dim = 1;
  int offset = 0;
  std::vector<std::string> values;  // string: possible values ; interacted_with: field names ; const: unused
  std::string top, bottom;          // rate
  bool normalize = false;
  double weight = 0;
  int div_top = std::numeric_limits<int>::max();
  double position = 0;
  int norm = N_NOOP;
  std::string ext_field;  // K_CONST / K_ITEM_EXTERNAL / K_BIENCODER query embedding: name of the request field
  int mapper = 0;         // K_LOCAL_TIME
};
struct Plan {
  std::vector<Feature> features;  // DatasetDescriptor order
  int dim = 0;
};

struct Thrown {  // exceptions of the reference that surface as HTTP 500
  int code;      // 1 ArithmeticException (/ by zero), 2 IllegalStateException dim mismatch, 3 IllegalArgumentException
};

// ---- extractors -----------------------------------------------------------------------------------

// M/feature/NumberFeature.scala:54-97
void number(const Feature &f, const Store &st, const CRequest &r, double *m, int D) {
  if (f.scope == SC_RANKING) {  // :77-82
    const CField *fld = fields_map_get(r, f.field);
    double v = (fld && fld->type == F_NUMBER) ? fld->num : NaN;
    for (int i = 0; i < r.n_items; ++i) m[(size_t)i * D + f.offset] = v;
    return;
  }
  for (int i = 0; i < r.n_items; ++i) {
    double v = NaN;
    bool overridden = false;
    ItemFields itf = item_fields(r, i);  // :86-92 item.fields.collectFirst { NumberField(name == field) }
    for (const CField *p = itf.begin; p != itf.end; ++p)
      if (p->type == F_NUMBER && f.field == p->name) { v = p->num; overridden = true; break; }
    if (!overridden) {
      std::string key;
      if (read_key(r, f.scope, f.name, r.item_ids[i], key)) {
        const FeatureValue *fv = st.get(key);  // :61-64 ScalarValue(SDouble) else None
        if (fv && fv->kind == V_SCALAR && fv->scalar.tag == S_DOUBLE) v = fv->scalar.d;
      }
    }
    m[(size_t)i * D + f.offset] = v;
  }
}

// M/feature/BooleanFeature.scala:53-67
void boolean(const Feature &f, const Store &st, const CRequest &r, double *m, int D) {
  for (int i = 0; i < r.n_items; ++i) {
    double v = NaN;
    std::string key;
    if (read_key(r, f.scope, f.name, r.item_ids[i], key)) {
      const FeatureValue *fv = st.get(key);
      if (fv && fv->kind == V_SCALAR && fv->scalar.tag == S_BOOLEAN) v = fv->scalar.b ? 1.0 : 0.0;
    }
    m[(size_t)i * D + f.offset] = v;
  }
}

// "\\s+".r.split(string).length  — M/feature/WordCountFeature.scala:73-76 (java.util.regex.Pattern.split:
// a leading empty token is kept when the input starts with whitespace, trailing empties are dropped,
// the empty string gives one token)
int token_count(const std::string &s) {
  auto is_ws = [](unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == 0x0B || c == '\f' || c == '\r'; };
  if (s.empty()) return 1;
  std::vector<std::string> toks;
  size_t i = 0, n = s.size();
  std::string cur;
  bool any_match = false;
  while (i < n) {
    if (is_ws((unsigned char)s[i])) {
      size_t j = i;
      while (j < n && is_ws((unsigned char)s[j])) ++j;
      any_match = true;
      toks.push_back(cur);
      cur.clear();
      i = j;
    } else {
      cur.push_back(s[i++]);
    }
  }
  if (!any_match) return 1;
  toks.push_back(cur);
  size_t len = toks.size();
  while (len > 0 && toks[len - 1].empty()) --len;
  return (int)len;
}

// M/feature/WordCountFeature.scala:53-71
void word_count(const Feature &f, const Store &st, const CRequest &r, double *m, int D) {
  for (int i = 0; i < r.n_items; ++i) {
    double v = NaN;
    if (f.scope == SC_RANKING) {
      const CField *fld = fields_map_get(r, f.field);
      if (fld && fld->type == F_STRING) v = token_count(fld->str);
    } else {
      std::string key;
      if (read_key(r, f.scope, f.name, r.item_ids[i], key)) {
        const FeatureValue *fv = st.get(key);
        if (fv && fv->kind == V_SCALAR && fv->scalar.tag == S_DOUBLE) v = fv->scalar.d;
      }
    }
    m[(size_t)i * D + f.offset] = v;
  }
}

// M/feature/NumVectorFeature.scala:58-73 + M/flow/ClickthroughQuery.scala:61-65 (arraycopy of values.length)
void vector_(const Feature &f, const Store &st, const CRequest &r, double *m, int D) {
  for (int i = 0; i < r.n_items; ++i) {
    double *row = m + (size_t)i * D + f.offset;
    std::string key;
    const FeatureValue *fv = nullptr;
    if (read_key(r, f.scope, f.name, r.item_ids[i], key)) fv = st.get(key);
    if (fv && fv->kind == V_SCALAR && fv->scalar.tag == S_DOUBLE_LIST) {
      const auto &dl = fv->scalar.dl;
      if ((int)dl.size() > f.dim) throw Thrown{2};  // would overrun the row (ArrayIndexOutOfBounds / neighbour clobber)
      for (size_t k = 0; k < dl.size(); ++k) row[k] = dl[k];  // cells beyond values.length keep 0.0
    } else {
      for (int k = 0; k < f.dim; ++k) row[k] = NaN;
    }
  }
}

// M/feature/StringFeature.scala:119-137 (index), M/util/OneHotEncoder.scala:11-22 (onehot)
void string_encode(const Feature &f, const std::vector<std::string> &vals, double *row) {
  if (f.kind == K_STRING_INDEX) {
    double idx = 0;
    if (!vals.empty()) {
      for (size_t k = 0; k < f.values.size(); ++k)
        if (f.values[k] == vals[0]) { idx = (double)(k + 1); break; }  // valueIndex: zipWithIndex.toMap => LAST duplicate wins
      // possibleValues.zipWithIndex.toMap keeps the last index of a duplicated value
      for (size_t k = 0; k < f.values.size(); ++k)
        if (f.values[k] == vals[0]) idx = (double)(k + 1);
    }
    row[0] = idx;
  } else {
    for (int k = 0; k < f.dim; ++k) row[k] = 0.0;
    for (const auto &v : vals) {
      for (size_t k = 0; k < f.values.size(); ++k)
        if (f.values[k] == v) { row[k] = 1.0; break; }  // indexOf: first match
    }
  }
}

// M/feature/StringFeature.scala:70-107
void string_(const Feature &f, const Store &st, const CRequest &r, double *m, int D) {
  if (f.field_is_ranking) {  // :87-93
    const CField *fld = fields_find(r, f.field);
    std::vector<std::string> vals;
    if (fld && fld->type == F_STRING) vals = {fld->str};
    else if (fld && fld->type == F_STRING_LIST) vals.assign(fld->strs, fld->strs + fld->n);
    std::vector<double> enc(f.dim);
    string_encode(f, vals, enc.data());
    for (int i = 0; i < r.n_items; ++i) std::copy(enc.begin(), enc.end(), m + (size_t)i * D + f.offset);
    return;
  }
  for (int i = 0; i < r.n_items; ++i) {
    double *row = m + (size_t)i * D + f.offset;
    std::vector<std::string> vals;
    bool overridden = false;
    ItemFields itf = item_fields(r, i);  // :96-99 collectFirst
    for (const CField *p = itf.begin; p != itf.end; ++p) {
      if (f.field != p->name) continue;
      if (p->type == F_STRING) { vals = {p->str}; overridden = true; break; }
      if (p->type == F_STRING_LIST) { vals.assign(p->strs, p->strs + p->n); overridden = true; break; }
    }
    if (!overridden) {
      std::string key;
      if (read_key(r, f.scope, f.name, r.item_ids[i], key)) {
        const FeatureValue *fv = st.get(key);  // :75-78 only SStringList matches
        if (fv && fv->kind == V_SCALAR && fv->scalar.tag == S_STRING_LIST) vals = fv->scalar.sl;
      }
    }
    string_encode(f, vals, row);
  }
}

// M/feature/InteractionCountFeature.scala:44-59
void interaction_count(const Feature &f, const Store &st, const CRequest &r, double *m, int D) {
  for (int i = 0; i < r.n_items; ++i) {
    double v = 0.0;
    std::string key;
    if (read_key(r, f.scope, f.name, r.item_ids[i], key)) {
      const FeatureValue *fv = st.get(key);
      if (fv && fv->kind == V_COUNTER) v = (double)fv->counter;
    }
    m[(size_t)i * D + f.offset] = v;
  }
}

// M/feature/WindowInteractionCountFeature.scala:50-63
void window_count(const Feature &f, const Store &st, const CRequest &r, double *m, int D) {
  for (int i = 0; i < r.n_items; ++i) {
    double *row = m + (size_t)i * D + f.offset;
    std::string key;
    const FeatureValue *fv = nullptr;
    if (read_key(r, f.scope, f.name, r.item_ids[i], key)) fv = st.get(key);
    if (fv && fv->kind == V_PERIODIC && (int)fv->periodic.size() == f.dim)
      for (int k = 0; k < f.dim; ++k) row[k] = (double)fv->periodic[k];
    else
      for (int k = 0; k < f.dim; ++k) row[k] = NaN;
  }
}

// Scala/Java Long division: truncation toward zero, ArithmeticException on zero divisor,
// Long.MinValue / -1 wraps.
int64_t long_div(int64_t a, int64_t b) {
  if (b == 0) throw Thrown{1};
  if (b == -1) return (int64_t)(0 - (uint64_t)a);
  return a / b;
}

// M/feature/RateFeature.scala:290-356
void rate(const Feature &f, const Store &st, const CRequest &r, double *m, int D) {
  const std::string top_name = f.name + "_" + f.top, bottom_name = f.name + "_" + f.bottom;
  const std::string top_global = f.name + "_" + f.top + "_norm", bottom_global = f.name + "_" + f.bottom + "_norm";
  for (int i = 0; i < r.n_items; ++i) {
    double *row = m + (size_t)i * D + f.offset;
    for (int k = 0; k < f.dim; ++k) row[k] = NaN;
    std::string scope;  // ScopeCodec.encode of the target scope; empty => None
    const std::string item = r.item_ids[i];
    if (f.scope == SC_ITEM) {
      scope = "item=" + item;
    } else if (f.scope == SC_ITEM_FIELD) {  // :297-301
      const FeatureValue *fld = st.get(key_item(item, f.name + "_field"));
      if (fld && fld->kind == V_SCALAR && fld->scalar.tag == S_STRING) scope = "field=" + f.scope_field + ":" + fld->scalar.s;
    } else if (f.scope == SC_RANKING_FIELD) {  // :302-310
      const CField *rf = fields_map_get(r, f.scope_field);
      if (rf && rf->type == F_STRING) scope = "irf=" + f.scope_field + ":" + rf->str + ":" + item;
    }
    if (scope.empty()) continue;
    const FeatureValue *t = st.get(scope + "/" + top_name);
    const FeatureValue *b = st.get(scope + "/" + bottom_name);
    if (!t || !b) continue;
    if (!f.normalize) {
      if (t->kind != V_PERIODIC || (int)t->periodic.size() != f.dim) continue;
      if (b->kind != V_PERIODIC || (int)b->periodic.size() != f.dim) continue;
      for (int k = 0; k < f.dim; ++k) row[k] = (double)t->periodic[k] / (double)b->periodic[k];  // :325 Long / Double
    } else {
      const FeatureValue *tg = st.get(key_global(top_global));
      const FeatureValue *bg = st.get(key_global(bottom_global));
      if (!tg || !bg) continue;
      if (t->kind != V_PERIODIC || (int)t->periodic.size() != f.dim) continue;
      if (b->kind != V_PERIODIC || (int)b->periodic.size() != f.dim) continue;
      if (tg->kind != V_PERIODIC || (int)tg->periodic.size() != f.dim) continue;
      if (bg->kind != V_PERIODIC || (int)bg->periodic.size() != f.dim) continue;
      for (int k = 0; k < f.dim; ++k) {
        // :346-348  (w + top) / (w * (bottomGlobal / topGlobal) + bottom), inner division is Long / Long
        int64_t ratio = long_div(bg->periodic[k], tg->periodic[k]);
        row[k] = (f.weight + (double)t->periodic[k]) / (f.weight * (double)ratio + (double)b->periodic[k]);
      }
    }
  }
}

// M/feature/InteractedWithFeature.scala:133-164
void interacted_with(const Feature &f, const Store &st, const CRequest &r, double *m, int D) {
  // makeVisitorKey :125-129
  std::string visitor;
  if (f.scope == SC_SESSION && r.session) visitor = std::string("session=") + r.session + "/" + f.name + "_interactions";
  else if (f.scope == SC_USER && r.user) visitor = std::string("user=") + r.user + "/" + f.name + "_interactions";
  const size_t nf = f.values.size();
  std::vector<std::unordered_map<std::string, int>> hist(nf);
  std::vector<bool> have(nf, false);
  if (!visitor.empty()) {
    const FeatureValue *bl = st.get(visitor);
    if (bl && bl->kind == V_BOUNDED_LIST) {
      for (size_t fi = 0; fi < nf; ++fi) {
        have[fi] = true;
        for (const Scalar &s : bl->list) {
          if (s.tag != S_STRING) continue;
          const FeatureValue *fv = st.get(key_item(s.s, f.name + "_" + f.values[fi]));
          if (fv && fv->kind == V_SCALAR && fv->scalar.tag == S_STRING_LIST)
            for (const auto &tok : fv->scalar.sl) hist[fi][tok] += 1;  // groupMapReduce(identity)(_ => 1)(_ + _)
        }
      }
    }
  }
  for (int i = 0; i < r.n_items; ++i) {
    double *row = m + (size_t)i * D + f.offset;
    for (size_t fi = 0; fi < nf; ++fi) {
      double cnt = 0.0;
      const FeatureValue *fv = st.get(key_item(r.item_ids[i], f.name + "_" + f.values[fi]));
      if (fv && fv->kind == V_SCALAR && fv->scalar.tag == S_STRING_LIST && have[fi])
        for (const auto &tok : fv->scalar.sl) {
          auto it = hist[fi].find(tok);
          cnt = cnt + (it == hist[fi].end() ? 0 : it->second);
        }
      row[fi] = cnt;
    }
  }
}

// org.apache.commons.math3.stat.descriptive.rank.Percentile (default EstimationType.LEGACY,
// NaNStrategy.REMOVED), evaluate(50.0)
double percentile50(std::vector<double> data) {
  if (data.empty()) return NaN;
  if (data.size() == 1) return data[0];
  std::vector<double> work;
  for (double v : data)
    if (!std::isnan(v)) work.push_back(v);
  if (work.empty()) return NaN;
  std::sort(work.begin(), work.end());
  const int length = (int)work.size();
  const double p = 50.0 / 100.0;
  double pos = p * (length + 1);  // LEGACY index: 0 for p == 0, length for p == 1
  double fpos = std::floor(pos);
  int intPos = (int)fpos;
  double dif = pos - fpos;
  if (pos < 1) return work[0];
  if (pos >= length) return work[length - 1];
  double lower = work[intPos - 1], upper = work[intPos];
  return lower + dif * (upper - lower);
}

// M/feature/DiversityFeature.scala:67-132
void diversity(const Feature &f, const Store &st, const CRequest &r, double *m, int D) {
  struct Present { int item; const Scalar *s; };
  std::vector<Present> fieldValues;
  for (int i = 0; i < r.n_items; ++i) {
    const FeatureValue *fv = st.get(key_item(r.item_ids[i], f.name));
    if (fv && fv->kind == V_SCALAR) fieldValues.push_back({i, &fv->scalar});
  }
  auto fill = [&](double v) { for (int i = 0; i < r.n_items; ++i) m[(size_t)i * D + f.offset] = v; };
  if (fieldValues.empty()) { fill(0.0); return; }  // emptyResponse
  const int head = fieldValues[0].s->tag;
  if (head == S_STRING || head == S_STRING_LIST) {
    // featureMap = features.toMap keyed by item id: equal ids carry equal state, so per-position lookup is the same
    std::unordered_map<std::string, int> counts;
    double sum = 0.0;
    int taken = 0;
    std::map<std::string, const Scalar *> by_id;
    for (const Present &p : fieldValues) {
      if (p.s->tag != S_STRING && p.s->tag != S_STRING_LIST) continue;
      by_id[r.item_ids[p.item]] = p.s;
      if (taken < f.div_top) {
        ++taken;
        if (p.s->tag == S_STRING) counts[p.s->s] += 1;
        else for (const auto &t : p.s->sl) counts[t] += 1;
      }
    }
    for (auto &kv : counts) sum += kv.second;
    for (int i = 0; i < r.n_items; ++i) {
      auto it = by_id.find(r.item_ids[i]);
      if (it == by_id.end()) { m[(size_t)i * D + f.offset] = NaN; continue; }
      double w = 0.0;
      auto add = [&](const std::string &t) { auto c = counts.find(t); w = w + (c == counts.end() ? 0 : c->second); };
      if (it->second->tag == S_STRING) add(it->second->s);
      else for (const auto &t : it->second->sl) add(t);
      m[(size_t)i * D + f.offset] = w / sum;
    }
  } else if (head == S_DOUBLE) {
    std::vector<double> data;
    std::map<std::string, double> by_id;
    for (const Present &p : fieldValues) {
      if (p.s->tag != S_DOUBLE) continue;
      by_id[r.item_ids[p.item]] = p.s->d;
      if ((int)data.size() < f.div_top) data.push_back(p.s->d);
    }
    double median = percentile50(data);
    for (int i = 0; i < r.n_items; ++i) {
      auto it = by_id.find(r.item_ids[i]);
      m[(size_t)i * D + f.offset] = it == by_id.end() ? NaN : it->second - median;
    }
  } else {
    fill(0.0);  // "expected state to be string/number" -> emptyResponse
  }
}

// java.lang.Math.round(double)
int64_t java_round(double a) {
  if (std::isnan(a)) return 0;
  if (a >= 9223372036854775807.0) return INT64_MAX;
  if (a <= -9223372036854775808.0) return INT64_MIN;
  if (std::fabs(a) >= 4503599627370496.0) return (int64_t)a;
  double fl = std::floor(a);
  return (int64_t)fl + ((a - fl) >= 0.5 ? 1 : 0);
}

// M/feature/ItemAgeFeature.scala:73-83, M/model/Timestamp.scala:22-24
void item_age(const Feature &f, const Store &st, const CRequest &r, double *m, int D) {
  for (int i = 0; i < r.n_items; ++i) {
    double v = NaN;
    // value() reads Key(ItemScope(id), name) whatever the configured scope; valueKeys loads
    // conf.readKeys, so a non-item scope never has that key in the request state.
    const FeatureValue *fv = f.scope == SC_ITEM ? st.get(key_item(r.item_ids[i], f.name)) : nullptr;
    if (fv && fv->kind == V_SCALAR && fv->scalar.tag == S_DOUBLE) {
      int64_t updated = java_round(fv->scalar.d * 1000);
      uint64_t d = (uint64_t)r.timestamp_ms - (uint64_t)updated;  // other.ts - ts, wrapping
      int64_t diff = (int64_t)d;
      if (diff < 0) diff = (int64_t)(0 - (uint64_t)diff);  // math.abs (Long.MinValue stays)
      // FiniteDuration(length, MILLISECONDS) requires |length| <= Long.MaxValue / 1e6
      if (diff < 0 || diff > INT64_MAX / 1000000) throw Thrown{3};
      v = (double)(diff / 1000);  // toSeconds truncates
    }
    m[(size_t)i * D + f.offset] = v;
  }
}

// M/feature/PositionFeature.scala:30-35 (OnlineInference) ; M/feature/LocalDateTimeFeature.scala and the
// host-computed request-level features: one value vector for every item.
void constant(const Feature &f, const CRequest &r, double *m, int D) {
  std::vector<double> vals(f.dim, NaN);
  if (f.kind == K_POSITION) {
    vals[0] = f.position;
  } else {
    const CField *fld = fields_map_get(r, f.ext_field);
    if (fld && fld->type == F_NUMBER && f.dim == 1) vals[0] = fld->num;
    else if (fld && fld->type == F_NUMBER_LIST) {
      if (fld->n != f.dim) throw Thrown{2};
      for (int k = 0; k < f.dim; ++k) vals[k] = fld->nums[k];
    }
  }
  for (int i = 0; i < r.n_items; ++i) std::copy(vals.begin(), vals.end(), m + (size_t)i * D + f.offset);
}

// ---- java.time restated for M/feature/LocalDateTimeFeature.scala:31-93 ---------------------------
int64_t floor_div(int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
int64_t floor_mod(int64_t a, int64_t b) { return a - floor_div(a, b) * b; }
struct LocalDT { int64_t year; int month, dow; int64_t second_of_day; int64_t epoch_second; };
// local = fields of the date-time in its own offset; epoch_second = instant
LocalDT from_epoch(int64_t epoch_second, int64_t offset_seconds) {
  int64_t local = epoch_second + offset_seconds;
  int64_t days = floor_div(local, 86400);
  LocalDT o;
  o.second_of_day = floor_mod(local, 86400);
  o.epoch_second = epoch_second;
  o.dow = (int)floor_mod(days + 3, 7) + 1;  // 1970-01-01 is a Thursday (ISO 4)
  // civil from days (proleptic Gregorian), days since 1970-01-01
  int64_t z = days + 719468;
  int64_t era = floor_div(z, 146097);
  int64_t doe = z - era * 146097;
  int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  int64_t y = yoe + era * 400;
  int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  int64_t mp = (5 * doy + 2) / 153;
  int m = (int)(mp < 10 ? mp + 3 : mp - 9);
  o.year = y + (m <= 2 ? 1 : 0);
  o.month = m;
  return o;
}
int64_t days_from_civil(int64_t y, int m, int d) {
  y -= m <= 2;
  int64_t era = floor_div(y, 400);
  int64_t yoe = y - era * 400;
  int64_t doy = (153 * (m > 2 ? m - 3 : m + 9) + 2) / 5 + d - 1;
  int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + doe - 719468;
}
// ZonedDateTime.parse(value, ISO_DATE_TIME) for the offset forms: yyyy-MM-ddTHH:mm[:ss[.fraction]](Z|+HH:mm[:ss]|+HH)
// Region ids ("[Europe/Paris]") need the tz database and are reported as unparseable (the product does the same).
bool parse_iso(const std::string &s, LocalDT &out) {
  int Y, M, D, h, mi;
  int pos = 0;
  if (sscanf(s.c_str(), "%d-%2d-%2dT%2d:%2d%n", &Y, &M, &D, &h, &mi, &pos) != 5) return false;
  const char *p = s.c_str() + pos;
  int sec = 0;
  if (*p == ':') { int n = 0; if (sscanf(p, ":%2d%n", &sec, &n) != 1) return false; p += n; }
  if (*p == '.' || *p == ',') { ++p; while (*p >= '0' && *p <= '9') ++p; }
  int64_t off = 0;
  if (*p == 'Z' || *p == 'z') { ++p; }
  else if (*p == '+' || *p == '-') {
    int sign = *p == '-' ? -1 : 1; ++p;
    int oh = 0, om = 0, os = 0, n = 0;
    if (sscanf(p, "%2d%n", &oh, &n) != 1) return false;
    p += n;
    if (*p == ':') { if (sscanf(p, ":%2d%n", &om, &n) != 1) return false; p += n; }
    if (*p == ':') { if (sscanf(p, ":%2d%n", &os, &n) != 1) return false; p += n; }
    off = sign * (oh * 3600 + om * 60 + os);
  } else return false;  // ISO_DATE_TIME into a ZonedDateTime needs an offset or zone
  if (*p != 0) return false;
  if (M < 1 || M > 12 || D < 1 || D > 31 || h > 23 || mi > 59 || sec > 59) return false;
  int64_t local = days_from_civil(Y, M, D) * 86400 + h * 3600 + mi * 60 + sec;
  out = from_epoch(local - off, off);
  return true;
}
double map_dt(int mapper, const LocalDT &d) {
  switch (mapper) {
    case MAP_TIME_OF_DAY: return (double)d.second_of_day / 3600.0;  // second / SECONDS_IN_HOUR
    case MAP_DAY_OF_WEEK: return (double)d.dow;
    case MAP_MONTH_OF_YEAR: return (double)d.month;
    case MAP_YEAR: return (double)d.year;
    default: return (double)d.epoch_second;
  }
}
void local_time(const Feature &f, const CRequest &r, double *m, int D) {
  double v = NaN;
  if (f.field_is_ranking && f.field == "timestamp") {  // :36-39 Instant.ofEpochMilli(ts) at UTC
    v = map_dt(f.mapper, from_epoch(floor_div(r.timestamp_ms, 1000), 0));
  } else {  // :41-52
    const CField *fld = fields_map_get(r, f.field);
    LocalDT d;
    if (fld && fld->type == F_STRING && parse_iso(fld->str, d)) v = map_dt(f.mapper, d);
  }
  for (int i = 0; i < r.n_items; ++i) m[(size_t)i * D + f.offset] = v;
}

// M/feature/RelevancyFeature.scala:36-51 : item.fields.find(_.name == "relevancy").collectFirst{NumberField}
// K_ITEM_EXTERNAL: same shape for host-computed per-item features (number or number list field named ext_field)
void item_external(const Feature &f, const CRequest &r, double *m, int D) {
  const std::string fname = f.kind == K_RELEVANCY ? "relevancy" : f.ext_field;
  for (int i = 0; i < r.n_items; ++i) {
    double *row = m + (size_t)i * D + f.offset;
    for (int k = 0; k < f.dim; ++k) row[k] = NaN;
    ItemFields itf = item_fields(r, i);
    for (const CField *p = itf.begin; p != itf.end; ++p) {
      if (fname != p->name) continue;
      if (p->type == F_NUMBER && f.dim == 1) row[0] = p->num;
      else if (p->type == F_NUMBER_LIST && f.kind != K_RELEVANCY) {
        if (p->n != f.dim) throw Thrown{2};
        for (int k = 0; k < f.dim; ++k) row[k] = p->nums[k];
      }
      break;  // find(): the first field of that name decides
    }
  }
}

// M/ml/onnx/distance/DistanceFunction.scala:14-26 : query is Array[Float], item Array[Double]
double cosine(const float *q, int n, const double *item) {
  double topSum = 0.0, aSum = 0.0, bSum = 0.0;
  for (int i = 0; i < n; ++i) {
    topSum += (double)q[i] * item[i];
    aSum += (double)(float)(q[i] * q[i]);  // Float * Float is a Float product
    bSum += item[i] * item[i];
  }
  return topSum / (std::sqrt(aSum) * std::sqrt(bSum));
}

int double_compare(double a, double b) {  // java.lang.Double.compare
  if (a < b) return -1;
  if (a > b) return 1;
  int64_t ab, bb;
  double ca = std::isnan(a) ? NaN : a, cb = std::isnan(b) ? NaN : b;
  memcpy(&ab, &ca, 8);
  memcpy(&bb, &cb, 8);
  if (std::isnan(a)) ab = 0x7ff8000000000000LL;
  if (std::isnan(b)) bb = 0x7ff8000000000000LL;
  return ab == bb ? 0 : (ab < bb ? -1 : 1);
}

// M/ml/onnx/Normalize.scala:13-45
void normalize(int kind, std::vector<double> &v) {
  if (kind == N_MINMAX) {
    bool any = false;
    double mn = 0, mx = 0;
    for (double x : v) {
      if (std::isnan(x)) continue;
      if (!any) { mn = mx = x; any = true; }
      else { if (double_compare(x, mn) < 0) mn = x; if (double_compare(x, mx) > 0) mx = x; }
    }
    if (any) for (double &x : v) x = (x - mn) / (mx - mn);
  } else if (kind == N_POSITION) {
    const double size = (double)v.size();
    std::vector<int> idx(v.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return double_compare(v[a], v[b]) < 0; });
    std::vector<double> out(v);
    for (size_t s = 0; s < idx.size(); ++s)
      if (!std::isnan(v[idx[s]])) out[idx[s]] = (double)s / size;
    v.swap(out);
  }
}

// M/feature/FieldMatchBiencoderFeature.scala:80-109 with the query embedding supplied by the host
// (tokenizer + ONNX stay on the JVM) as a NumberListField named ext_field holding f32 values.
void biencoder(const Feature &f, const Store &st, const CRequest &r, double *m, int D) {
  const CField *q = fields_map_get(r, f.ext_field);
  std::vector<double> raw(r.n_items, NaN);
  if (q && q->type == F_NUMBER_LIST) {
    std::vector<float> qf(q->n);
    for (int k = 0; k < q->n; ++k) qf[k] = (float)q->nums[k];
    for (int i = 0; i < r.n_items; ++i) {
      const FeatureValue *fv = st.get(key_item(r.item_ids[i], f.name));
      if (fv && fv->kind == V_SCALAR && fv->scalar.tag == S_DOUBLE_LIST) {
        if ((int)fv->scalar.dl.size() < q->n) throw Thrown{2};  // ArrayIndexOutOfBounds in the reference
        raw[i] = cosine(qf.data(), q->n, fv->scalar.dl.data());
      }
    }
    normalize(f.norm, raw);
  }
  for (int i = 0; i < r.n_items; ++i) m[(size_t)i * D + f.offset] = raw[i];
}

void assemble(const Plan &p, const Store &st, const CRequest &r, double *m) {
  const int D = p.dim;
  std::fill(m, m + (size_t)r.n_items * D, 0.0);  // new Array[Double](dim), ClickthroughQuery.scala:51
  for (const Feature &f : p.features) {
    switch (f.kind) {
      case K_NUMBER: number(f, st, r, m, D); break;
      case K_BOOLEAN: boolean(f, st, r, m, D); break;
      case K_WORD_COUNT: word_count(f, st, r, m, D); break;
      case K_VECTOR: vector_(f, st, r, m, D); break;
      case K_STRING_INDEX: case K_STRING_ONEHOT: string_(f, st, r, m, D); break;
      case K_INTERACTION_COUNT: interaction_count(f, st, r, m, D); break;
      case K_WINDOW_COUNT: window_count(f, st, r, m, D); break;
      case K_RATE: rate(f, st, r, m, D); break;
      case K_INTERACTED_WITH: interacted_with(f, st, r, m, D); break;
      case K_DIVERSITY: diversity(f, st, r, m, D); break;
      case K_ITEM_AGE: item_age(f, st, r, m, D); break;
      case K_POSITION: case K_CONST: constant(f, r, m, D); break;
      case K_RELEVANCY: case K_ITEM_EXTERNAL: item_external(f, r, m, D); break;
      case K_BIENCODER: biencoder(f, st, r, m, D); break;
      case K_LOCAL_TIME: local_time(f, r, m, D); break;
    }
  }
}

}  // namespace

extern "C" {

// ---- store ----
void *orc_store_new() { return new Store(); }
void orc_store_free(void *s) { delete (Store *)s; }
int64_t orc_store_size(void *s) { return (int64_t)((Store *)s)->kv.size(); }
void orc_store_delete(void *s, const char *key) { ((Store *)s)->kv.erase(key); }
static FeatureValue &slot(void *s, const char *key, int kind) {
  FeatureValue &fv = ((Store *)s)->kv[key];
  fv = FeatureValue();
  fv.kind = kind;
  return fv;
}
void orc_put_double(void *s, const char *key, double v) { auto &f = slot(s, key, V_SCALAR); f.scalar.tag = S_DOUBLE; f.scalar.d = v; }
void orc_put_bool(void *s, const char *key, int v) { auto &f = slot(s, key, V_SCALAR); f.scalar.tag = S_BOOLEAN; f.scalar.b = v != 0; }
void orc_put_string(void *s, const char *key, const char *v) { auto &f = slot(s, key, V_SCALAR); f.scalar.tag = S_STRING; f.scalar.s = v; }
void orc_put_string_list(void *s, const char *key, const char *const *v, int n) {
  auto &f = slot(s, key, V_SCALAR); f.scalar.tag = S_STRING_LIST; f.scalar.sl.assign(v, v + n);
}
void orc_put_double_list(void *s, const char *key, const double *v, int n) {
  auto &f = slot(s, key, V_SCALAR); f.scalar.tag = S_DOUBLE_LIST; f.scalar.dl.assign(v, v + n);
}
void orc_put_counter(void *s, const char *key, int64_t v) { auto &f = slot(s, key, V_COUNTER); f.counter = v; }
void orc_put_periodic(void *s, const char *key, const int64_t *v, int n) { auto &f = slot(s, key, V_PERIODIC); f.periodic.assign(v, v + n); }
void orc_put_bounded_list(void *s, const char *key, const char *const *v, int n) {
  auto &f = slot(s, key, V_BOUNDED_LIST);
  for (int i = 0; i < n; ++i) { Scalar sc; sc.tag = S_STRING; sc.s = v[i]; f.list.push_back(sc); }
}

// ---- plan ----
void *orc_plan_new() { return new Plan(); }
void orc_plan_free(void *p) { delete (Plan *)p; }
int orc_plan_dim(void *p) { return ((Plan *)p)->dim; }
// generic feature registration; unused parameters are ignored per kind.  strs = `values` list
// (string: possible values, interacted_with: field names).
int orc_plan_add(void *pp, int kind, const char *name, int scope, const char *scope_field, const char *field,
                 int field_is_ranking, int dim, const char *const *strs, int n_strs, const char *top,
                 const char *bottom, int normalize, double weight, int div_top, double position, int norm,
                 const char *ext_field, int mapper) {
  Plan *p = (Plan *)pp;
  Feature f;
  f.kind = kind;
  f.name = name ? name : "";
  f.scope = scope;
  f.scope_field = scope_field ? scope_field : "";
  f.field = field ? field : "";
  f.field_is_ranking = field_is_ranking != 0;
  f.dim = dim;
  if (strs) f.values.assign(strs, strs + n_strs);
  f.top = top ? top : "";
  f.bottom = bottom ? bottom : "";
  f.normalize = normalize != 0;
  f.weight = weight;
  f.div_top = div_top;
  f.position = position;
  f.norm = norm;
  f.ext_field = ext_field ? ext_field : "";
  f.mapper = mapper;
  f.offset = p->dim;
  p->dim += dim;
  p->features.push_back(f);
  return f.offset;
}

// ClickthroughQuery(ItemValue.fromState(...)).values : n_items * dim row-major f64.
// returns 0, or the reference's exception class: 1 ArithmeticException, 2 IllegalState (dim), 3 IllegalArgument
int orc_assemble(void *plan, void *store, const CRequest *req, double *out_matrix) {
  try {
    assemble(*(Plan *)plan, *(Store *)store, *req, out_matrix);
    return 0;
  } catch (const Thrown &t) {
    return t.code;
  }
}

// Ranker.rerank's final ordering (M/ml/Ranker.scala:52-67): sortBy(-_.score) = stable sort by
// java.lang.Double.compare(-a, -b); out_order[k] = request index of the k-th response item.
void orc_sort_order(const double *scores, int n, int32_t *out_order) {
  std::vector<int32_t> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return double_compare(-scores[a], -scores[b]) < 0; });
  for (int i = 0; i < n; ++i) out_order[i] = idx[i];
}

int orc_token_count(const char *s) { return token_count(s); }
double orc_percentile50(const double *v, int n) { return percentile50(std::vector<double>(v, v + n)); }
int64_t orc_java_round(double v) { return java_round(v); }
void orc_normalize(int kind, double *v, int n) {
  std::vector<double> x(v, v + n);
  normalize(kind, x);
  std::copy(x.begin(), x.end(), v);
}
double orc_cosine(const float *q, int n, const double *item) { return cosine(q, n, item); }

}  // extern "C"
