// Model readers: LightGBM model string, XGBoost JSON/UBJSON, Metarank container; forest packing.
//
// The reference reaches these formats through third-party natives that are not vendored
// (ltrlib 0.2.6 -> xgboost4j / lightgbm4j 4.6.0-1, build.sbt:57-58); the formats and the
// prediction semantics are restated from those libraries' published behaviour (SURVEY.md §8c,
// Appendix B).  The reference's own call sites: ml/rank/LambdaMARTRanker.scala:228-232 (load),
// :348 (predictMat).
#include "forest.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <stdexcept>

#include "json.hpp"

namespace mrk {

// Booster.weights() of ltrlib (reference call site ml/rank/LambdaMARTRanker.scala:391-392) = the library's feature importance.
// LightGBM, GBDT::FeatureImportance(num_iteration = 0, type) (src/boosting/gbdt.cpp): type 0 ("split") counts the splits with
// split_gain > 0, type 1 ("gain") adds their float gains into a double, trees in order, nodes in order; one entry per feature up
// to max_feature_idx.  XGBoost, GBTree::FeatureScore (src/gbm/gbtree.h): "weight" counts every split, "total_gain" adds
// RTreeNodeStat::loss_chg in FLOAT in RegTree::WalkTree order, "gain" = total_gain / weight (float); features never split on are absent from the map the JVM
// gets - 0 here.  Both restated from the libraries' published sources (absent from /root/reference): unpinned, like the scores.
void Forest::feature_importance(int type, double *out, int n) const {
  for (int i = 0; i < n; ++i) out[i] = 0.0;
  if (type != 0)
    for (const Tree &t : trees)
      if (!t.feat.empty() && t.gain.size() != t.feat.size()) throw UnsupportedModel("the model file carries no split gains (split_gain / loss_changes): only the split-count importance is available");
  if (backend == Backend::LightGBM) {
    for (const Tree &t : trees)
      for (size_t i = 0; i < t.feat.size(); ++i) {
        const bool has = t.gain.size() == t.feat.size();
        if (has && !(t.gain[i] > 0.f)) continue;           // gbdt.cpp: `if (split_gain(split_idx) > 0)`
        out[t.feat[i]] += type == 0 ? 1.0 : (double)t.gain[i];
      }
    return;
  }
  std::vector<float> total((size_t)n, 0.f);
  std::vector<uint64_t> count((size_t)n, 0);
  // float sums depend on the order: RegTree::WalkTree (include/xgboost/tree_model.h) is a stack walk from the root that pushes
  // the left child, then the right one - so it visits a node, then its RIGHT subtree, then its left subtree
  std::vector<int32_t> stack;
  for (const Tree &t : trees) {
    if (t.feat.empty()) continue;
    stack.assign(1, 0);
    while (!stack.empty()) {
      const int32_t i = stack.back();
      stack.pop_back();
      count[(size_t)t.feat[(size_t)i]] += 1;
      if (type != 0) total[(size_t)t.feat[(size_t)i]] += t.gain[(size_t)i];
      if (t.left[(size_t)i] >= 0) stack.push_back(t.left[(size_t)i]);
      if (t.right[(size_t)i] >= 0) stack.push_back(t.right[(size_t)i]);
    }
  }
  for (int i = 0; i < n; ++i) {
    if (!count[(size_t)i]) continue;
    out[i] = type == 0 ? (double)count[(size_t)i] : type == 2 ? (double)total[(size_t)i] : (double)(total[(size_t)i] / (float)count[(size_t)i]);
  }
}

int64_t Forest::n_nodes() const {
  int64_t n = 0;
  for (auto &t : trees) n += (int64_t)t.feat.size();
  return n;
}
int64_t Forest::n_leaves() const {
  int64_t n = 0;
  for (auto &t : trees) n += (int64_t)t.leaf.size();
  return n;
}
int Forest::max_depth() const {
  int d = 0;
  for (auto &t : trees) d = std::max(d, t.depth);
  return d;
}
int64_t Forest::n_categorical() const {
  int64_t n = 0;
  for (auto &t : trees)
    for (auto f : t.flags) n += (f & NF_CATEGORICAL) ? 1 : 0;
  return n;
}

// feature indices index the caller's matrix row; beyond this a model is corrupt, and index + 1 must not overflow
constexpr int MAX_FEATURE_INDEX = 1 << 24;
constexpr int64_t MAX_CATEGORY = 1 << 24;

static int tree_depth(const Tree &t) {
  if (t.feat.empty()) return 0;
  // iterative DFS; children indices are validated by the callers
  std::vector<std::pair<int, int>> st{{0, 1}};
  int best = 0;
  size_t visited = 0;
  while (!st.empty()) {
    auto [n, d] = st.back();
    st.pop_back();
    if (++visited > 4 * t.feat.size() + 8) throw std::runtime_error("tree has a cycle");
    best = std::max(best, d);
    if (t.left[n] >= 0) st.push_back({t.left[n], d + 1});
    if (t.right[n] >= 0) st.push_back({t.right[n], d + 1});
  }
  return best;
}

static void validate_tree(const Tree &t, int n_features_hint) {
  const int nn = (int)t.feat.size(), nl = (int)t.leaf.size();
  if (nl < 1) throw std::runtime_error("tree without leaves");
  for (int i = 0; i < nn; ++i) {
    auto chk = [&](int c) {
      if (c >= 0) {
        if (c >= nn) throw std::runtime_error("child index out of range");
      } else if (~c >= nl) {
        throw std::runtime_error("leaf index out of range");
      }
    };
    chk(t.left[i]);
    chk(t.right[i]);
    if (t.feat[i] < 0) throw std::runtime_error("negative split feature");
    if (t.feat[i] >= MAX_FEATURE_INDEX) throw std::runtime_error("split feature index out of range");
    (void)n_features_hint;
  }
  // a TREE: every internal node but the root and every leaf hangs under exactly one parent.  (Shared children pass
  // the index checks above but make the in-order walk of pack_forest_qs visit more leaves than the tree has.)
  std::vector<uint8_t> node_refs((size_t)nn, 0), leaf_refs((size_t)nl, 0);
  for (int i = 0; i < nn; ++i)
    for (int c : {t.left[i], t.right[i]}) {
      uint8_t &r = c >= 0 ? node_refs[(size_t)c] : leaf_refs[(size_t)~c];
      if (++r > 1) throw std::runtime_error("malformed tree: a node has two parents");
    }
  if (nn > 0) {
    if (node_refs[0] != 0) throw std::runtime_error("malformed tree: the root is somebody's child");
    for (int i = 1; i < nn; ++i)
      if (node_refs[(size_t)i] != 1) throw std::runtime_error("malformed tree: an internal node is not reachable from the root");
    for (int i = 0; i < nl; ++i)
      if (leaf_refs[(size_t)i] != 1) throw std::runtime_error("malformed tree: a leaf is not reachable from the root");
  }
}

// ------------------------------------------------------------------ LightGBM text model

namespace {

struct Lines {
  const char *p, *end;
  bool next(std::string &line) {
    if (p >= end) return false;
    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
    const char *e = nl ? nl : end;
    const char *le = e;
    if (le > p && le[-1] == '\r') --le;
    line.assign(p, le);
    p = nl ? nl + 1 : end;
    return true;
  }
};

template <typename T, typename F>
std::vector<T> split_parse(const std::string &s, F conv) {
  std::vector<T> out;
  const char *p = s.c_str();
  while (*p) {
    while (*p == ' ') ++p;
    if (!*p) break;
    const char *b = p;
    while (*p && *p != ' ') ++p;
    out.push_back(conv(std::string(b, p)));
  }
  return out;
}

double parse_f64(const std::string &s) {
  // LightGBM writes %.17g and reads with a correctly rounded parser; "inf"/"nan" are accepted by strtod
  return strtod(s.c_str(), nullptr);
}

}  // namespace

// Booster::Predict (C_API_PREDICT_NORMAL, what lightgbm4j's predictForMat asks for) passes the summed leaves through the
// objective's ConvertOutput: the identity for lambdarank / rank_xendcg (what Metarank trains, LambdaMARTRanker.scala:148-170)
// and the plain regression losses; sigmoid / exp / softmax for the rest - not implemented here, so refused, not mis-scored.
static void check_lgbm_objective(const std::string &objective) {
  if (objective.empty()) return;
  const std::string name = objective.substr(0, objective.find(' '));
  const char *identity[] = {"lambdarank", "rank_xendcg", "regression", "regression_l2", "l2", "mean_squared_error", "mse", "regression_l1", "l1",
                            "mean_absolute_error", "mae", "huber", "fair", "quantile", "mape", "custom", "none"};
  bool ok = false;
  for (const char *o : identity) ok = ok || name == o;
  if (ok && objective.find("sqrt") != std::string::npos) ok = false;   // reg_sqrt: sign(x) * x^2 on the way out
  if (!ok) throw UnsupportedModel("lightgbm: objective '" + objective.substr(0, 64) + "' converts the raw score on output (sigmoid / exp / softmax): only identity-output objectives such as lambdarank are supported");
}

Forest parse_lightgbm_text(const char *text, size_t len) {
  Forest f;
  f.backend = Backend::LightGBM;
  Lines ln{text, text + len};
  std::string line;
  std::map<std::string, std::string> header;
  bool in_tree = false, saw_tree = false, done = false;
  std::map<std::string, std::string> kv;
  int num_class = 1, num_tree_per_iteration = 1;

  auto finish_tree = [&]() {
    if (!in_tree) return;
    Tree t;
    auto get = [&](const char *k) -> const std::string * {
      auto it = kv.find(k);
      return it == kv.end() ? nullptr : &it->second;
    };
    const std::string *nl = get("num_leaves");
    if (!nl) throw std::runtime_error("lightgbm: tree without num_leaves");
    int num_leaves = atoi(nl->c_str());
    if (num_leaves < 1) throw std::runtime_error("lightgbm: bad num_leaves");
    int num_cat = get("num_cat") ? atoi(get("num_cat")->c_str()) : 0;
    if (get("is_linear") && atoi(get("is_linear")->c_str()) != 0)
      throw UnsupportedModel("lightgbm: linear trees are not supported");
    const std::string *lv = get("leaf_value");
    if (!lv) throw std::runtime_error("lightgbm: tree without leaf_value");
    t.leaf = split_parse<double>(*lv, parse_f64);
    if ((int)t.leaf.size() != num_leaves) throw std::runtime_error("lightgbm: leaf_value length mismatch");
    if (num_leaves > 1) {
      auto need = [&](const char *k) -> const std::string & {
        const std::string *v = get(k);
        if (!v) throw std::runtime_error(std::string("lightgbm: tree without ") + k);
        return *v;
      };
      auto toi = [](const std::string &s) { return (int32_t)strtol(s.c_str(), nullptr, 10); };
      t.feat = split_parse<int32_t>(need("split_feature"), toi);
      t.thr = split_parse<double>(need("threshold"), parse_f64);
      std::vector<int32_t> dt = split_parse<int32_t>(need("decision_type"), toi);
      t.left = split_parse<int32_t>(need("left_child"), toi);
      t.right = split_parse<int32_t>(need("right_child"), toi);
      const size_t nn = (size_t)num_leaves - 1;
      if (const std::string *sg = get("split_gain")) {  // Tree::split_gain_ is std::vector<float> (include/LightGBM/tree.h)
        t.gain = split_parse<float>(*sg, [](const std::string &x) { return strtof(x.c_str(), nullptr); });
        if (t.gain.size() != nn) throw std::runtime_error("lightgbm: split_gain length mismatch");
      }
      if (t.feat.size() != nn || t.thr.size() != nn || dt.size() != nn || t.left.size() != nn ||
          t.right.size() != nn)
        throw std::runtime_error("lightgbm: split array length mismatch");
      std::vector<int32_t> cat_boundaries;
      std::vector<uint32_t> cat_threshold;
      if (num_cat > 0) {
        cat_boundaries = split_parse<int32_t>(need("cat_boundaries"), toi);
        cat_threshold = split_parse<uint32_t>(
            need("cat_threshold"), [](const std::string &s) { return (uint32_t)strtoul(s.c_str(), nullptr, 10); });
        if (cat_boundaries.size() != (size_t)num_cat + 1) throw std::runtime_error("lightgbm: cat_boundaries length");
      }
      t.flags.resize(nn);
      t.cat_begin.assign(nn, 0);
      t.cat_words.assign(nn, 0);
      for (size_t i = 0; i < nn; ++i) {
        // include/LightGBM/tree.h: kCategoricalMask = 1, kDefaultLeftMask = 2, missing type = (dt >> 2) & 3
        uint8_t fl = 0;
        int d = dt[i];
        if (d & 1) fl |= NF_CATEGORICAL;
        if (d & 2) fl |= NF_DEFAULT_LEFT;
        int mt = (d >> 2) & 3;
        if (mt == 1) fl |= NF_MISS_ZERO;
        else if (mt == 2) fl |= NF_MISS_NAN;
        t.flags[i] = fl;
        if (fl & NF_CATEGORICAL) {
          if (!(t.thr[i] >= 0.0 && t.thr[i] < (double)num_cat)) throw std::runtime_error("lightgbm: categorical threshold index out of range");
          int ci = (int)t.thr[i];
          if (ci < 0 || ci >= num_cat) throw std::runtime_error("lightgbm: categorical threshold index out of range");
          int b = cat_boundaries[ci], e = cat_boundaries[ci + 1];
          if (b < 0 || e < b || (size_t)e > cat_threshold.size()) throw std::runtime_error("lightgbm: cat_boundaries out of range");
          t.cat_begin[i] = (uint32_t)f.cat_bits.size();
          t.cat_words[i] = (uint32_t)(e - b);
          f.cat_bits.insert(f.cat_bits.end(), cat_threshold.begin() + b, cat_threshold.begin() + e);
        }
      }
    }
    validate_tree(t, 0);
    t.depth = tree_depth(t);
    f.trees.push_back(std::move(t));
    kv.clear();
    in_tree = false;
  };

  while (!done && ln.next(line)) {
    if (line.empty()) {
      finish_tree();
      continue;
    }
    if (line.rfind("Tree=", 0) == 0) {
      finish_tree();
      in_tree = true;
      saw_tree = true;
      continue;
    }
    if (line == "end of trees") {
      finish_tree();
      done = true;
      break;
    }
    size_t eq = line.find('=');
    if (in_tree) {
      if (eq != std::string::npos) kv[line.substr(0, eq)] = line.substr(eq + 1);
    } else if (!saw_tree) {
      if (eq != std::string::npos) header[line.substr(0, eq)] = line.substr(eq + 1);
      else header[line] = "";
    }
  }
  finish_tree();
  if (header.find("tree") == header.end() && header.find("version") == header.end() && f.trees.empty())
    throw std::runtime_error("lightgbm: not a LightGBM model string");
  if (header.count("num_class")) num_class = atoi(header["num_class"].c_str());
  if (header.count("num_tree_per_iteration")) num_tree_per_iteration = atoi(header["num_tree_per_iteration"].c_str());
  if (num_class != 1 || num_tree_per_iteration != 1)
    throw UnsupportedModel("lightgbm: only single-output models are supported (num_class=1)");
  if (header.count("max_feature_idx")) {
    const long mfi = strtol(header["max_feature_idx"].c_str(), nullptr, 10);
    if (mfi < -1 || mfi >= MAX_FEATURE_INDEX) throw std::runtime_error("lightgbm: max_feature_idx out of range");
    f.n_features = (int)mfi + 1;
  }
  if (header.count("objective")) f.objective = header["objective"];
  check_lgbm_objective(f.objective);
  f.average_output = header.count("average_output") > 0;
  for (auto &t : f.trees)
    for (auto ft : t.feat) f.n_features = std::max(f.n_features, ft + 1);
  return f;
}

// ------------------------------------------------------------------ XGBoost JSON / UBJSON

namespace {

// one XGBoost tree as the library keeps it: leaves and internal nodes in one array
struct XgbRawTree {
  std::vector<int32_t> lc, rc, split_index;   // children (-1 = leaf), split feature
  std::vector<float> cond;                    // split condition, or the leaf value of a leaf
  std::vector<float> loss_chg;                // RTreeNodeStat::loss_chg per node (empty: the file has none)
  std::vector<uint8_t> default_left, split_type;
  std::map<int, std::vector<int64_t>> categories;  // categorical node -> its categories
};

void check_objective(const std::string &objective) {
  if (objective.empty()) return;
  // Only identity-link objectives keep base_score == base margin and need no output transform.
  const char *ok[] = {"rank:pairwise", "rank:ndcg", "rank:map", "reg:squarederror", "reg:linear"};
  bool found = false;
  for (auto o : ok) found = found || objective == o;
  if (!found) throw UnsupportedModel("xgboost: objective '" + objective.substr(0, 64) + "' is not supported (need an identity-link objective such as rank:ndcg)");
}

// Renumber (internal nodes and leaves get their own index spaces, deleted nodes drop out) and append to the forest
void add_xgb_tree(Forest &f, const XgbRawTree &r) {
  const size_t n = r.lc.size();
  if (r.rc.size() != n || r.split_index.size() != n || r.cond.size() != n || r.default_left.size() != n)
    throw std::runtime_error("xgboost: node array length mismatch");
  if (n == 0) throw std::runtime_error("xgboost: empty tree");
  std::vector<int> inner_id(n, -1), leaf_id(n, -1);
  // reachability from root 0 (deleted nodes may linger in the arrays)
  std::vector<int> order;
  {
    std::vector<int> st{0};
    std::vector<char> seen(n, 0);
    while (!st.empty()) {
      int u = st.back();
      st.pop_back();
      if (u < 0 || (size_t)u >= n) throw std::runtime_error("xgboost: child index out of range");
      if (seen[u]) throw std::runtime_error("xgboost: tree has a cycle");
      seen[u] = 1;
      order.push_back(u);
      if (r.lc[u] != -1) {
        st.push_back(r.rc[u]);
        st.push_back(r.lc[u]);
      }
    }
  }
  Tree t;
  for (int u : order) {
    if (r.lc[u] == -1) {
      leaf_id[u] = (int)t.leaf.size();
      t.leaf.push_back((double)r.cond[u]);
    } else {
      inner_id[u] = (int)t.feat.size();
      t.feat.push_back(0);
    }
  }
  const size_t nn = t.feat.size();
  if (!r.loss_chg.empty() && r.loss_chg.size() != n) throw std::runtime_error("xgboost: loss_changes length mismatch");
  if (!r.loss_chg.empty()) t.gain.assign(nn, 0.f);
  t.thr.assign(nn, 0.0);
  t.flags.assign(nn, 0);
  t.left.assign(nn, 0);
  t.right.assign(nn, 0);
  t.cat_begin.assign(nn, 0);
  t.cat_words.assign(nn, 0);
  for (int u : order) {
    int id = inner_id[u];
    if (id < 0) continue;
    const int l = r.lc[u], rr = r.rc[u];
    t.feat[id] = r.split_index[u];
    if (!r.loss_chg.empty()) t.gain[id] = r.loss_chg[u];
    t.left[id] = inner_id[l] >= 0 ? inner_id[l] : ~leaf_id[l];
    t.right[id] = inner_id[rr] >= 0 ? inner_id[rr] : ~leaf_id[rr];
    uint8_t fl = NF_MISS_NAN;
    if (r.default_left[u]) fl |= NF_DEFAULT_LEFT;
    if (u < (int)r.split_type.size() && r.split_type[u] == 1) {
      fl |= NF_CATEGORICAL;
      auto it = r.categories.find(u);
      if (it == r.categories.end()) throw std::runtime_error("xgboost: categorical node without categories");
      int64_t maxc = -1;
      for (int64_t c : it->second) {
        if (c < 0) throw std::runtime_error("xgboost: negative category");
        if (c >= MAX_CATEGORY) throw std::runtime_error("xgboost: category out of range");  // (f32 holds integers up to 2^24 exactly)
        maxc = std::max<int64_t>(maxc, c);
      }
      uint32_t words = (uint32_t)(maxc < 0 ? 0 : (maxc / 32 + 1));
      t.cat_begin[id] = (uint32_t)f.cat_bits.size();
      t.cat_words[id] = words;
      f.cat_bits.resize(f.cat_bits.size() + words, 0u);
      for (int64_t c : it->second) {
        if (c < 0) throw std::runtime_error("xgboost: negative category");
        f.cat_bits[t.cat_begin[id] + (size_t)(c / 32)] |= 1u << (c % 32);
      }
    } else {
      t.thr[id] = (double)r.cond[u];
    }
    t.flags[id] = fl;
  }
  validate_tree(t, f.n_features);
  t.depth = tree_depth(t);
  f.trees.push_back(std::move(t));
}

// ---- the legacy binary serialisation (what Booster.toByteArray() of xgboost4j before 2.0 emits by default; still
// read by every later version: learner.cc LearnerImpl::LoadModel, gbtree_model.cc GBTreeModel::Load, tree_model.cc
// RegTree::Load).  Raw little-endian structs:
//   ["binf"]                                   optional 4-byte header
//   LearnerModelParamLegacy  136 B             f32 base_score, u32 num_feature, i32 num_class, i32 contain_extra_attrs,
//                                              i32 contain_eval_metrics, u32 major_version, u32 minor_version, ...
//   u64 n + n bytes                            objective name;  the same for the booster name ("gbtree")
//   GBTreeModelParam         160 B             i32 num_trees, ..., i32 size_leaf_vector at +28
//   per tree: TreeParam      148 B             i32 num_roots, i32 num_nodes, i32 num_deleted, i32 max_depth, i32 num_feature,
//                                              i32 size_leaf_vector, i32 reserved[31]
//             num_nodes x Node        20 B     i32 parent, i32 cleft, i32 cright, u32 sindex (bit 31 = default_left),
//                                              f32 leaf_value | split_cond
//             num_nodes x NodeStat    16 B     skipped
//             [u64 n + n x f32]                leaf vector, only when size_leaf_vector != 0 (pre-1.0 files)
//   num_trees x i32                            tree_info (output group; must be 0)
// The format has no categorical splits.  Stated from the published source of XGBoost 1.x; no file written by the real
// library is available in this environment (tests/golden/README.md: `make_real_goldens.py` writes one where it is).
struct LegacyIn {
  const uint8_t *p, *end;
  void need(size_t n) const { if ((size_t)(end - p) < n) throw std::runtime_error("xgboost legacy model: truncated"); }
  template <typename T> T get() { need(sizeof(T)); T v; memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
  void skip(size_t n) { need(n); p += n; }
  std::string str() {
    const uint64_t n = get<uint64_t>();
    if (n > (uint64_t)(end - p)) throw std::runtime_error("xgboost legacy model: bad string length");
    std::string s((const char *)p, (size_t)n);
    p += n;
    return s;
  }
};

Forest parse_xgboost_legacy(const uint8_t *bytes, size_t len) {
  LegacyIn in{bytes, bytes + len};
  if (len >= 4 && memcmp(bytes, "binf", 4) == 0) in.skip(4);
  if (len >= 4 && memcmp(bytes, "bs64", 4) == 0) throw std::runtime_error("xgboost: base64 models are not supported");
  Forest f;
  f.backend = Backend::XGBoost;
  const uint8_t *mp = in.p;
  in.skip(136);
  float base_score;
  uint32_t num_feature, major;
  int32_t num_class;
  memcpy(&base_score, mp, 4);
  memcpy(&num_feature, mp + 4, 4);
  memcpy(&num_class, mp + 8, 4);
  memcpy(&major, mp + 20, 4);
  if (num_class > 1) throw UnsupportedModel("xgboost: multi-class models are not supported");
  if (num_feature > (1u << 24)) throw std::runtime_error("xgboost: not a legacy binary model (num_feature out of range)");
  f.n_features = (int)num_feature;
  f.objective = in.str();
  const std::string booster = in.str();
  if (booster != "gbtree") throw UnsupportedModel("xgboost: only gbtree boosters are supported (found '" + booster.substr(0, 32) + "')");
  check_objective(f.objective);
  f.base_score = (double)base_score;  // identity-link objectives: ProbToMargin (applied to pre-1.0 files) is the identity
  if (major > 3) throw std::runtime_error("xgboost: unknown serialisation (not JSON / UBJSON, and the legacy header carries major version " + std::to_string(major) + ")");
  const uint8_t *gp = in.p;
  in.skip(160);
  int32_t num_trees, gb_leaf_vec;
  memcpy(&num_trees, gp, 4);
  memcpy(&gb_leaf_vec, gp + 28, 4);
  if (num_trees < 0 || num_trees > (1 << 24)) throw std::runtime_error("xgboost legacy model: bad tree count");
  for (int t = 0; t < num_trees; ++t) {
    const uint8_t *tp = in.p;
    in.skip(148);
    int32_t num_roots, num_nodes, leaf_vec;
    memcpy(&num_roots, tp, 4);
    if (num_roots != 1) throw std::runtime_error("xgboost legacy model: unknown serialisation (a tree with num_roots = " + std::to_string(num_roots) + ")");
    memcpy(&num_nodes, tp + 4, 4);
    memcpy(&leaf_vec, tp + 20, 4);
    if (num_nodes <= 0 || (uint64_t)num_nodes * 36 > (uint64_t)(in.end - in.p)) throw std::runtime_error("xgboost legacy model: bad node count");
    XgbRawTree r;
    r.lc.resize(num_nodes); r.rc.resize(num_nodes); r.split_index.resize(num_nodes);
    r.cond.resize(num_nodes); r.default_left.resize(num_nodes);
    for (int i = 0; i < num_nodes; ++i) {
      (void)in.get<int32_t>();  // parent
      r.lc[i] = in.get<int32_t>();
      r.rc[i] = in.get<int32_t>();
      const uint32_t sindex = in.get<uint32_t>();
      r.cond[i] = in.get<float>();
      r.split_index[i] = (int32_t)(sindex & 0x7fffffffu);
      r.default_left[i] = (uint8_t)(sindex >> 31);
    }
    r.loss_chg.resize(num_nodes);
    for (int i = 0; i < num_nodes; ++i) {  // RTreeNodeStat: f32 loss_chg, f32 sum_hess, f32 base_weight, i32 leaf_child_cnt
      r.loss_chg[i] = in.get<float>();
      in.skip(12);
    }
    if (leaf_vec != 0) {
      const uint64_t n = in.get<uint64_t>();
      if (n > (uint64_t)(in.end - in.p) / 4) throw std::runtime_error("xgboost legacy model: bad leaf vector");
      in.skip((size_t)n * 4);
    }
    add_xgb_tree(f, r);
  }
  for (int t = 0; t < num_trees; ++t)
    if (in.get<int32_t>() != 0) throw std::runtime_error("xgboost: multi-group models are not supported");
  for (auto &t : f.trees)
    for (auto ft : t.feat) f.n_features = std::max(f.n_features, ft + 1);
  return f;
}

}  // namespace

Forest parse_xgboost(const uint8_t *bytes, size_t len) {
  if (len == 0) throw std::runtime_error("xgboost: empty model");
  json::Value root;
  // JSON text starts with '{' followed by whitespace or '"'; UBJSON starts with '{' followed by a length marker;
  // anything else is the legacy binary serialisation ("binf" header or the raw parameter struct).
  bool is_json = false;
  if (bytes[0] == '{') {
    size_t k = 1;
    while (k < len && (bytes[k] == ' ' || bytes[k] == '\n' || bytes[k] == '\r' || bytes[k] == '\t')) ++k;
    is_json = k < len && (bytes[k] == '"' || bytes[k] == '}');
  } else {
    return parse_xgboost_legacy(bytes, len);
  }
  root = is_json ? json::parse((const char *)bytes, len) : json::parse_ubjson(bytes, len);

  Forest f;
  f.backend = Backend::XGBoost;
  const json::Value &learner = root.at("learner");
  const json::Value &lmp = learner.at("learner_model_param");
  // base_score as every writer spells it: a JSON number, the string "5E-1" (1.x / 2.x), the bracketed string "[5E-1]" (3.x: one
  // value per target) or a JSON / UBJSON array of one value; margin space for rank:* objectives (identity link)
  {
    const json::Value &bs = lmp.at("base_score");
    if (bs.is_array()) {
      if (bs.arr.size() != 1) throw UnsupportedModel("xgboost: base_score with " + std::to_string(bs.arr.size()) + " values (multi-target models are not supported)");
      f.base_score = (double)bs.arr[0].as_float();
    } else {
      if (bs.is_string() && bs.as_string().find(',') != std::string::npos) throw UnsupportedModel("xgboost: base_score with several values (multi-target models are not supported)");
      f.base_score = (double)bs.as_float();
    }
    if (!std::isfinite(f.base_score)) throw std::runtime_error("xgboost: base_score is not finite");
  }
  if (const json::Value *nf = lmp.find("num_feature")) {
    const int64_t v = nf->as_int();
    if (v < 0 || v > MAX_FEATURE_INDEX) throw std::runtime_error("xgboost: num_feature out of range");
    f.n_features = (int)v;
  }
  if (const json::Value *nc = lmp.find("num_class"))
    if (nc->as_int() > 1) throw UnsupportedModel("xgboost: multi-class models are not supported");
  if (const json::Value *nt = lmp.find("num_target"))
    if (nt->as_int() > 1) throw UnsupportedModel("xgboost: multi-target models are not supported");
  if (const json::Value *obj = learner.find("objective"))
    if (const json::Value *nm = obj->find("name")) f.objective = nm->as_string();
  check_objective(f.objective);
  const json::Value &gb = learner.at("gradient_booster");
  if (const json::Value *nm = gb.find("name"))
    if (nm->as_string() != "gbtree")   // dart scales every tree by weight_drop, gblinear has no trees
      throw UnsupportedModel("xgboost: only gbtree boosters are supported (found '" + nm->as_string().substr(0, 32) + "')");
  if (gb.find("weight_drop")) throw UnsupportedModel("xgboost: dart boosters (weight_drop) are not supported");
  const json::Value &model = gb.at("model");
  const json::Value &trees = model.at("trees");
  if (!trees.is_array()) throw std::runtime_error("xgboost: trees is not an array");

  for (const json::Value &jt : trees.arr) {
    if (const json::Value *tp = jt.find("tree_param"))
      if (const json::Value *slv = tp->find("size_leaf_vector"))
        if (slv->as_int() > 1) throw UnsupportedModel("xgboost: vector leaves (size_leaf_vector > 1) are not supported");
    const auto &lc = jt.at("left_children").arr;
    const auto &rc = jt.at("right_children").arr;
    const auto &si = jt.at("split_indices").arr;
    const auto &sc = jt.at("split_conditions").arr;
    const auto &dl = jt.at("default_left").arr;
    const size_t n = lc.size();
    if (rc.size() != n || si.size() != n || sc.size() != n || dl.size() != n)
      throw std::runtime_error("xgboost: node array length mismatch");
    XgbRawTree r;
    r.lc.resize(n); r.rc.resize(n); r.split_index.resize(n); r.cond.resize(n); r.default_left.resize(n);
    r.split_type.assign(n, 0);
    for (size_t i = 0; i < n; ++i) {
      r.lc[i] = (int32_t)lc[i].as_int();
      r.rc[i] = (int32_t)rc[i].as_int();
      r.split_index[i] = (int32_t)si[i].as_int();
      r.cond[i] = sc[i].as_float();
      r.default_left[i] = dl[i].as_bool() ? 1 : 0;
    }
    if (const json::Value *st = jt.find("split_type"))
      for (size_t i = 0; i < n && i < st->arr.size(); ++i) r.split_type[i] = (uint8_t)st->arr[i].as_int();
    if (const json::Value *lch = jt.find("loss_changes"))
      if (lch->is_array() && lch->arr.size() == n) {
        r.loss_chg.resize(n);
        for (size_t i = 0; i < n; ++i) r.loss_chg[i] = lch->arr[i].as_float();
      }
    // categorical side tables
    const json::Value *cats = jt.find("categories");
    if (const json::Value *cn = jt.find("categories_nodes")) {
      const auto &segs = jt.at("categories_segments").arr;
      const auto &sizes = jt.at("categories_sizes").arr;
      for (size_t k = 0; k < cn->arr.size(); ++k) {
        const int64_t b = segs.at(k).as_int(), sz = sizes.at(k).as_int();
        if (!cats || b < 0 || sz < 0 || (uint64_t)b > cats->arr.size() || (uint64_t)sz > cats->arr.size() - (uint64_t)b)
          throw std::runtime_error("xgboost: categories segment out of range");
        std::vector<int64_t> &v = r.categories[(int)cn->arr[k].as_int()];
        for (int64_t j = 0; j < sz; ++j) v.push_back(cats->arr[(size_t)(b + j)].as_int());
      }
    }
    add_xgb_tree(f, r);
  }
  if (const json::Value *ti = model.find("tree_info"))
    for (auto &g : ti->arr)
      if (g.as_int() != 0) throw UnsupportedModel("xgboost: multi-group models are not supported");
  // one tree per boosting round is what the scorer's "base + sum of leaves" means without further thought: random-forest rounds
  // (num_parallel_tree > 1) and multi-output rounds are refused rather than assumed
  if (const json::Value *mp = model.find("gbtree_model_param")) {
    if (const json::Value *npt = mp->find("num_parallel_tree"))
      if (npt->as_int() > 1) throw UnsupportedModel("xgboost: num_parallel_tree > 1 is not supported");
    if (const json::Value *slv = mp->find("size_leaf_vector"))
      if (slv->as_int() > 1) throw UnsupportedModel("xgboost: vector leaves (size_leaf_vector > 1) are not supported");
  }
  if (const json::Value *ip = model.find("iteration_indptr"))
    for (size_t i = 0; i < ip->arr.size(); ++i)
      if (ip->arr[i].as_int() != (int64_t)i) throw UnsupportedModel("xgboost: more than one tree per boosting round (iteration_indptr) is not supported");
  for (auto &t : f.trees)
    for (auto ft : t.feat) f.n_features = std::max(f.n_features, ft + 1);
  return f;
}

// ------------------------------------------------------------------ Metarank container

namespace {
struct Rd {
  const uint8_t *p, *end;
  void need(size_t n) {
    if ((size_t)(end - p) < n) throw std::runtime_error("model container: truncated");
  }
  uint8_t u8() { need(1); return *p++; }
  int32_t i32() { need(4); int32_t v = (int32_t)((uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]); p += 4; return v; }
  uint16_t u16() { need(2); uint16_t v = (uint16_t)(p[0] << 8 | p[1]); p += 2; return v; }
  // java.io.DataInput.readUTF: u16 byte length + modified UTF-8 (U+0000 as C0 80, supplementary
  // chars as surrogate pairs of 3-byte sequences).  Feature names are plain identifiers; the bytes
  // are returned verbatim, which equals standard UTF-8 for everything inside the BMP except U+0000.
  std::string utf() { uint16_t n = u16(); need(n); std::string s((const char *)p, n); p += n; return s; }
};
}  // namespace

Container parse_container(const uint8_t *blob, size_t len) {
  Rd r{blob, blob + len};
  Container c;
  c.version = r.u8();
  if (c.version != 2 && c.version != 3)
    throw std::runtime_error("model container: unsupported bitstream version " + std::to_string(c.version));
  int32_t nf = r.i32();
  if (nf < 0 || nf > 1 << 20) throw std::runtime_error("model container: bad feature count");
  for (int i = 0; i < nf; ++i) c.features.push_back(r.utf());
  c.booster_tag = r.u8();
  if (c.booster_tag != 0 && c.booster_tag != 1)
    throw std::runtime_error("unsupported booster tag " + std::to_string(c.booster_tag));
  int32_t sz = r.i32();
  if (sz < 0) throw std::runtime_error("model container: bad booster size");
  r.need((size_t)sz);
  c.inner = r.p;
  c.inner_len = (size_t)sz;
  r.p += sz;
  // v3 appends warm-up requests: i32 count + RankingEventFormat records (LambdaMARTRanker.scala:219-224,384-388)
  if (c.version >= 3 && r.p < r.end) {
    c.n_warmup = r.i32();
    if (c.n_warmup < 0) throw std::runtime_error("model container: bad warm-up count");
    c.warmup = r.p;
    c.warmup_len = (size_t)(r.end - r.p);
  }
  return c;
}

// ------------------------------------------------------------------ packing

PackedForest pack_forest(const Forest &f, uint32_t chunk_bytes) {
  PackedForest pf;
  const bool f64 = f.backend == Backend::LightGBM;
  const uint32_t leaf_sz = f64 ? 8 : 4;
  auto tree_bytes = [&](const Tree &t) {
    uint32_t b = (uint32_t)t.feat.size() * 16 + (uint32_t)t.leaf.size() * leaf_sz;
    return (b + 15u) & ~15u;
  };
  pf.trees.resize(f.trees.size());
  ChunkRef cur{0, 0, 0, 0};
  auto close_chunk = [&]() {
    if (cur.n_trees == 0) return;
    pf.chunks.push_back(cur);
    pf.max_chunk_bytes = std::max(pf.max_chunk_bytes, cur.byte_len);
    pf.max_chunk_trees = std::max(pf.max_chunk_trees, cur.n_trees);
    cur = ChunkRef{cur.byte_off + cur.byte_len, 0, cur.first_tree + cur.n_trees, 0};
  };
  for (size_t ti = 0; ti < f.trees.size(); ++ti) {
    const Tree &t = f.trees[ti];
    if (t.feat.size() > 32767 || t.leaf.size() > 32768)
      throw std::runtime_error("tree too large for the packed node format (> 32767 internal nodes)");
    uint32_t tb = tree_bytes(t);
    if (tb > chunk_bytes) throw std::runtime_error("a single tree exceeds the LDS chunk budget");
    if (cur.byte_len + tb > chunk_bytes) close_chunk();
    TreeRef &tr = pf.trees[ti];
    tr.node_off = cur.byte_len;
    tr.leaf_off = cur.byte_len + (uint32_t)t.feat.size() * 16;
    tr.n_nodes = (uint16_t)t.feat.size();
    tr.depth = (uint16_t)t.depth;
    size_t base = pf.image.size();
    pf.image.resize(base + tb, 0);
    uint8_t *dst = pf.image.data() + base;
    for (size_t i = 0; i < t.feat.size(); ++i) {
      if (t.feat[i] > 65535) throw std::runtime_error("split feature index exceeds 65535");
      const uint8_t fl = t.flags[i];
      if (f64) {
        PackedNode64 n{};
        if (fl & NF_CATEGORICAL) {
          uint64_t bits = (uint64_t)t.cat_begin[i] | ((uint64_t)t.cat_words[i] << 32);
          memcpy(&n.thr, &bits, 8);
        } else {
          n.thr = t.thr[i];
        }
        n.feat = (uint16_t)t.feat[i];
        n.flags = fl;
        // where NaN goes for a numerical node (LightGBM Tree::NumericalDecision):
        //   missing None : NaN is replaced by 0.0 and compared
        //   missing Zero : NaN -> 0.0 -> IsZero -> default direction
        //   missing NaN  : default direction
        bool nan_left;
        if (fl & (NF_MISS_ZERO | NF_MISS_NAN)) nan_left = (fl & NF_DEFAULT_LEFT) != 0;
        else nan_left = 0.0 <= t.thr[i];
        n.nan_left = nan_left ? 1 : 0;
        n.left = (int16_t)t.left[i];
        n.right = (int16_t)t.right[i];
        memcpy(dst + i * 16, &n, 16);
      } else {
        PackedNode32 n{};
        if (fl & NF_CATEGORICAL) {
          uint32_t b = t.cat_begin[i];
          memcpy(&n.thr, &b, 4);
          n.pad2 = t.cat_words[i];
        } else {
          n.thr = (float)t.thr[i];
        }
        n.feat = (uint16_t)t.feat[i];
        n.flags = fl;
        n.left = (int16_t)t.left[i];
        n.right = (int16_t)t.right[i];
        memcpy(dst + i * 16, &n, 16);
      }
    }
    uint8_t *ldst = dst + t.feat.size() * 16;
    for (size_t i = 0; i < t.leaf.size(); ++i) {
      if (f64) {
        memcpy(ldst + i * 8, &t.leaf[i], 8);
      } else {
        float v = (float)t.leaf[i];
        memcpy(ldst + i * 4, &v, 4);
      }
    }
    cur.byte_len += tb;
    cur.n_trees += 1;
  }
  close_chunk();
  return pf;
}

// ------------------------------------------------------------------ bit-vector image (scorer "qs")

PackedForestQS pack_forest_qs(const Forest &f, int n_cols) {
  PackedForestQS pf;
  auto fail = [&](const std::string &why) { pf.ok = false; pf.why = why; return pf; };
  const bool f64 = f.backend == Backend::LightGBM;
  pf.f64 = f64;
  pf.n_trees = (int)f.trees.size();
  const uint32_t leaf_sz = f64 ? 8 : 4;
  int nf = std::max(n_cols, 0);
  for (auto &t : f.trees) {
    if (t.leaf.size() > (size_t)QS_LEAVES) return fail("a tree has more than 16 leaves");
    if (t.feat.size() > (size_t)QS_SLOTS - 1) return fail("a tree has more than 15 internal nodes");
    for (auto ft : t.feat) nf = std::max(nf, ft + 1);
  }
  if (f.trees.empty()) return fail("empty forest");
  if (nf > 4096) return fail("more than 4096 columns");

  // which view of its column a node reads
  auto node_kind = [&](const Tree &t, size_t i) -> int {
    const uint8_t fl = t.flags[i];
    if (fl & NF_CATEGORICAL) return QV_CAT;
    const bool dl = (fl & NF_DEFAULT_LEFT) != 0;
    if (!f64) return dl ? QV_NAN_LEFT : QV_NAN_RIGHT;  // XGBoost: NaN is "missing"
    // LightGBM Tree::NumericalDecision: MissingType None compares NaN as 0.0; Zero sends NaN and
    // |x| <= kZeroThreshold to the default side; NaN sends NaN to the default side.
    if (fl & NF_MISS_ZERO) return dl ? QV_MISS_LEFT : QV_MISS_RIGHT;
    if (fl & NF_MISS_NAN) return dl ? QV_NAN_LEFT : QV_NAN_RIGHT;
    return QV_NAN_ZERO;  // NaN is compared as 0.0
  };
  std::vector<std::vector<double>> tabs(nf);
  std::map<std::pair<int, int>, int> view_ids;  // (feature, kind) -> tile column; ordered => grouped by feature
  for (auto &t : f.trees)
    for (size_t i = 0; i < t.feat.size(); ++i) {
      if (!(t.flags[i] & NF_CATEGORICAL)) {
        if (t.thr[i] != t.thr[i]) return fail("NaN threshold");
        tabs[t.feat[i]].push_back(t.thr[i]);
      } else if ((uint64_t)t.cat_words[i] * 32 > QS_CAT_BEYOND) {
        return fail("categorical bitset with more than 32765 categories");
      }
      view_ids.emplace(std::make_pair(t.feat[i], node_kind(t, i)), 0);
    }
  pf.feats.assign(nf, QsFeature{0, 0, 0, 0, 0, 0, 0});
  for (int ft = 0; ft < nf; ++ft) {
    auto &v = tabs[ft];
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());  // -0.0 == +0.0: one entry, compared numerically on the device too
    if (v.size() > 32766) return fail("more than 32766 distinct thresholds on one column");
    pf.feats[ft].thr_off = (uint32_t)pf.thr.size();
    pf.feats[ft].thr_len = (uint16_t)v.size();
    pf.feats[ft].zero_bin = (uint16_t)(std::lower_bound(v.begin(), v.end(), 0.0) - v.begin());  // #{t < 0.0}: LightGBM only
    pf.thr.insert(pf.thr.end(), v.begin(), v.end());
    // +inf up to the next multiple of the staging chunk: the assembly kernels stage a table in whole chunks and search the
    // staged copy in a FIXED number of steps (qs_device.hpp qs_bin_search_staged) - an entry past the table compares as
    // "not below" for every value
    while (pf.thr.size() % QS_STAGE_CHUNK) pf.thr.push_back(std::numeric_limits<double>::infinity());
    // the compact copy (resident-table sinks): exact length - their search halves a compile-time length, every read inside the table
    pf.rt_off.push_back((uint32_t)pf.thr_rt.size());
    if (v.size() > 256) pf.rt_len.push_back(QS_RT_NONE);
    else {
      pf.rt_len.push_back((uint32_t)v.size());
      pf.thr_rt.insert(pf.thr_rt.end(), v.begin(), v.end());
    }
  }
  if (pf.thr_rt.size() % 2) pf.thr_rt.push_back(std::numeric_limits<double>::infinity());   // copied 16 bytes at a time
  {
    int id = 0, cur = -1;
    for (auto &kv : view_ids) {
      kv.second = id;
      pf.views.push_back(QsView{(uint16_t)kv.first.first, (uint8_t)kv.first.second, 0});
      if (kv.first.first != cur) {
        pf.feats[kv.first.first].view_begin = (uint8_t)id;
        cur = kv.first.first;
      }
      if (id >= QS_MAX_VIEWS) return fail("more than 255 tile columns");
      QsFeature &qf = pf.feats[kv.first.first];
      qf.view_end = (uint8_t)(id + 1);
      qf.view_kinds |= (uint32_t)kv.first.second << (4 * (id - qf.view_begin));  // <= 6 kinds per column
      ++id;
    }
  }

  if (pf.views.size() > (size_t)QS_MAX_VIEWS) return fail("more than 255 tile columns");
  pf.nodes.assign((size_t)(pf.n_trees + 1) * QS_TREE_WORDS, 0);
  pf.leaves.assign((size_t)pf.n_trees * QS_LEAVES * leaf_sz, 0);
  std::vector<int> lo, hi;  // per internal node: leaf positions [lo, hi) of its left subtree
  for (size_t ti = 0; ti < f.trees.size(); ++ti) {
    const Tree &t = f.trees[ti];
    const size_t ni = t.feat.size();
    uint32_t *nd = pf.nodes.data() + ti * QS_TREE_WORDS;  // zero: mask 0 never changes the bit vector
    uint8_t *lv = pf.leaves.data() + ti * QS_LEAVES * leaf_sz;
    auto put_leaf = [&](int pos, double v) {
      if (pos < 0 || pos >= QS_LEAVES) throw std::runtime_error("malformed tree: more leaf positions than leaves");  // (validate_tree rules it out)
      if (f64) memcpy(lv + (size_t)pos * 8, &v, 8);
      else { float x = (float)v; memcpy(lv + (size_t)pos * 4, &x, 4); }
    };
    const uint32_t cat_first = (uint32_t)pf.cat_nodes.size();  // first categorical node of this tree
    if (cat_first >= (1u << 24)) return fail("more than 2^24 categorical nodes");
    nd[QS_SLOTS - 1] = cat_first;
    if (ni == 0) {
      put_leaf(0, t.leaf.empty() ? 0.0 : t.leaf[0]);
      continue;
    }
    // left-to-right leaf positions by an explicit in-order walk (left subtree first)
    lo.assign(ni, 0);
    hi.assign(ni, 0);
    int pos = 0;
    std::vector<std::pair<int, int>> st{{0, 0}};  // (node, stage)
    while (!st.empty()) {
      const int n = st.back().first;
      const int stage = st.back().second++;
      if (stage == 0) {
        lo[n] = pos;
        const int c = t.left[n];
        if (c >= 0) st.push_back({c, 0});
        else put_leaf(pos++, t.leaf[~c]);
      } else if (stage == 1) {
        hi[n] = pos;
        const int c = t.right[n];
        if (c >= 0) st.push_back({c, 0});
        else put_leaf(pos++, t.leaf[~c]);
      } else {
        st.pop_back();
      }
    }
    uint32_t slot = 0, n_cat = 0;
    for (size_t i = 0; i < ni; ++i) {
      const int kind = node_kind(t, i);
      const uint32_t view = (uint32_t)view_ids.at({t.feat[i], kind});
      const uint32_t m = ((1u << hi[i]) - 1u) ^ ((1u << lo[i]) - 1u);
      if (kind == QV_CAT) {
        QsCatNode cn{};
        cn.view_dl = view | (((t.flags[i] & NF_DEFAULT_LEFT) ? 1u : 0u) << 16);
        cn.mm = m | (m << 16);
        cn.bits_begin = (uint32_t)pf.cat_bits.size();
        cn.bits_words = t.cat_words[i];
        pf.cat_bits.insert(pf.cat_bits.end(), f.cat_bits.begin() + t.cat_begin[i],
                           f.cat_bits.begin() + t.cat_begin[i] + t.cat_words[i]);
        pf.cat_nodes.push_back(cn);
        ++n_cat;
        continue;
      }
      const double *t0 = pf.thr.data() + pf.feats[t.feat[i]].thr_off;
      const double *t1 = t0 + pf.feats[t.feat[i]].thr_len;
      const uint32_t kbin = (uint32_t)(std::lower_bound(t0, t1, t.thr[i]) - t0);
      nd[slot] = kbin | (kbin << 16);
      nd[QS_SLOTS + slot] = m | (view << 24);
      ++slot;
    }
    nd[QS_SLOTS - 1] = cat_first | (n_cat << 24);
  }
  pf.ok = true;
  return pf;
}

uint32_t qs_stage_cap(const PackedForestQS &pf) {
  uint32_t longest = 1;
  for (const QsFeature &f : pf.feats)
    if (f.view_begin != f.view_end && f.thr_len <= 256u) longest = std::max<uint32_t>(longest, f.thr_len);
  return (longest + QS_STAGE_CHUNK - 1) / QS_STAGE_CHUNK * QS_STAGE_CHUNK;
}

QsSignature qs_signature(const PackedForestQS &pf, uint32_t thr_cap) {
  QsSignature sg;
  if (!pf.ok || thr_cap == 0 || thr_cap % QS_STAGE_CHUNK) return sg;
  sg.thr_cap = thr_cap;
  sg.n_views = (int)pf.views.size();
  uint32_t off = 0;
  for (const QsFeature &f : pf.feats) {
    const uint32_t chunks = (f.thr_len + QS_STAGE_CHUNK - 1) / QS_STAGE_CHUNK;
    // pack_forest_qs lays the tables out in column order, each padded to whole chunks: a column's offset follows from the
    // chunk counts before it (a different layout: no signature, the kernels keep reading descriptors)
    if (f.thr_off != off || f.view_end - f.view_begin > 6 || f.view_end < f.view_begin) return QsSignature{};
    const size_t fi = sg.cols.size();
    if (fi >= pf.rt_off.size() || fi >= pf.rt_len.size()) return QsSignature{};
    // (a column the forest never splits on is never searched: its compact table's place is not part of the key)
    const bool used = f.view_begin != f.view_end;
    const uint32_t rt_off = used ? pf.rt_off[fi] : 0u, rt_len = used ? pf.rt_len[fi] : 0u;
    sg.cols.push_back(QsSig{off, (uint16_t)chunks, f.view_begin, f.view_end, f.view_kinds, rt_off, rt_len});
    sg.text += "{" + std::to_string(off) + "u," + std::to_string(chunks) + "," + std::to_string(f.view_begin) + "," + std::to_string(f.view_end) + "," +
               std::to_string(f.view_kinds) + "u," + std::to_string(rt_off) + "u," + std::to_string(rt_len) + "u},";
    off += chunks * QS_STAGE_CHUNK;
  }
  if ((size_t)off != pf.thr.size()) return QsSignature{};
  sg.thr_total = off;
  sg.rt_total = (uint32_t)pf.thr_rt.size();
  sg.text = "/*cap " + std::to_string(thr_cap) + " views " + std::to_string(sg.n_views) + "*/" + sg.text;
  sg.ok = true;
  return sg;
}

}  // namespace mrk
