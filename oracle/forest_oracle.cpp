// TEST INFRASTRUCTURE — CPU oracle for the forest-scoring half of the /rank hot path.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// The product (metarank_amd/) never links, imports or executes anything under oracle/.
//
// What it restates: Booster.predictMat(values, rows, cols) as called from
//   /root/reference/src/main/scala/ai/metarank/ml/rank/LambdaMARTRanker.scala:348
// The arithmetic itself lives in third-party natives that are NOT vendored in the reference
// (build.sbt:57-58: io.github.metarank:ltrlib_2.13:0.2.6 -> ml.dmlc:xgboost4j, and
// io.github.metarank:lightgbm4j:4.6.0-1 = LightGBM 4.6.0), so the two evaluators below follow
// the published algorithms of those libraries:
//   LightGBM 4.6  include/LightGBM/tree.h  Tree::Predict / GetLeaf / NumericalDecision /
//                 CategoricalDecision, src/boosting/gbdt_prediction.cpp GBDT::PredictRaw,
//                 src/c_api.cpp RowFunctionFromDenseMatric (|x| <= 1e-35 cells are dropped)
//   XGBoost       src/predictor/cpu_predictor.cc PredictByAllTrees / GetLeafIndex,
//                 include/xgboost/tree_model.h RegTree::GetNext, src/common/categorical.h Decision,
//                 src/data/data.cc (inf rejected when `missing` is NaN)
//
// PARITY UNPINNED for forest scores: the reference's own tests never assert a score
// (SURVEY.md F7, §8c); the model fixtures are Git-LFS pointers.  This file is pinned only
// against (i) hand-computed vectors in tests/test_oracle_forest.py and (ii) scikit-learn's
// independent tree evaluator on forests exported into both on-disk formats
// (tests/golden/make_sklearn_golden.py).
//
// Deliberately simple: one row at a time, pointer chasing exactly like the libraries do.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct LgbmTree {
  int num_leaves = 1;
  std::vector<int> split_feature;
  std::vector<double> threshold;
  std::vector<int> decision_type;
  std::vector<int> left_child, right_child;
  std::vector<double> leaf_value;
  std::vector<int> cat_boundaries;
  std::vector<uint32_t> cat_threshold;
};

struct XgbTree {
  std::vector<int> left, right, split_index, default_left, split_type;
  std::vector<float> split_cond;                 // leaf value for leaves
  std::vector<std::vector<int>> categories;      // per node, the categories that go RIGHT
};

struct Forest {
  int backend = 0;  // 0 lightgbm, 1 xgboost
  float base_score = 0.5f;
  std::vector<LgbmTree> lgbm;
  std::vector<XgbTree> xgb;
};

// ---- LightGBM -------------------------------------------------------------------------------
const double kZeroThreshold = 1e-35f;  // include/LightGBM/meta.h
inline bool IsZero(double fval) { return fval >= -kZeroThreshold && fval <= kZeroThreshold; }

inline bool FindInBitset(const uint32_t *bits, int n, int pos) {
  int i1 = pos / 32;
  if (i1 >= n) return false;
  int i2 = pos % 32;
  return (bits[i1] >> i2) & 1;
}

int lgbm_numerical_decision(const LgbmTree &t, double fval, int node) {
  int dt = t.decision_type[node];
  int missing_type = (dt >> 2) & 3;  // 0 None, 1 Zero, 2 NaN
  if (std::isnan(fval) && missing_type != 2) fval = 0.0;
  if ((missing_type == 1 && IsZero(fval)) || (missing_type == 2 && std::isnan(fval))) {
    if (dt & 2) return t.left_child[node];  // kDefaultLeftMask
    return t.right_child[node];
  }
  if (fval <= t.threshold[node]) return t.left_child[node];
  return t.right_child[node];
}

int lgbm_categorical_decision(const LgbmTree &t, double fval, int node) {
  int int_fval;
  if (std::isnan(fval)) return t.right_child[node];
  // static_cast<int>(double) is undefined out of range; x86 yields INT_MIN (negative -> right).
  // Saturate explicitly so every platform agrees: > INT_MAX never is a member of a bitset.
  if (fval >= 2147483648.0) return t.right_child[node];
  if (fval <= -2147483649.0) return t.right_child[node];
  int_fval = static_cast<int>(fval);
  if (int_fval < 0) return t.right_child[node];
  int cat_idx = static_cast<int>(t.threshold[node]);
  if (FindInBitset(t.cat_threshold.data() + t.cat_boundaries[cat_idx],
                   t.cat_boundaries[cat_idx + 1] - t.cat_boundaries[cat_idx], int_fval))
    return t.left_child[node];
  return t.right_child[node];
}

double lgbm_tree_predict(const LgbmTree &t, const double *row) {
  if (t.num_leaves <= 1) return t.leaf_value[0];
  int node = 0;
  while (node >= 0) {
    double fval = row[t.split_feature[node]];
    if (t.decision_type[node] & 1) node = lgbm_categorical_decision(t, fval, node);
    else node = lgbm_numerical_decision(t, fval, node);
  }
  return t.leaf_value[~node];
}

// ---- XGBoost --------------------------------------------------------------------------------
bool xgb_cat_goes_left(const std::vector<int> &cats_right, float fvalue) {
  // common::Decision: invalid categories and categories beyond the stored set go left; members of
  // the set go right.
  if (fvalue < 0 || fvalue >= 16777216.0f) return true;
  int cat = static_cast<int>(fvalue);
  for (int c : cats_right)
    if (c == cat) return false;
  return true;
}

float xgb_tree_predict(const XgbTree &t, const float *row) {
  int nid = 0;
  while (t.left[nid] != -1) {
    float fvalue = row[t.split_index[nid]];
    bool go_left;
    if (std::isnan(fvalue)) go_left = t.default_left[nid] != 0;
    else if (t.split_type[nid] == 1) go_left = xgb_cat_goes_left(t.categories[nid], fvalue);
    else go_left = fvalue < t.split_cond[nid];
    nid = go_left ? t.left[nid] : t.right[nid];
  }
  return t.split_cond[nid];
}

}  // namespace

extern "C" {

void *orc_forest_new(int backend, float base_score) {
  Forest *f = new Forest();
  f->backend = backend;
  f->base_score = base_score;
  return f;
}
void orc_forest_free(void *h) { delete (Forest *)h; }
int orc_forest_num_trees(void *h) {
  Forest *f = (Forest *)h;
  return (int)(f->backend == 0 ? f->lgbm.size() : f->xgb.size());
}

void orc_forest_add_lgbm_tree(void *h, int num_leaves, const int *split_feature, const double *threshold,
                              const int *decision_type, const int *left_child, const int *right_child,
                              const double *leaf_value, int n_cat_boundaries, const int *cat_boundaries,
                              int n_cat_threshold, const uint32_t *cat_threshold) {
  Forest *f = (Forest *)h;
  LgbmTree t;
  t.num_leaves = num_leaves;
  int nn = num_leaves - 1;
  if (nn > 0) {
    t.split_feature.assign(split_feature, split_feature + nn);
    t.threshold.assign(threshold, threshold + nn);
    t.decision_type.assign(decision_type, decision_type + nn);
    t.left_child.assign(left_child, left_child + nn);
    t.right_child.assign(right_child, right_child + nn);
  }
  t.leaf_value.assign(leaf_value, leaf_value + num_leaves);
  if (n_cat_boundaries > 0) t.cat_boundaries.assign(cat_boundaries, cat_boundaries + n_cat_boundaries);
  if (n_cat_threshold > 0) t.cat_threshold.assign(cat_threshold, cat_threshold + n_cat_threshold);
  f->lgbm.push_back(std::move(t));
}

// categories: concatenated per-node category lists, cat_offsets[n_nodes+1]
void orc_forest_add_xgb_tree(void *h, int n_nodes, const int *left, const int *right, const int *split_index,
                             const float *split_cond, const int *default_left, const int *split_type,
                             const int *cat_offsets, const int *categories) {
  Forest *f = (Forest *)h;
  XgbTree t;
  t.left.assign(left, left + n_nodes);
  t.right.assign(right, right + n_nodes);
  t.split_index.assign(split_index, split_index + n_nodes);
  t.split_cond.assign(split_cond, split_cond + n_nodes);
  t.default_left.assign(default_left, default_left + n_nodes);
  t.split_type.assign(split_type, split_type + n_nodes);
  t.categories.resize(n_nodes);
  for (int i = 0; i < n_nodes; ++i)
    t.categories[i].assign(categories + cat_offsets[i], categories + cat_offsets[i + 1]);
  f->xgb.push_back(std::move(t));
}

// returns 0, or 1 when an XGBoost forest meets +-inf after the Double->Float narrowing (the
// reference raises "Input data contains `inf`"); out is still filled for the finite rows.
int orc_forest_predict(void *h, const double *X, int rows, int cols, double *out) {
  Forest *f = (Forest *)h;
  int status = 0;
  if (f->backend == 0) {
    std::vector<double> buf(cols);
    for (int r = 0; r < rows; ++r) {
      // c_api.cpp RowFunctionFromDenseMatric + Predictor::CopyToPredictBuffer: cells with
      // |x| <= kZeroThreshold (and not NaN) are not copied, the buffer holds 0.0 for them.
      for (int c = 0; c < cols; ++c) {
        double x = X[(size_t)r * cols + c];
        buf[c] = (std::fabs(x) > kZeroThreshold || std::isnan(x)) ? x : 0.0;
      }
      double sum = 0.0;  // GBDT::PredictRaw: output[k] += tree->Predict(features), iteration order
      for (const LgbmTree &t : f->lgbm) sum += lgbm_tree_predict(t, buf.data());
      out[r] = sum;
    }
  } else {
    std::vector<float> buf(cols);
    for (int r = 0; r < rows; ++r) {
      for (int c = 0; c < cols; ++c) {
        float v = (float)X[(size_t)r * cols + c];  // ltrlib: values.map(_.toFloat), missing = NaN
        if (std::isinf(v)) status = 1;
        buf[c] = v;
      }
      float psum = f->base_score;  // predictions start from the base margin
      for (const XgbTree &t : f->xgb) psum += xgb_tree_predict(t, buf.data());
      out[r] = (double)psum;
    }
  }
  return status;
}

// Per-tree leaf values for one row (debug aid for parity failures).
void orc_forest_leaves(void *h, const double *row, int cols, double *out_per_tree) {
  Forest *f = (Forest *)h;
  if (f->backend == 0) {
    std::vector<double> buf(row, row + cols);
    for (auto &x : buf) x = (std::fabs(x) > kZeroThreshold || std::isnan(x)) ? x : 0.0;
    for (size_t i = 0; i < f->lgbm.size(); ++i) out_per_tree[i] = lgbm_tree_predict(f->lgbm[i], buf.data());
  } else {
    std::vector<float> buf(cols);
    for (int c = 0; c < cols; ++c) buf[c] = (float)row[c];
    for (size_t i = 0; i < f->xgb.size(); ++i) out_per_tree[i] = xgb_tree_predict(f->xgb[i], buf.data());
  }
}

}  // extern "C"
