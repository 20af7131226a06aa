// Forest IR + model readers + device packing for the scorer.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "device_types.hpp"

namespace mrk {

enum class Backend : int { LightGBM = 0, XGBoost = 1 };

// a well-formed model that uses something the scorer does not implement (dart, multi-output, vector leaves, a non-identity
// objective ...): MRK_ERR_UNSUPPORTED at the C ABI, never a silently different score; malformed input is std::runtime_error
// (MRK_ERR_PARSE)
struct UnsupportedModel : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// decision flags of one internal node (already normalised across both libraries)
enum : uint8_t {
  NF_CATEGORICAL = 1,   // categorical split, `cat_begin/cat_words` address the bitset
  NF_DEFAULT_LEFT = 2,  // where "missing" goes
  NF_MISS_ZERO = 4,     // LightGBM MissingType::Zero  (|v| <= 1e-35 counts as missing)
  NF_MISS_NAN = 8,      // LightGBM MissingType::NaN / XGBoost (NaN is missing)
};

struct Tree {
  // internal nodes; child >= 0 is an internal node index, child < 0 is ~leaf_index
  std::vector<int32_t> feat;
  std::vector<double> thr;  // f64 for LightGBM, exact widening of the f32 condition for XGBoost
  std::vector<uint8_t> flags;
  std::vector<int32_t> left, right;
  std::vector<uint32_t> cat_begin, cat_words;  // into Forest::cat_bits (only for NF_CATEGORICAL)
  std::vector<double> leaf;                    // f64 (LightGBM) or exact widening of f32 (XGBoost)
  std::vector<float> gain;                     // per internal node: LightGBM split_gain / XGBoost loss_chg; empty when the file has none
  int depth = 0;                               // longest root->leaf path, in internal-node visits
};

struct Forest {
  Backend backend = Backend::LightGBM;
  int n_features = 0;
  double base_score = 0.0;  // XGBoost margin-space base score; 0 for LightGBM
  bool average_output = false;
  std::vector<Tree> trees;
  std::vector<uint32_t> cat_bits;  // bitset words: bit c of word w <=> category 32*w+c is in the set
  std::string objective;

  // Booster.weights(): per-feature importance, `n` >= n_features entries (the rest 0).  type: 0 split count, 1 gain, 2 total gain
  // - each with the library's own arithmetic (forest.cpp); throws std::runtime_error when gains are asked for and the file has none
  void feature_importance(int type, double *out, int n) const;
  int64_t n_nodes() const;
  int64_t n_leaves() const;
  int max_depth() const;
  int64_t n_categorical() const;
};

// --- readers (throw std::runtime_error on malformed input) -----------------------------------
Forest parse_lightgbm_text(const char *text, size_t len);
Forest parse_xgboost(const uint8_t *bytes, size_t len);  // JSON or UBJSON, auto-detected

struct Container {
  int version = 0;
  std::vector<std::string> features;
  int booster_tag = 0;
  const uint8_t *inner = nullptr;
  size_t inner_len = 0;
  int n_warmup = 0;                // v3: number of RankingEventFormat records that follow
  const uint8_t *warmup = nullptr; // their bytes (to the end of the blob)
  size_t warmup_len = 0;
};
// Metarank bitstream v2/v3 (reference: ml/rank/LambdaMARTRanker.scala:192-236,367-389)
Container parse_container(const uint8_t *blob, size_t len);

// --- device image --------------------------------------------------------------------------
// The forest is cut into chunks of consecutive trees; each chunk is one contiguous byte image
// that a workgroup copies into LDS verbatim.  Inside a chunk every tree is
//   [internal nodes][leaves]  (f64 model: 16 B nodes + 8 B leaves, f32 model: 8 B + 4 B)
// padded to 16 B.  A tree never straddles chunks.
struct PackedNode64 {  // LightGBM
  double thr;          // numerical: threshold; categorical: lo32 = cat_begin, hi32 = cat_words
  uint16_t feat;
  uint8_t flags;
  uint8_t nan_left;    // precomputed: where NaN goes (1 = left)
  int16_t left, right; // >=0 internal node, <0 ~leaf
};
static_assert(sizeof(PackedNode64) == 16, "node64 must be 16 bytes");

struct PackedNode32 {  // XGBoost
  float thr;           // numerical: split condition; categorical: bits = cat_begin | cat_words<<24
  uint16_t feat;
  uint8_t flags;
  uint8_t pad;
  int16_t left, right;
  uint32_t pad2;
};
static_assert(sizeof(PackedNode32) == 16, "node32 must be 16 bytes");

struct TreeRef {       // per tree, lives in the chunk header table
  uint32_t node_off;   // byte offset of the node array inside the chunk image
  uint32_t leaf_off;   // byte offset of the leaf array inside the chunk image
  uint16_t n_nodes;    // 0 => single-leaf tree
  uint16_t depth;
};

struct ChunkRef {
  uint32_t byte_off;    // offset of the chunk image in the packed buffer (16 B aligned)
  uint32_t byte_len;    // multiple of 16
  uint32_t first_tree;
  uint32_t n_trees;
};

struct PackedForest {
  std::vector<uint8_t> image;     // all chunk images back to back
  std::vector<TreeRef> trees;     // n_trees
  std::vector<ChunkRef> chunks;
  uint32_t max_chunk_bytes = 0;
  uint32_t max_chunk_trees = 0;
};

PackedForest pack_forest(const Forest &f, uint32_t chunk_bytes);

// --- bit-vector device image (scorer "qs") ---------------------------------------------------
// For forests of small trees (<= 16 leaves: LightGBM's numLeaves default in Metarank,
// config/BoosterConfig.scala:19-28) the scorer does not walk trees.  It evaluates EVERY internal
// node of a tree with data that is uniform across the wavefront (thresholds and masks live in
// scalar registers) and finds the exit leaf with the QuickScorer bit-vector rule
// (Lucchese et al., SIGIR'15): leaves are numbered left to right, a node whose test is false
// removes the leaves of its left subtree, and the exit leaf is the lowest leaf left.
//
// Binning makes every test one 16-bit integer compare.  With T_f the sorted distinct thresholds
// the forest uses on feature f and bin(x) = #{t in T_f : t < x} (LightGBM, `x <= t` goes left) or
// #{t : t <= x} (XGBoost, `x < t` goes left):   x (<=|<) T_f[k]  <=>  bin(x) <= k,  exactly.
// Missing values are folded into the cell by giving a feature up to four "views" (columns of the
// binned tile), one per way a node can treat a missing value; a node reads the view that matches
// its own rule, so the kernel has no special cases:
//   QV_NAN_RIGHT   NaN -> 0x7FFF (greater than every k: right)   else bin
//   QV_NAN_LEFT    NaN -> 0      (<= every k: left)              else bin
//   QV_NAN_ZERO    NaN -> bin(0.0)  (LightGBM MissingType::None compares NaN as 0.0: one view serves the
//                  nodes on both sides of zero)
//   QV_MISS_RIGHT  NaN or 0.0 -> 0x7FFF   (LightGBM MissingType::Zero, default right)
//   QV_MISS_LEFT   NaN or 0.0 -> 0        (LightGBM MissingType::Zero, default left)
//   QV_CAT         the category id itself (one view per categorical column): 0..0x7FFC, 0x7FFD = a valid
//                  category beyond every bitset, 0x7FFE = invalid (XGBoost: negative or >= 2^24),
//                  0x7FFF = NaN (and, for LightGBM, negative)
// Numerical node (2 dwords, read with scalar loads): {k | k << 16,  m | view << 24} where m (16 bits)
// has bit p set for every leaf position p in the node's left subtree and the view index sits in
// bits 24-31 (so that `word >> 16` is the view's byte offset in a tile of 128 rows).  A tree is two
// arrays of QS_SLOTS dwords (all k words, all m/view words) and arrives in two s_load_dwordx16;
// unused slots have m = 0 (no effect).  The last slot is never a node: its k word holds
// `first categorical node | count << 24`.  Categorical nodes are rare (about one per
// ten trees); they live in a side list {view | default_left << 16, m | m << 16, bitset begin, bitset
// words} and are tested against their bitset one by one.  One all-zero tree is appended so that the
// scorer's prefetch of "the next tree" never leaves the array.
// Leaves are stored in left-to-right position order, QS_LEAVES per tree.
// (constants and device structs of this format: device_types.hpp)

struct PackedForestQS {
  bool ok = false;   // false: the forest does not fit this format; the tree-walk kernel is used
  std::string why;
  bool f64 = true;
  int n_trees = 0;
  std::vector<uint32_t> nodes;   // (n_trees + 1) * QS_SLOTS * 2
  std::vector<uint8_t> leaves;   // n_trees * QS_LEAVES * (8 | 4)
  std::vector<double> thr;
  // the same tables once more, COMPACT: back to back at their exact lengths (no 128-entry chunks, no padding) - a model with
  // 50 distinct thresholds per column is 9 KB here and 24 KB in `thr`; what the resident-table sinks keep in LDS.
  // Columns with more than 256 thresholds take no room (rt_len = QS_RT_NONE).
  std::vector<double> thr_rt;
  std::vector<uint32_t> rt_off, rt_len;   // per feature
  std::vector<QsFeature> feats;  // n_features
  std::vector<QsView> views;
  std::vector<QsCatNode> cat_nodes;
  std::vector<uint32_t> cat_bits;
  size_t device_bytes() const {
    return nodes.size() * 4 + leaves.size() + (thr.size() + thr_rt.size()) * 8 + feats.size() * 16 + views.size() * 4 + cat_nodes.size() * 16 + cat_bits.size() * 4;
  }
};

PackedForestQS pack_forest_qs(const Forest &f, int n_cols);

// The forest's VIEW SIGNATURE: per matrix column its views (which tile columns, of which kinds) and the number of
// 128-entry chunks its threshold table takes - what decides the CODE of a kernel that bins values for this forest, as
// opposed to the data it reads (thresholds, bin(0.0), the trees).  Forests retrained on the same features usually keep it.
// The run-time specialised assembly kernels are keyed by it (jit.cpp; device side: rank_device.hpp CellSink<F64, QS>).
struct QsSignature {
  bool ok = false;             // false: the image does not have the layout the signature assumes - kernels read descriptors
  uint32_t thr_cap = 0;        // doubles per LDS staging buffer (QsDev::thr_cap)
  int n_views = 0;
  uint32_t thr_total = 0;      // doubles of all threshold tables together (whole chunks per column)
  uint32_t rt_total = 0;       // doubles of the compact tables (PackedForestQS::thr_rt)
  std::vector<QsSig> cols;     // n_features
  std::string text;            // the rows as a C++ initialiser list: part of the specialised translation unit, and the key
};
// doubles per LDS staging buffer of the assembly kernels: the longest table they stage (<= 256 entries), in whole chunks
uint32_t qs_stage_cap(const PackedForestQS &pf);
QsSignature qs_signature(const PackedForestQS &pf, uint32_t thr_cap);

}  // namespace mrk
