// Forest IR + model readers + device packing for the scorer.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mrk {

enum class Backend : int { LightGBM = 0, XGBoost = 1 };

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

}  // namespace mrk
