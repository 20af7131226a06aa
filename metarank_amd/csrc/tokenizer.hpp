// WordPiece tokenizer driven by a HuggingFace `tokenizer.json`.
// Reference boundary: OnnxSession.load builds `HuggingFaceTokenizer.newInstance(tok, {padding: true, truncation: true})`
// (ml/onnx/sbert/OnnxSession.scala:42-43) and the encoders call `tokenizer.batchEncode` on single strings
// (OnnxBiEncoder.scala:14) or on (query, item text) pairs (OnnxCrossEncoder.scala:27).  DJL's HuggingFaceTokenizer is
// a JNI wrapper of the HuggingFace `tokenizers` library; the pipeline restated here is that library's, for the
// BERT family the reference ships (BertNormalizer -> BertPreTokenizer -> WordPiece -> BertProcessing |
// TemplateProcessing -> LongestFirst truncation -> BatchLongest padding).  Anything else in the JSON is rejected
// at load time (MRK_ERR_UNSUPPORTED) rather than approximated.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace mrk {

struct Encoding {
  std::vector<int32_t> ids, type_ids, mask;
};

class Tokenizer {
 public:
  // throws StatusError(MRK_ERR_PARSE | MRK_ERR_UNSUPPORTED)
  static Tokenizer from_json(const char *json, size_t len);

  // batchEncode: every row padded to the longest row of the batch; `b == nullptr` for single sequences.
  // Returns the padded length; rows are appended to `out` (n encodings).
  int encode_batch(const char *const *a, const char *const *b, int n, std::vector<Encoding> &out) const;

  // one sequence (or pair), truncated and with special tokens, no padding
  Encoding encode(const std::string &a, const std::string *b) const;
  // wordpiece ids of one text without special tokens / truncation (what the cross-encoder column stores per item)
  std::vector<int32_t> pieces(const std::string &text) const;
  // [CLS] A [SEP] (B [SEP]) from already-tokenised pieces, with the same truncation as encode()
  Encoding assemble(std::vector<int32_t> a, const std::vector<int32_t> *b) const;

  int max_length() const { return max_length_; }
  int pad_id() const { return pad_id_; }
  int vocab_size() const { return (int)vocab_size_; }

 private:
  struct Piece { bool special; int32_t id; int type; int seq; };  // template element: special token or $A/$B
  std::unordered_map<std::string, int32_t> vocab_;
  std::vector<std::pair<std::string, int32_t>> added_;  // special / added tokens matched on the raw text
  size_t vocab_size_ = 0;
  std::string prefix_ = "##";
  int32_t unk_id_ = 0;
  int max_chars_ = 100;
  bool clean_text_ = true, chinese_ = true, strip_accents_ = true, lowercase_ = true;
  std::vector<Piece> single_, pair_;
  int max_length_ = 512;
  int32_t pad_id_ = 0, pad_type_ = 0;

  void normalize(const std::vector<uint32_t> &in, std::vector<uint32_t> &out) const;
  void wordpiece(const std::vector<uint32_t> &word, std::vector<int32_t> &out) const;
  void tokenize_plain(const std::vector<uint32_t> &cps, std::vector<int32_t> &out) const;
};

}  // namespace mrk
