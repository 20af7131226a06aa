// BERT-family text encoder on the device: the graph behind OnnxBiEncoder / OnnxCrossEncoder
// (ml/onnx/sbert/OnnxBiEncoder.scala:13-60, OnnxCrossEncoder.scala:22-51) with the ONNX runtime replaced by
// hand-written gfx950 kernels (encoder.hip).  Weights are read from the file the reference loads
// (`pytorch_model.onnx`, OnnxSession.scala:29-35) or from the same checkpoint as safetensors (weights.cpp).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "runtime.hpp"
#include "tokenizer.hpp"

namespace mrk {

// host copy of one checkpoint tensor, converted to f32
struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  int64_t numel() const { int64_t n = 1; for (int64_t d : shape) n *= d; return n; }
};

// HuggingFace BertModel state_dict names (no "bert." / wrapper prefix), linear weights as [out, in]
struct Checkpoint {
  std::map<std::string, HostTensor> tensors;
  int heads = 0;  // 0 = unknown
};

// weights.cpp: content-sniffing reader (safetensors header | ONNX ModelProto)
Checkpoint read_checkpoint(const uint8_t *bytes, size_t len);

struct EncoderShape {
  int layers = 0, hidden = 0, heads = 0, inter = 0, vocab = 0, max_pos = 0, type_vocab = 0;
  bool classifier = false;
  float eps = 1e-12f;
};

struct LayerDev {
  const uint16_t *wqkv, *wo, *w1, *w2;           // fp16 [out, in]
  const float *bqkv, *bo, *b1, *b2, *ln1g, *ln1b, *ln2g, *ln2b;
};

// the f32 copies of the matrices an encoder loaded with precision f32 keeps (mrk_encoder_load_ex): every product then runs
// on f32 operands with f32 accumulation on the f32-input matrix instruction - the arithmetic of the reference's fp32 ONNX graph
struct LayerDev32 {
  const float *wqkv, *wo, *w1, *w2;              // f32 [out, in]
};

struct EncoderDev {
  EncoderShape shape;
  const uint16_t *word, *pos, *type;             // fp16 embedding tables
  const float *embg, *embb;
  std::vector<LayerDev> layers;
  const uint16_t *pool_w = nullptr;              // fp16 [H, H]
  const float *pool_b = nullptr, *cls_w = nullptr, *cls_b = nullptr;
  bool f32 = false;                              // precision f32: the fields below are set and the forward pass uses them
  const float *word32 = nullptr, *pos32 = nullptr, *type32 = nullptr, *pool_w32 = nullptr;
  std::vector<LayerDev32> layers32;
};

// activations of one forward call (grow-only, owned by the encoder handle)
struct EncoderScratch {
  DevBuf ids, x, xh, qkv, ctx, mid, y, out;
};

// encoder.hip ------------------------------------------------------------------------------------
// d_ids: 3 x n x seq int32 (ids | type_ids | mask) already on the device.  Leaves the final hidden states in
// scratch.x (f32 [n*seq, H]).
// sizes the activation buffers for n x seq tokens (never called while a graph is being captured)
void encoder_reserve(const EncoderDev &enc, EncoderScratch &s, int n, int seq);
void encoder_forward(const EncoderDev &enc, EncoderScratch &s, int n, int seq, hipStream_t stream);
// the same over PACKED tokens: d_ids = [ids | type_ids | position ids] (3 x M) then cu (n + 1 offsets); the sequences lie
// back to back without padding, max_len = the longest one.  Leaves the hidden states packed in scratch.x (f32 [M, H]).
void encoder_forward_packed(const EncoderDev &enc, EncoderScratch &s, int n, int max_len, int M, hipStream_t stream);
// OnnxBiEncoder.avgpool: scratch.x -> out f32 [n, H] (device); M_packed > 0: after encoder_forward_packed
void encoder_meanpool(const EncoderDev &enc, EncoderScratch &s, int n, int seq, int M_packed, float *d_out, hipStream_t stream);
// pooler + classifier on the [CLS] row: -> out f32 [n] (device)
void encoder_classify(const EncoderDev &enc, EncoderScratch &s, int n, int seq, int M_packed, float *d_out, hipStream_t stream);

// capi_encoder.cpp / features.cpp
void encoder_retain(mrk_encoder *e);
void encoder_release(mrk_encoder *e);
// EmbeddingCache semantics: embeddings of `texts` (cached by text; misses run through the device in one batch)
void encoder_embed_cached(mrk_encoder *e, const std::vector<std::string> &texts, std::vector<std::vector<float>> &out);
// logits of already assembled ([CLS] a [SEP] b [SEP]) rows; padded per chunk, any number of rows
void encoder_score_rows(mrk_encoder *e, const std::vector<Encoding> &rows, float *out);
void bind_encoder(mrk_ctx *ctx, const char *feature, mrk_encoder *enc);

}  // namespace mrk

struct mrk_encoder {
  mrk_ctx *ctx = nullptr;
  mrk::Tokenizer tok;
  mrk::EncoderDev dev;
  mrk::DevBuf weights;     // one allocation: fp16 matrices then f32 vectors
  mrk::DevBuf weights32;   // precision f32: the matrices once more, as f32
  mrk::EncoderScratch scratch;
  mrk::PinBuf h_ids, h_out;  // pinned staging of one call
  hipStream_t stream = nullptr;
  std::map<std::tuple<int, int, int>, hipGraphExec_t> graphs;  // (n, seq, mode) -> recorded forward pass
  std::mutex mu;           // one forward at a time per handle (scratch is shared)
  int64_t device_bytes = 0;
  // EmbeddingCache: query text -> embedding (FieldMatchBiencoderFeature.scala:96-99)
  std::map<std::string, std::vector<float>> cache;
  int precision = 0;       // MRK_ENCODER_FP16 | _F32 (mrk_encoder_load_ex; _AUTO = _F32)
  std::atomic<int> refs{1};
};
