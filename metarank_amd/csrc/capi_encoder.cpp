// C ABI of the text-encoder leg (include/mrk.h "text encoders"): tokenizer handles (host only) and encoder
// handles (BERT-family graph on the device, encoder.hip).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <tuple>

#include "encoder.hpp"
#include "runtime.hpp"
#include "tokenizer.hpp"

using namespace mrk;

struct mrk_tokenizer {
  Tokenizer tok;
};

namespace {
template <typename F>
int guard(F &&f) {
  try {
    f();
    return MRK_OK;
  } catch (const StatusError &e) {
    set_last_error(e.what());
    return e.status;
  } catch (const std::bad_alloc &) {
    set_last_error("out of host memory");
    return MRK_ERR_DEVICE;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return MRK_ERR_PARSE;
  }
}

void need(bool ok, const char *what) {
  if (!ok) throw StatusError(MRK_ERR_INVALID_ARG, what);
}

void flatten(const std::vector<Encoding> &rows, int len, int32_t *ids, int32_t *types, int32_t *mask);

enum { MODE_HIDDEN, MODE_POOL, MODE_LOGIT };

uint16_t to_half(float f) {
  const _Float16 h = (_Float16)f;  // round to nearest even, as a checkpoint saved in fp16 would be
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}

// Validates the checkpoint against the BertModel layout, converts and uploads it (one device allocation).
void build_encoder(mrk_encoder &e, const Checkpoint &ck, bool f32) {
  auto get = [&](const std::string &name) -> const HostTensor & {
    auto it = ck.tensors.find(name);
    if (it == ck.tensors.end()) throw StatusError(MRK_ERR_PARSE, "encoder weights: tensor " + name + " is missing (BERT-family graphs only)");
    return it->second;
  };
  auto shape2 = [&](const HostTensor &t, int64_t a, int64_t b, const std::string &name) {
    if (t.shape.size() != 2 || t.shape[0] != a || t.shape[1] != b)
      throw StatusError(MRK_ERR_PARSE, "encoder weights: " + name + " has an unexpected shape");
  };
  EncoderShape sh;
  const HostTensor &word = get("embeddings.word_embeddings.weight");
  if (word.shape.size() != 2) throw StatusError(MRK_ERR_PARSE, "encoder weights: word embeddings are not a matrix");
  sh.vocab = (int)word.shape[0];
  sh.hidden = (int)word.shape[1];
  sh.max_pos = (int)get("embeddings.position_embeddings.weight").shape.at(0);
  sh.type_vocab = (int)get("embeddings.token_type_embeddings.weight").shape.at(0);
  while (ck.tensors.count("encoder.layer." + std::to_string(sh.layers) + ".attention.self.query.weight")) ++sh.layers;
  if (!sh.layers) throw StatusError(MRK_ERR_PARSE, "encoder weights: no encoder.layer.N tensors found (fused / non-BERT graph?)");
  sh.inter = (int)get("encoder.layer.0.intermediate.dense.weight").shape.at(0);
  sh.heads = ck.heads;
  sh.classifier = ck.tensors.count("pooler.dense.weight") && ck.tensors.count("classifier.weight");
  if (sh.classifier && get("classifier.weight").numel() != sh.hidden)
    throw StatusError(MRK_ERR_UNSUPPORTED, "encoder weights: classifier heads with more than one logit are not supported");
  if (sh.heads <= 0) throw StatusError(MRK_ERR_INVALID_ARG, "encoder: number of attention heads unknown - pass `heads`");
  const int dh = sh.hidden % sh.heads == 0 ? sh.hidden / sh.heads : 0;
  if ((dh != 32 && dh != 64) || sh.hidden % 64 || sh.inter % 64 || sh.hidden > 1024)
    throw StatusError(MRK_ERR_UNSUPPORTED, "encoder: hidden " + std::to_string(sh.hidden) + " / heads " + std::to_string(sh.heads) + " / intermediate " +
                      std::to_string(sh.inter) + " is outside the supported shapes (head size 32 or 64, widths multiple of 64, hidden <= 1024)");
  const int H = sh.hidden, I = sh.inter;

  // host image: fp16 block then f32 block, every tensor 256-byte aligned
  std::vector<uint16_t> hs;
  std::vector<float> fs;
  auto put_h = [&](const HostTensor &t) { size_t off = hs.size(); for (float v : t.data) hs.push_back(to_half(v)); hs.resize((hs.size() + 127) / 128 * 128); return off; };
  auto put_f = [&](const HostTensor &t) { size_t off = fs.size(); fs.insert(fs.end(), t.data.begin(), t.data.end()); fs.resize((fs.size() + 63) / 64 * 64); return off; };
  struct LayerOff { size_t wqkv, wo, w1, w2, bqkv, bo, b1, b2, g1, be1, g2, be2; };
  std::vector<LayerOff> lo(sh.layers);
  const size_t o_word = put_h(word), o_pos = put_h(get("embeddings.position_embeddings.weight")), o_type = put_h(get("embeddings.token_type_embeddings.weight"));
  const size_t o_eg = put_f(get("embeddings.LayerNorm.weight")), o_eb = put_f(get("embeddings.LayerNorm.bias"));
  for (int l = 0; l < sh.layers; ++l) {
    const std::string p = "encoder.layer." + std::to_string(l) + ".";
    HostTensor wqkv, bqkv;
    for (const char *nm : {"query", "key", "value"}) {
      const HostTensor &w = get(p + "attention.self." + nm + ".weight"), &b = get(p + "attention.self." + nm + ".bias");
      shape2(w, H, H, p + nm);
      wqkv.data.insert(wqkv.data.end(), w.data.begin(), w.data.end());
      bqkv.data.insert(bqkv.data.end(), b.data.begin(), b.data.end());
    }
    shape2(get(p + "attention.output.dense.weight"), H, H, p + "attention.output.dense.weight");
    shape2(get(p + "intermediate.dense.weight"), I, H, p + "intermediate.dense.weight");
    shape2(get(p + "output.dense.weight"), H, I, p + "output.dense.weight");
    lo[l] = LayerOff{put_h(wqkv), put_h(get(p + "attention.output.dense.weight")), put_h(get(p + "intermediate.dense.weight")),
                     put_h(get(p + "output.dense.weight")), put_f(bqkv), put_f(get(p + "attention.output.dense.bias")),
                     put_f(get(p + "intermediate.dense.bias")), put_f(get(p + "output.dense.bias")),
                     put_f(get(p + "attention.output.LayerNorm.weight")), put_f(get(p + "attention.output.LayerNorm.bias")),
                     put_f(get(p + "output.LayerNorm.weight")), put_f(get(p + "output.LayerNorm.bias"))};
  }
  size_t o_pw = 0, o_pb = 0, o_cw = 0, o_cb = 0;
  if (sh.classifier) {
    shape2(get("pooler.dense.weight"), H, H, "pooler.dense.weight");
    o_pw = put_h(get("pooler.dense.weight"));
    o_pb = put_f(get("pooler.dense.bias"));
    o_cw = put_f(get("classifier.weight"));
    o_cb = put_f(get("classifier.bias"));
  }
  const size_t hbytes = hs.size() * 2, fbytes = fs.size() * 4;
  e.weights.reserve(hbytes + fbytes);
  MRK_HIP(hipMemcpy(e.weights.p, hs.data(), hbytes, hipMemcpyHostToDevice));
  MRK_HIP(hipMemcpy((char *)e.weights.p + hbytes, fs.data(), fbytes, hipMemcpyHostToDevice));
  e.device_bytes = (int64_t)(hbytes + fbytes);
  const uint16_t *hb = e.weights.as<uint16_t>();
  const float *fb = (const float *)((char *)e.weights.p + hbytes);
  EncoderDev &d = e.dev;
  d.shape = sh;
  d.word = hb + o_word; d.pos = hb + o_pos; d.type = hb + o_type;
  d.embg = fb + o_eg; d.embb = fb + o_eb;
  for (auto &o : lo)
    d.layers.push_back(LayerDev{hb + o.wqkv, hb + o.wo, hb + o.w1, hb + o.w2, fb + o.bqkv, fb + o.bo, fb + o.b1, fb + o.b2,
                                fb + o.g1, fb + o.be1, fb + o.g2, fb + o.be2});
  if (sh.classifier) { d.pool_w = hb + o_pw; d.pool_b = fb + o_pb; d.cls_w = fb + o_cw; d.cls_b = fb + o_cb; }
  if (f32) {  // the same matrices at the same offsets (in elements), as f32
    std::vector<float> ms(hs.size(), 0.f);
    auto put_m = [&](const HostTensor &t, size_t off) { std::copy(t.data.begin(), t.data.end(), ms.begin() + off); };
    put_m(word, o_word);
    put_m(get("embeddings.position_embeddings.weight"), o_pos);
    put_m(get("embeddings.token_type_embeddings.weight"), o_type);
    for (int l = 0; l < sh.layers; ++l) {
      const std::string p = "encoder.layer." + std::to_string(l) + ".";
      size_t at = lo[l].wqkv;
      for (const char *nm : {"query", "key", "value"}) {
        const HostTensor &w = get(p + "attention.self." + nm + ".weight");
        put_m(w, at);
        at += w.data.size();
      }
      put_m(get(p + "attention.output.dense.weight"), lo[l].wo);
      put_m(get(p + "intermediate.dense.weight"), lo[l].w1);
      put_m(get(p + "output.dense.weight"), lo[l].w2);
    }
    if (sh.classifier) put_m(get("pooler.dense.weight"), o_pw);
    e.weights32.reserve(ms.size() * 4);
    MRK_HIP(hipMemcpy(e.weights32.p, ms.data(), ms.size() * 4, hipMemcpyHostToDevice));
    e.device_bytes += (int64_t)(ms.size() * 4);
    const float *mb = e.weights32.as<float>();
    d.f32 = true;
    d.word32 = mb + o_word; d.pos32 = mb + o_pos; d.type32 = mb + o_type;
    for (auto &o : lo) d.layers32.push_back(LayerDev32{mb + o.wqkv, mb + o.wo, mb + o.w1, mb + o.w2});
    if (sh.classifier) d.pool_w32 = mb + o_pw;
  }
}

constexpr size_t GRAPH_MAX_TOKENS = 8192, GRAPH_MAX_CACHED = 256;

void drop_graphs(mrk_encoder &e) {
  for (auto &kv : e.graphs) (void)hipGraphExecDestroy(kv.second);
  e.graphs.clear();
}

// Which arithmetic a handle's calls run in is a property of the HANDLE (mrk_encoder_load_ex), never of the size of a call:
// the f32 kernels give a sequence the same bits whatever batch it travels in, so a request's scores do not depend on how
// many concurrent callers mrk_rank's combining front merged.  (ABI <= 7 had a size-based MRK_ENCODER_AUTO; it is f32 now.)
bool call_in_f32(const mrk_encoder &e) { return e.precision != MRK_ENCODER_FP16; }

// one forward pass over a padded id batch; `out` is host memory sized by the mode
void run_encoder(mrk_encoder &e, const int32_t *ids, const int32_t *types, const int32_t *mask, int n, int seq, int mode, float *out, bool f32) {
  if (n == 0) return;
  e.dev.f32 = f32 && !e.dev.layers32.empty();
  const EncoderShape &sh = e.dev.shape;
  if (seq > sh.max_pos) throw StatusError(MRK_ERR_INVALID_ARG, "encoder: sequence length " + std::to_string(seq) + " exceeds the model's " +
                                          std::to_string(sh.max_pos) + " positions");
  if (mode == MODE_LOGIT && !sh.classifier) throw StatusError(MRK_ERR_UNSUPPORTED, "encoder: the model has no pooler/classifier head (not a cross-encoder)");
  MRK_HIP(hipSetDevice(e.ctx->device));
  // Per-sequence results (pooled embeddings, logits) do not care where a sequence's tokens sit, so the batch runs PACKED:
  // real tokens back to back, no padding anywhere - every matrix product, LayerNorm and GELU touches real tokens only
  // (a batch of 3 840 search queries padded to its longest is 48 % padding).  A row's bits are the same either way
  // (tests/test_encoder_gpu.py pins padded == packed).  Hidden states (n x seq x H, parity checks) keep the padded layout,
  // and so does a mask that is not a prefix of ones.
  std::vector<int> lens;
  bool packed = mode != MODE_HIDDEN && switches().encoder_packed;
  size_t Mp = 0;
  int max_len = 0;
  if (packed) {
    lens.resize((size_t)n);
    for (int b = 0; b < n && packed; ++b) {
      int l = 0;
      while (l < seq && mask[(size_t)b * seq + l] != 0) ++l;
      for (int j = l; j < seq; ++j) packed = packed && mask[(size_t)b * seq + j] == 0;
      packed = packed && l > 0;
      lens[(size_t)b] = l;
      Mp += (size_t)l;
      max_len = std::max(max_len, l);
    }
  }
  const size_t M = packed ? Mp : (size_t)n * seq;
  const size_t out_n = mode == MODE_HIDDEN ? M * sh.hidden : mode == MODE_POOL ? (size_t)n * sh.hidden : (size_t)n;
  const size_t id_words = 3 * M + (packed ? (size_t)n + 1 : 0);
  // every buffer is sized before anything is enqueued: a captured graph holds raw pointers, so a buffer that
  // moves invalidates the graphs recorded so far
  const void *before[] = {e.h_ids.p, e.h_out.p, e.scratch.ids.p, e.scratch.x.p, e.scratch.xh.p, e.scratch.qkv.p,
                          e.scratch.ctx.p, e.scratch.mid.p, e.scratch.y.p, e.scratch.out.p};
  e.h_ids.reserve(id_words * 4);
  e.h_out.reserve(out_n * 4);
  e.scratch.ids.reserve(id_words * 4);
  e.scratch.out.reserve(out_n * 4);
  encoder_reserve(e.dev, e.scratch, 1, (int)M);
  const void *after[] = {e.h_ids.p, e.h_out.p, e.scratch.ids.p, e.scratch.x.p, e.scratch.xh.p, e.scratch.qkv.p,
                         e.scratch.ctx.p, e.scratch.mid.p, e.scratch.y.p, e.scratch.out.p};
  if (memcmp(before, after, sizeof before) != 0) drop_graphs(e);

  int32_t *h = e.h_ids.as<int32_t>();
  if (packed) {  // [ids | type_ids | position ids] x M, then cu x (n + 1)
    int32_t *cu = h + 3 * M;
    size_t at = 0;
    for (int b = 0; b < n; ++b) {
      cu[b] = (int32_t)at;
      const int l = lens[(size_t)b];
      memcpy(h + at, ids + (size_t)b * seq, (size_t)l * 4);
      if (types) memcpy(h + M + at, types + (size_t)b * seq, (size_t)l * 4); else memset(h + M + at, 0, (size_t)l * 4);
      for (int j = 0; j < l; ++j) h[2 * M + at + (size_t)j] = j;
      at += (size_t)l;
    }
    cu[n] = (int32_t)at;
  } else {
    memcpy(h, ids, M * 4);
    if (types) memcpy(h + M, types, M * 4); else memset(h + M, 0, M * 4);
    memcpy(h + 2 * M, mask, M * 4);
  }
  const int f_seq = packed ? max_len : seq, f_packed = packed ? (int)M : 0;

  // mrk_profile_enable: the forward pass (embedding lookup ... pooling / classifier, no copies) between two events on the
  // encoder's own stream, accumulated under the name "encoder" (bench.py: the dominant launch sequence of config 5)
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  const bool timed = e.ctx->profile && !(switches().encoder_graph && !e.dev.f32) && hipEventCreate(&ev_a) == hipSuccess && hipEventCreate(&ev_b) == hipSuccess;
  auto enqueue = [&]() {
    MRK_HIP(hipMemcpyAsync(e.scratch.ids.p, h, id_words * 4, hipMemcpyHostToDevice, e.stream));
    if (timed) (void)hipEventRecord(ev_a, e.stream);
    if (packed) encoder_forward_packed(e.dev, e.scratch, n, f_seq, f_packed, e.stream);
    else encoder_forward(e.dev, e.scratch, n, seq, e.stream);
    const float *src = e.scratch.x.as<float>();
    if (mode != MODE_HIDDEN) {
      src = e.scratch.out.as<float>();
      if (mode == MODE_POOL) encoder_meanpool(e.dev, e.scratch, n, f_seq, f_packed, e.scratch.out.as<float>(), e.stream);
      else encoder_classify(e.dev, e.scratch, n, f_seq, f_packed, e.scratch.out.as<float>(), e.stream);
    }
    if (timed) (void)hipEventRecord(ev_b, e.stream);
    MRK_HIP(hipMemcpyAsync(e.h_out.p, src, out_n * 4, hipMemcpyDeviceToHost, e.stream));
  };
  // Small shapes (a request's query, a request's item pairs) are ~40 kernels of a few microseconds each; their launch
  // sequence can be recorded once per (n, seq, mode) as a HIP graph and replayed.
  // Opt-in (MRK_ENCODER_GRAPH=1): measured 0.230 vs 0.240 ms for a 9-token query -- the forward pass of a small batch
  // is bound by the dependent-kernel chain, not by launch overhead -- and rocprofv3's kernel tracing crashes inside
  // the HIP runtime when a captured graph is launched.
  const bool use_graphs = switches().encoder_graph && !e.dev.f32;
  if (use_graphs && M <= GRAPH_MAX_TOKENS) {
    // (a packed batch's launch geometry depends on its token count and its longest sequence)
    const std::tuple<int, int, int> key(n, packed ? -(int)M : seq, mode * 4096 + (packed ? max_len : 0));
    auto it = e.graphs.find(key);
    if (it == e.graphs.end()) {
      if (e.graphs.size() >= GRAPH_MAX_CACHED) drop_graphs(e);
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      MRK_HIP(hipStreamBeginCapture(e.stream, hipStreamCaptureModeThreadLocal));
      try {
        enqueue();
      } catch (...) {
        (void)hipStreamEndCapture(e.stream, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
      }
      MRK_HIP(hipStreamEndCapture(e.stream, &graph));
      const hipError_t rc = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      MRK_HIP(rc);
      it = e.graphs.emplace(key, exec).first;
    }
    MRK_HIP(hipGraphLaunch(it->second, e.stream));
  } else {
    enqueue();
  }
  const hipError_t sync_rc = hipStreamSynchronize(e.stream);
  if (timed && sync_rc == hipSuccess) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev_a, ev_b) == hipSuccess) {
      std::lock_guard<std::mutex> lk(e.ctx->mu);
      auto &t = e.ctx->timers["encoder"];
      t.total_ms += ms;
      t.launches += 1;
    }
  }
  if (ev_a) (void)hipEventDestroy(ev_a);
  if (ev_b) (void)hipEventDestroy(ev_b);
  MRK_HIP(sync_rc);
  memcpy(out, e.h_out.p, out_n * 4);
}

void encode_texts(mrk_encoder &e, const char *const *a, const char *const *b, int n, int mode, float *out, bool f32) {
  std::vector<Encoding> rows;
  const int len = e.tok.encode_batch(a, b, n, rows);
  std::vector<int32_t> ids((size_t)n * len), types((size_t)n * len), mask((size_t)n * len);
  flatten(rows, len, ids.data(), types.data(), mask.data());
  run_encoder(e, ids.data(), types.data(), mask.data(), n, len, mode, out, f32);
}

}  // namespace

namespace mrk {

void encoder_retain(mrk_encoder *e) { e->refs.fetch_add(1); }

void encoder_embed_cached(mrk_encoder *e, const std::vector<std::string> &texts, std::vector<std::vector<float>> &out) {
  std::lock_guard<std::mutex> lk(e->mu);
  const int H = e->dev.shape.hidden;
  out.assign(texts.size(), std::vector<float>());
  const bool f32 = call_in_f32(*e);
  auto &cache = e->cache;
  std::vector<size_t> miss;
  for (size_t i = 0; i < texts.size(); ++i) {
    auto it = cache.find(texts[i]);
    if (it != cache.end()) out[i] = it->second; else miss.push_back(i);
  }
  constexpr size_t CHUNK = 256, CACHE_MAX = 1 << 16;
  std::vector<float> buf;
  for (size_t at = 0; at < miss.size(); at += CHUNK) {
    const size_t n = std::min(CHUNK, miss.size() - at);
    std::vector<const char *> ptrs(n);
    for (size_t k = 0; k < n; ++k) ptrs[k] = texts[miss[at + k]].c_str();
    buf.resize(n * H);
    encode_texts(*e, ptrs.data(), nullptr, (int)n, MODE_POOL, buf.data(), f32);
    if (cache.size() + n > CACHE_MAX) cache.clear();
    for (size_t k = 0; k < n; ++k) {
      out[miss[at + k]].assign(buf.begin() + k * H, buf.begin() + (k + 1) * H);
      cache[texts[miss[at + k]]] = out[miss[at + k]];
    }
  }
}

void encoder_score_rows(mrk_encoder *e, const std::vector<Encoding> &rows, float *out) {
  std::lock_guard<std::mutex> lk(e->mu);
  constexpr size_t CHUNK = 2048;
  std::vector<int32_t> ids, types, mask;
  for (size_t at = 0; at < rows.size(); at += CHUNK) {
    const size_t n = std::min(CHUNK, rows.size() - at);
    size_t len = 1;
    for (size_t k = 0; k < n; ++k) len = std::max(len, rows[at + k].ids.size());
    ids.assign(n * len, e->tok.pad_id());
    types.assign(n * len, 0);
    mask.assign(n * len, 0);
    for (size_t k = 0; k < n; ++k) {
      const Encoding &r = rows[at + k];
      std::copy(r.ids.begin(), r.ids.end(), ids.begin() + k * len);
      std::copy(r.type_ids.begin(), r.type_ids.end(), types.begin() + k * len);
      std::copy(r.mask.begin(), r.mask.end(), mask.begin() + k * len);
    }
    run_encoder(*e, ids.data(), types.data(), mask.data(), (int)n, (int)len, MODE_LOGIT, out + at, call_in_f32(*e));
  }
}

void encoder_release(mrk_encoder *e) {
  if (e->refs.fetch_sub(1) != 1) return;
  mrk_ctx *ctx = e->ctx;
  if (ctx) (void)hipSetDevice(ctx->device);
  if (e->stream) { (void)hipStreamSynchronize(e->stream); drop_graphs(*e); (void)hipStreamDestroy(e->stream); }
  delete e;
  if (ctx) ctx_release(ctx);
}

}  // namespace mrk

namespace {

// rows of a padded batch -> three n x len int32 arrays
void flatten(const std::vector<Encoding> &rows, int len, int32_t *ids, int32_t *types, int32_t *mask) {
  for (size_t i = 0; i < rows.size(); ++i) {
    if (ids) memcpy(ids + i * len, rows[i].ids.data(), sizeof(int32_t) * len);
    if (types) memcpy(types + i * len, rows[i].type_ids.data(), sizeof(int32_t) * len);
    if (mask) memcpy(mask + i * len, rows[i].mask.data(), sizeof(int32_t) * len);
  }
}
}  // namespace

extern "C" {

int mrk_tokenizer_load(const char *tokenizer_json, size_t len, mrk_tokenizer **out) {
  return guard([&] {
    need(tokenizer_json && out, "mrk_tokenizer_load: null argument");
    *out = new mrk_tokenizer{Tokenizer::from_json(tokenizer_json, len)};
  });
}

int mrk_tokenizer_encode_batch(mrk_tokenizer *tok, const char *const *a, const char *const *b, int n, int32_t *ids,
                               int32_t *type_ids, int32_t *mask, int capacity, int *seq_len) {
  return guard([&] {
    need(tok && (a || n == 0) && n >= 0 && seq_len, "mrk_tokenizer_encode_batch: null argument");
    std::vector<Encoding> rows;
    const int len = tok->tok.encode_batch(a, b, n, rows);
    *seq_len = len;
    if (len > capacity) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_tokenizer_encode_batch: capacity " + std::to_string(capacity) +
                                          " < padded length " + std::to_string(len));
    flatten(rows, len, ids, type_ids, mask);
  });
}

void mrk_tokenizer_free(mrk_tokenizer *tok) { delete tok; }

int mrk_encoder_load(mrk_ctx *ctx, const uint8_t *weights, size_t len, const char *tokenizer_json, size_t tok_len, int heads,
                     mrk_encoder **out) {
  return mrk_encoder_load_ex(ctx, weights, len, tokenizer_json, tok_len, heads, MRK_ENCODER_F32, out);
}

int mrk_encoder_load_ex(mrk_ctx *ctx, const uint8_t *weights, size_t len, const char *tokenizer_json, size_t tok_len, int heads,
                        int precision, mrk_encoder **out) {
  return guard([&] {
    need(ctx && weights && tokenizer_json && out, "mrk_encoder_load: null argument");
    need(precision == MRK_ENCODER_FP16 || precision == MRK_ENCODER_F32 || precision == MRK_ENCODER_AUTO, "mrk_encoder_load_ex: unknown precision");
    std::unique_ptr<mrk_encoder> e(new mrk_encoder);
    e->tok = Tokenizer::from_json(tokenizer_json, tok_len);
    Checkpoint ck = read_checkpoint(weights, len);
    if (heads > 0) ck.heads = heads;
    MRK_HIP(hipSetDevice(ctx->device));
    build_encoder(*e, ck, precision != MRK_ENCODER_FP16);   // (f32 keeps the matrices as f32 too)
    e->precision = precision;
    MRK_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->ctx = ctx;
    ctx_retain(ctx);
    *out = e.release();
  });
}

int mrk_checkpoint_describe(const uint8_t *weights, size_t len, char *out, size_t cap, size_t *needed) {
  return guard([&] {
    need(weights && needed, "mrk_checkpoint_describe: null argument");
    const Checkpoint ck = read_checkpoint(weights, len);
    std::string js = "{\"heads\": " + std::to_string(ck.heads) + ", \"tensors\": {";
    bool first = true;
    char num[64];
    for (auto &kv : ck.tensors) {
      double s = 0.0, a = 0.0;
      for (float v : kv.second.data) { s += (double)v; a += std::fabs((double)v); }
      js += std::string(first ? "" : ", ") + "\"" + kv.first + "\": {\"shape\": [";
      for (size_t i = 0; i < kv.second.shape.size(); ++i) js += (i ? ", " : "") + std::to_string(kv.second.shape[i]);
      snprintf(num, sizeof num, "%.17g", s);
      js += std::string("], \"sum\": ") + num;
      snprintf(num, sizeof num, "%.17g", a);
      js += std::string(", \"abs_sum\": ") + num + "}";
      first = false;
    }
    js += "}}";
    *needed = js.size() + 1;
    if (!out || cap < js.size() + 1) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_checkpoint_describe: output buffer too small");
    memcpy(out, js.c_str(), js.size() + 1);
  });
}

int mrk_encoder_get_info(mrk_encoder *enc, mrk_encoder_info *info) {
  return guard([&] {
    need(enc && info, "mrk_encoder_get_info: null argument");
    const EncoderShape &s = enc->dev.shape;
    *info = mrk_encoder_info{s.layers, s.hidden, s.heads, s.inter, s.vocab, s.max_pos, s.type_vocab, s.classifier ? 1 : 0,
                             enc->tok.max_length(), enc->device_bytes, (double)s.eps};
  });
}

int mrk_encoder_hidden_ids(mrk_encoder *enc, const int32_t *ids, const int32_t *type_ids, const int32_t *mask, int n, int seq_len,
                           float *out) {
  return guard([&] {
    need(enc && ids && mask && out && n >= 0 && seq_len > 0, "mrk_encoder_hidden_ids: bad argument");
    std::lock_guard<std::mutex> lk(enc->mu);
    run_encoder(*enc, ids, type_ids, mask, n, seq_len, MODE_HIDDEN, out, call_in_f32(*enc));
  });
}

int mrk_encoder_embed_ids(mrk_encoder *enc, const int32_t *ids, const int32_t *type_ids, const int32_t *mask, int n, int seq_len,
                          float *out) {
  return guard([&] {
    need(enc && ids && mask && out && n >= 0 && seq_len > 0, "mrk_encoder_embed_ids: bad argument");
    std::lock_guard<std::mutex> lk(enc->mu);
    run_encoder(*enc, ids, type_ids, mask, n, seq_len, MODE_POOL, out, call_in_f32(*enc));
  });
}

int mrk_encoder_score_ids(mrk_encoder *enc, const int32_t *ids, const int32_t *type_ids, const int32_t *mask, int n, int seq_len,
                          float *out) {
  return guard([&] {
    need(enc && ids && mask && out && n >= 0 && seq_len > 0, "mrk_encoder_score_ids: bad argument");
    std::lock_guard<std::mutex> lk(enc->mu);
    run_encoder(*enc, ids, type_ids, mask, n, seq_len, MODE_LOGIT, out, call_in_f32(*enc));
  });
}

int mrk_encoder_embed(mrk_encoder *enc, const char *const *texts, int n, float *out) {
  return guard([&] {
    need(enc && (texts || n == 0) && out && n >= 0, "mrk_encoder_embed: bad argument");
    if (n == 0) return;
    std::lock_guard<std::mutex> lk(enc->mu);
    encode_texts(*enc, texts, nullptr, n, MODE_POOL, out, call_in_f32(*enc));
  });
}

int mrk_encoder_score_pairs(mrk_encoder *enc, const char *const *a, const char *const *b, int n, float *out) {
  return guard([&] {
    need(enc && ((a && b) || n == 0) && out && n >= 0, "mrk_encoder_score_pairs: bad argument");
    if (n == 0) return;  // OnnxCrossEncoder.scala:23-24: empty batch -> empty result
    std::lock_guard<std::mutex> lk(enc->mu);
    encode_texts(*enc, a, b, n, MODE_LOGIT, out, call_in_f32(*enc));
  });
}

void mrk_encoder_free(mrk_encoder *enc) {
  if (enc) encoder_release(enc);
}

}  // extern "C"

extern "C" int mrk_config_bind_encoder(mrk_ctx *ctx, const char *feature, mrk_encoder *enc) {
  return guard([&] {
    need(ctx && feature && enc, "mrk_config_bind_encoder: null argument");
    bind_encoder(ctx, feature, enc);
  });
}
