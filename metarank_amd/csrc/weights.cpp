// Checkpoint readers for the text encoder: the file the reference hands to onnxruntime
// (`pytorch_model.onnx`, ml/onnx/sbert/OnnxSession.scala:29-35,49) and the same weights as safetensors.
//
// ONNX: only the protobuf wire format and the handful of ModelProto / GraphProto / NodeProto / TensorProto fields
// needed to recover the BERT parameters are read (onnx.proto3: ModelProto.graph=7; GraphProto.node=1,
// initializer=5; NodeProto.input=1, output=2, op_type=4, attribute=5; AttributeProto.name=1, t=5;
// TensorProto.dims=1, data_type=2, float_data=4, int64_data=7, name=8, raw_data=9, data_location=14).
// A torch.onnx export keeps parameter names for embeddings, LayerNorm and biases and stores every Linear weight
// transposed as an anonymous `onnx::MatMul_N` initializer; those are recovered by following bias -> Add -> MatMul.
// Pinned by tests/test_encoder_cpu.py on tests/golden/{encoder,cross}_tiny.onnx (real torch.onnx exports).
#include <cstring>

#include "encoder.hpp"
#include "json.hpp"

namespace mrk {
namespace {

[[noreturn]] void bad(const std::string &m) { throw StatusError(MRK_ERR_PARSE, "encoder weights: " + m); }

float half_to_float(uint16_t h) {
  const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 0x1F, m = h & 0x3FF;
  uint32_t u;
  if (e == 0) {
    if (m == 0) u = s;
    else {
      int sh = 0;
      uint32_t mm = m;
      while (!(mm & 0x400)) { mm <<= 1; ++sh; }
      u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 0x3FF) << 13);
    }
  } else if (e == 31) u = s | 0x7F800000u | (m << 13);
  else u = s | ((e - 15 + 127) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// strip wrapper prefixes ("bert.", "m.", "0.auto_model.") down to the HuggingFace BertModel parameter name
std::string canonical(const std::string &name) {
  static const char *anchors[] = {"embeddings.", "encoder.layer.", "pooler.dense.", "classifier."};
  size_t best = std::string::npos;
  for (const char *a : anchors) {
    size_t p = name.find(a);
    if (p != std::string::npos && (p == 0 || name[p - 1] == '.') && p < best) best = p;
  }
  return best == std::string::npos ? std::string() : name.substr(best);
}

// ---- safetensors -----------------------------------------------------------------------------------
Checkpoint read_safetensors(const uint8_t *b, size_t len) {
  uint64_t hl;
  memcpy(&hl, b, 8);
  if (hl > len - 8) bad("safetensors header length exceeds the file");
  json::Value h = json::parse((const char *)b + 8, (size_t)hl);
  const uint8_t *data = b + 8 + hl;
  const size_t dlen = len - 8 - (size_t)hl;
  Checkpoint ck;
  for (auto &kv : h.obj) {
    if (kv.first == "__metadata__") {
      if (const json::Value *v = kv.second.find("num_attention_heads")) ck.heads = atoi(v->as_string().c_str());
      continue;
    }
    const std::string name = canonical(kv.first);
    if (name.empty()) continue;
    const std::string &dt = kv.second.at("dtype").as_string();
    HostTensor t;
    for (auto &d : kv.second.at("shape").arr) t.shape.push_back(d.as_int());
    const uint64_t lo = (uint64_t)kv.second.at("data_offsets").arr.at(0).as_int(), hi = (uint64_t)kv.second.at("data_offsets").arr.at(1).as_int();
    const int64_t n = t.numel();
    const size_t esz = dt == "F32" ? 4 : (dt == "F16" || dt == "BF16") ? 2 : 0;
    if (!esz) continue;  // integer buffers (position_ids)
    if (hi > dlen || lo > hi || hi - lo != (uint64_t)n * esz) bad("tensor " + kv.first + " has inconsistent offsets");
    t.data.resize((size_t)n);
    const uint8_t *p = data + lo;
    for (int64_t i = 0; i < n; ++i) {
      if (esz == 4) memcpy(&t.data[i], p + 4 * i, 4);
      else {
        uint16_t hbits;
        memcpy(&hbits, p + 2 * i, 2);
        if (dt == "F16") t.data[i] = half_to_float(hbits);
        else { uint32_t u = (uint32_t)hbits << 16; memcpy(&t.data[i], &u, 4); }
      }
    }
    ck.tensors.emplace(name, std::move(t));
  }
  return ck;
}

// ---- protobuf wire format --------------------------------------------------------------------------
struct PB {
  const uint8_t *p, *e;
  PB(const uint8_t *b, size_t n) : p(b), e(b + n) {}
  uint64_t varint() {
    uint64_t v = 0;
    for (int sh = 0; sh < 70; sh += 7) {
      if (p >= e) bad("truncated protobuf varint");
      const uint8_t c = *p++;
      v |= (uint64_t)(c & 0x7F) << sh;
      if (!(c & 0x80)) return v;
    }
    bad("protobuf varint too long");
  }
  // one field: wire types 0 (val), 1 (8 bytes at data), 2 (len bytes at data), 5 (4 bytes at data)
  bool next(uint32_t &field, uint32_t &wt, uint64_t &val, const uint8_t *&data, size_t &n) {
    if (p >= e) return false;
    const uint64_t key = varint();
    field = (uint32_t)(key >> 3);
    wt = (uint32_t)(key & 7);
    val = 0; data = nullptr; n = 0;
    switch (wt) {
      case 0: val = varint(); break;
      case 1: data = p; n = 8; break;
      case 2: n = (size_t)varint(); data = p; break;
      case 5: data = p; n = 4; break;
      default: bad("unsupported protobuf wire type");
    }
    if (wt != 0) {
      if (n > (size_t)(e - p)) bad("truncated protobuf field");
      p += n;
    }
    return true;
  }
};

struct PbTensor {
  std::string name;
  std::vector<int64_t> dims;
  int dtype = 0;  // 1 f32, 7 i64, 10 f16, 6 i32
  const uint8_t *raw = nullptr; size_t raw_n = 0;
  std::vector<float> f32; std::vector<int64_t> i64;
  bool external = false;

  int64_t numel() const { int64_t n = 1; for (int64_t d : dims) n *= d; return n; }
  std::vector<int64_t> ints() const {
    if (!i64.empty() || dtype != 7) return i64;
    std::vector<int64_t> v(raw_n / 8);
    if (raw_n) memcpy(v.data(), raw, v.size() * 8);
    return v;
  }
  HostTensor floats() const {
    if (external) bad("initializer " + name + " uses external data, which is not supported");
    HostTensor t;
    t.shape = dims;
    const int64_t n = numel();
    t.data.resize((size_t)n);
    if (dtype == 1) {
      if (!f32.empty()) { if ((int64_t)f32.size() != n) bad("float_data size mismatch in " + name); t.data = f32; }
      else { if ((int64_t)raw_n != 4 * n) bad("raw_data size mismatch in " + name); if (n) memcpy(t.data.data(), raw, raw_n); }
    } else if (dtype == 10) {
      if ((int64_t)raw_n != 2 * n) bad("raw_data size mismatch in " + name);
      for (int64_t i = 0; i < n; ++i) { uint16_t h; memcpy(&h, raw + 2 * i, 2); t.data[i] = half_to_float(h); }
    } else bad("initializer " + name + " has an unsupported element type");
    return t;
  }
};

PbTensor parse_tensor(const uint8_t *b, size_t n) {
  PbTensor t;
  PB pb(b, n);
  uint32_t f, wt; uint64_t v; const uint8_t *d; size_t len;
  while (pb.next(f, wt, v, d, len)) {
    switch (f) {
      case 1: if (wt == 0) t.dims.push_back((int64_t)v); else { PB q(d, len); while (q.p < q.e) t.dims.push_back((int64_t)q.varint()); } break;
      case 2: t.dtype = (int)v; break;
      case 4: if (wt == 5) { float x; memcpy(&x, d, 4); t.f32.push_back(x); } else { t.f32.resize(len / 4); if (len) memcpy(t.f32.data(), d, len); } break;
      case 7: if (wt == 0) t.i64.push_back((int64_t)v); else { PB q(d, len); while (q.p < q.e) t.i64.push_back((int64_t)q.varint()); } break;
      case 8: t.name.assign((const char *)d, len); break;
      case 9: t.raw = d; t.raw_n = len; break;
      case 14: t.external = v == 1; break;
      default: break;
    }
  }
  return t;
}

struct PbNode {
  std::string op;
  std::vector<std::string> in, out;
  PbTensor value;  // Constant: attribute "value"
  bool has_value = false;
};

PbNode parse_node(const uint8_t *b, size_t n) {
  PbNode nd;
  PB pb(b, n);
  uint32_t f, wt; uint64_t v; const uint8_t *d; size_t len;
  while (pb.next(f, wt, v, d, len)) {
    if (f == 1) nd.in.emplace_back((const char *)d, len);
    else if (f == 2) nd.out.emplace_back((const char *)d, len);
    else if (f == 4) nd.op.assign((const char *)d, len);
    else if (f == 5) {
      PB a(d, len);
      uint32_t af, awt; uint64_t av; const uint8_t *ad; size_t alen;
      std::string an; const uint8_t *tp = nullptr; size_t tn = 0;
      while (a.next(af, awt, av, ad, alen)) {
        if (af == 1) an.assign((const char *)ad, alen);
        else if (af == 5) { tp = ad; tn = alen; }
      }
      if (an == "value" && tp) { nd.value = parse_tensor(tp, tn); nd.has_value = true; }
    }
  }
  return nd;
}

// the projections are reshaped to [batch, seq, heads, head_dim] or, in newer exports, [batch, seq, -1, head_dim]
int heads_of(int64_t heads, int64_t head_dim, int64_t hidden) {
  if (head_dim <= 0 || hidden % head_dim) return 0;
  if (heads == -1 || heads * head_dim == hidden) return (int)(hidden / head_dim);
  return 0;
}

Checkpoint read_onnx(const uint8_t *b, size_t len) {
  PB model(b, len);
  uint32_t f, wt; uint64_t v; const uint8_t *d; size_t n;
  const uint8_t *graph = nullptr; size_t graph_n = 0;
  while (model.next(f, wt, v, d, n))
    if (f == 7 && wt == 2) { graph = d; graph_n = n; }
  if (!graph) bad("no graph in the ONNX model");
  std::map<std::string, PbTensor> inits;
  std::vector<PbNode> nodes;
  PB g(graph, graph_n);
  while (g.next(f, wt, v, d, n)) {
    if (f == 5 && wt == 2) { PbTensor t = parse_tensor(d, n); std::string nm = t.name; inits.emplace(std::move(nm), std::move(t)); }
    else if (f == 1 && wt == 2) nodes.push_back(parse_node(d, n));
  }
  std::map<std::string, const PbNode *> producer;
  for (auto &nd : nodes) for (auto &o : nd.out) producer[o] = &nd;
  auto constant = [&](const std::string &name) -> const PbTensor * {
    auto it = inits.find(name);
    if (it != inits.end()) return &it->second;
    auto p = producer.find(name);
    if (p != producer.end() && p->second->op == "Constant" && p->second->has_value) return &p->second->value;
    return nullptr;
  };

  Checkpoint ck;
  for (auto &kv : inits) {
    const std::string name = canonical(kv.first);
    if (name.empty() || (kv.second.dtype != 1 && kv.second.dtype != 10)) continue;
    ck.tensors.emplace(name, kv.second.floats());
  }
  // Linear weights exported as anonymous [in, out] MatMul operands: bias -> Add -> MatMul -> initializer
  for (auto &nd : nodes) {
    if (nd.op != "Add" || nd.in.size() != 2) continue;
    for (int side = 0; side < 2; ++side) {
      const std::string bias = canonical(nd.in[side]);
      if (bias.size() < 5 || bias.compare(bias.size() - 5, 5, ".bias") != 0 || !inits.count(nd.in[side])) continue;
      const std::string wname = bias.substr(0, bias.size() - 5) + ".weight";
      if (ck.tensors.count(wname)) continue;
      auto p = producer.find(nd.in[1 - side]);
      if (p == producer.end() || p->second->op != "MatMul" || p->second->in.size() != 2) continue;
      auto w = inits.find(p->second->in[1]);
      if (w == inits.end() || w->second.dims.size() != 2) continue;
      HostTensor src = w->second.floats(), dst;
      const int64_t in = src.shape[0], out = src.shape[1];
      dst.shape = {out, in};
      dst.data.resize(src.data.size());
      for (int64_t i = 0; i < in; ++i)
        for (int64_t o = 0; o < out; ++o) dst.data[o * in + i] = src.data[i * out + o];
      ck.tensors.emplace(wname, std::move(dst));
    }
  }
  // attention heads: the [batch, seq, heads, head_dim] reshape of the projections
  int64_t hidden = 0;
  auto q = ck.tensors.find("encoder.layer.0.attention.self.query.bias");
  if (q != ck.tensors.end()) hidden = q->second.numel();
  for (auto &nd : nodes) {
    if (ck.heads || !hidden) break;
    if (nd.op == "Reshape" && nd.in.size() == 2) {
      if (const PbTensor *s = constant(nd.in[1])) {
        std::vector<int64_t> e = s->ints();
        if (e.size() == 4) ck.heads = heads_of(e[2], e[3], hidden);
      } else {
        auto p = producer.find(nd.in[1]);
        if (p != producer.end() && p->second->op == "Concat" && p->second->in.size() == 4) {
          int64_t dims[2] = {0, 0};
          for (int k = 0; k < 2; ++k) {
            std::string src = p->second->in[2 + k];
            auto u = producer.find(src);
            if (u != producer.end() && u->second->op == "Unsqueeze" && !u->second->in.empty()) src = u->second->in[0];
            if (const PbTensor *c = constant(src)) { std::vector<int64_t> e = c->ints(); if (e.size() == 1) dims[k] = e[0]; }
          }
          ck.heads = heads_of(dims[0], dims[1], hidden);
        }
      }
    }
  }
  return ck;
}

}  // namespace

Checkpoint read_checkpoint(const uint8_t *bytes, size_t len) {
  if (!bytes || len < 16) bad("empty model file");
  uint64_t hl;
  memcpy(&hl, bytes, 8);
  if (hl < len && hl > 1 && bytes[8] == '{') return read_safetensors(bytes, len);
  return read_onnx(bytes, len);
}

}  // namespace mrk
