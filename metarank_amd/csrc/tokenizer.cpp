// See tokenizer.hpp.  The pipeline follows HuggingFace `tokenizers` (the library behind DJL's HuggingFaceTokenizer,
// which the reference instantiates at ml/onnx/sbert/OnnxSession.scala:42): normalizers/bert.rs, pre_tokenizers/bert.rs,
// models/wordpiece/mod.rs, processors/{bert,template}.rs, utils/truncation.rs, utils/padding.rs.  Parity is pinned by
// tests/test_tokenizer.py against that library itself (installed in the image) and by tests/golden/tokenizer_*.json.
#include "tokenizer.hpp"

#include <algorithm>
#include <cstring>

#include "json.hpp"
#include "runtime.hpp"
#include "unicode_tables.hpp"

namespace mrk {
namespace {

// ---- UTF-8 ------------------------------------------------------------------------------------
void decode_utf8(const char *s, size_t n, std::vector<uint32_t> &out) {
  const unsigned char *p = (const unsigned char *)s, *e = p + n;
  while (p < e) {
    uint32_t c = *p;
    if (c < 0x80) { out.push_back(c); ++p; continue; }
    const int need = (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : -1;
    if (need < 0 || e - p <= need) { out.push_back(0xFFFD); ++p; continue; }
    uint32_t cp = need == 1 ? (c & 0x1F) : need == 2 ? (c & 0x0F) : (c & 0x07);
    bool ok = true;
    for (int k = 1; k <= need; ++k) {
      if ((p[k] & 0xC0) != 0x80) { ok = false; break; }
      cp = (cp << 6) | (p[k] & 0x3F);
    }
    if (!ok || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) { out.push_back(0xFFFD); ++p; continue; }
    out.push_back(cp);
    p += need + 1;
  }
}

void encode_utf8(const uint32_t *cps, size_t n, std::string &out) {
  for (size_t i = 0; i < n; ++i) {
    uint32_t cp = cps[i];
    if (cp < 0x80) out.push_back((char)cp);
    else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
      out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      out.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
      out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
      out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
    }
  }
}

// ---- Unicode properties (tables generated from the UCD, tools/gen_unicode_tables.py) -------------
template <typename R, size_t N>
const R *find_range(const R (&tab)[N], uint32_t cp) {
  size_t lo = 0, hi = N;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (tab[mid].hi < cp) lo = mid + 1; else hi = mid;
  }
  return lo < N && tab[lo].lo <= cp ? &tab[lo] : nullptr;
}
template <typename M, size_t N>
const M *find_map(const M (&tab)[N], uint32_t cp) {
  size_t lo = 0, hi = N;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (tab[mid].cp < cp) lo = mid + 1; else hi = mid;
  }
  return lo < N && tab[lo].cp == cp ? &tab[lo] : nullptr;
}

// Rust char::is_whitespace (White_Space property)
bool is_ws(uint32_t c) {
  return (c >= 9 && c <= 13) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
         c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
// normalizers/bert.rs is_control: \t \n \r are not control; otherwise categories Cc, Cf, Co
bool is_control(uint32_t c) {
  if (c == '\t' || c == '\n' || c == '\r') return false;
  return find_range(uni::OTHER, c) != nullptr;
}
bool is_chinese(uint32_t c) {
  return (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0x3400 && c <= 0x4DBF) || (c >= 0x20000 && c <= 0x2A6DF) ||
         (c >= 0x2A700 && c <= 0x2B73F) || (c >= 0x2B740 && c <= 0x2B81F) || (c >= 0x2B920 && c <= 0x2CEAF) ||
         (c >= 0xF900 && c <= 0xFAFF) || (c >= 0x2F800 && c <= 0x2FA1F);
}
// pre_tokenizers/bert.rs is_bert_punc: ASCII punctuation or a P* category
bool is_punct(uint32_t c) {
  if (c < 0x80) return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126);
  return find_range(uni::PUNCT, c) != nullptr;
}
uint8_t ccc_of(uint32_t c) {
  if (c < 0x300) return 0;
  const uni::RangeV *r = find_range(uni::CCC, c);
  return r ? r->v : 0;
}

void nfd(const std::vector<uint32_t> &in, std::vector<uint32_t> &out) {
  out.clear();
  out.reserve(in.size() + 8);
  for (uint32_t c : in) {
    if (c < 0xC0) { out.push_back(c); continue; }
    if (c >= 0xAC00 && c <= 0xD7A3) {  // Hangul syllable: algorithmic decomposition
      uint32_t s = c - 0xAC00;
      out.push_back(0x1100 + s / 588);
      out.push_back(0x1161 + (s % 588) / 28);
      if (s % 28) out.push_back(0x11A7 + s % 28);
      continue;
    }
    if (const uni::Map4 *m = find_map(uni::NFD, c)) {
      for (int k = 0; k < m->n; ++k) out.push_back(m->to[k]);
    } else {
      out.push_back(c);
    }
  }
  // canonical ordering: stable sort of every run of non-starters by combining class
  size_t i = 0;
  while (i < out.size()) {
    if (ccc_of(out[i]) == 0) { ++i; continue; }
    size_t j = i;
    while (j < out.size() && ccc_of(out[j]) != 0) ++j;
    if (j - i > 1) std::stable_sort(out.begin() + i, out.begin() + j, [](uint32_t a, uint32_t b) { return ccc_of(a) < ccc_of(b); });
    i = j;
  }
}

[[noreturn]] void unsupported(const std::string &what) {
  throw StatusError(MRK_ERR_UNSUPPORTED, "tokenizer.json: " + what + " is not supported (BERT WordPiece pipelines only)");
}

const std::string &type_of(const json::Value &v) { return v.at("type").as_string(); }

}  // namespace

// ---- normalizers/bert.rs -----------------------------------------------------------------------
void Tokenizer::normalize(const std::vector<uint32_t> &in, std::vector<uint32_t> &out) const {
  std::vector<uint32_t> a, b;
  a.reserve(in.size() + 8);
  for (uint32_t c : in) {
    if (clean_text_) {
      if (c == 0 || c == 0xFFFD || is_control(c)) continue;
      if (c == '\t' || c == '\n' || c == '\r' || is_ws(c)) c = ' ';
    }
    if (chinese_ && is_chinese(c)) { a.push_back(' '); a.push_back(c); a.push_back(' '); }
    else a.push_back(c);
  }
  if (strip_accents_) {
    nfd(a, b);
    a.clear();
    for (uint32_t c : b)
      if (!(c >= 0x300 && find_range(uni::MN, c))) a.push_back(c);
  }
  out.clear();
  if (lowercase_) {
    for (uint32_t c : a) {
      if (c < 0x80) { out.push_back(c >= 'A' && c <= 'Z' ? c + 32 : c); continue; }
      if (const uni::Map3 *m = find_map(uni::LOWER, c)) for (int k = 0; k < m->n; ++k) out.push_back(m->to[k]);
      else out.push_back(c);
    }
  } else {
    out = a;
  }
}

// ---- models/wordpiece/mod.rs: greedy longest-match-first ------------------------------------------
void Tokenizer::wordpiece(const std::vector<uint32_t> &word, std::vector<int32_t> &out) const {
  if ((int)word.size() > max_chars_) { out.push_back(unk_id_); return; }
  const size_t mark = out.size();
  size_t start = 0;
  std::string key;
  while (start < word.size()) {
    size_t end = word.size();
    int32_t found = -1;
    while (start < end) {
      key.clear();
      if (start > 0) key = prefix_;
      encode_utf8(word.data() + start, end - start, key);
      auto it = vocab_.find(key);
      if (it != vocab_.end()) { found = it->second; break; }
      --end;
    }
    if (found < 0) { out.resize(mark); out.push_back(unk_id_); return; }
    out.push_back(found);
    start = end;
  }
}

// normalise + BertPreTokenizer (split on whitespace, isolate punctuation) + WordPiece of one text span
void Tokenizer::tokenize_plain(const std::vector<uint32_t> &cps, std::vector<int32_t> &out) const {
  std::vector<uint32_t> norm, word;
  normalize(cps, norm);
  for (size_t i = 0; i <= norm.size(); ++i) {
    const bool end = i == norm.size();
    const uint32_t c = end ? ' ' : norm[i];
    if (end || is_ws(c)) {
      if (!word.empty()) { wordpiece(word, out); word.clear(); }
    } else if (is_punct(c)) {
      if (!word.empty()) { wordpiece(word, out); word.clear(); }
      word.push_back(c);
      wordpiece(word, out);
      word.clear();
    } else {
      word.push_back(c);
    }
  }
}

std::vector<int32_t> Tokenizer::pieces(const std::string &text) const {
  std::vector<int32_t> out;
  // tokenizer/added_vocabulary.rs: special tokens are cut out of the raw text first (leftmost-longest)
  size_t pos = 0, seg = 0;
  std::vector<uint32_t> cps;
  auto flush = [&](size_t upto) {
    if (upto > seg) {
      cps.clear();
      decode_utf8(text.data() + seg, upto - seg, cps);
      tokenize_plain(cps, out);
    }
  };
  while (pos < text.size() && !added_.empty()) {
    const std::pair<std::string, int32_t> *best = nullptr;
    for (auto &a : added_)
      if (!a.first.empty() && text.compare(pos, a.first.size(), a.first) == 0 && (!best || a.first.size() > best->first.size())) best = &a;
    if (best) {
      flush(pos);
      out.push_back(best->second);
      pos += best->first.size();
      seg = pos;
    } else {
      ++pos;
    }
  }
  flush(text.size());
  return out;
}

// utils/truncation.rs truncate_encodings (LongestFirst, direction Right) + the post-processor template
Encoding Tokenizer::assemble(std::vector<int32_t> a, const std::vector<int32_t> *b_in) const {
  std::vector<int32_t> b;
  if (b_in) b = *b_in;
  const std::vector<Piece> &tpl = b_in ? pair_ : single_;
  int n_added = 0;
  for (auto &p : tpl) n_added += p.special ? 1 : 0;
  const size_t max_len = (size_t)std::max(0, max_length_ - n_added);
  const size_t total = a.size() + b.size();
  if (max_len == 0) { a.clear(); b.clear(); }
  else if (total > max_len) {
    if (b_in) {
      size_t n1 = a.size(), n2 = b.size();
      bool swap = false;
      if (n1 > n2) { swap = true; std::swap(n1, n2); }
      if (n1 > max_len) n2 = n1; else n2 = std::max(n1, max_len - n1);
      if (n1 + n2 > max_len) { n1 = max_len / 2; n2 = n1 + max_len % 2; }
      if (swap) std::swap(n1, n2);
      if (a.size() > n1) a.resize(n1);
      if (b.size() > n2) b.resize(n2);
    } else {
      a.resize(max_len);
    }
  }
  Encoding e;
  for (auto &p : tpl) {
    if (p.special) { e.ids.push_back(p.id); e.type_ids.push_back(p.type); }
    else {
      const std::vector<int32_t> &src = p.seq == 0 ? a : b;
      for (int32_t id : src) { e.ids.push_back(id); e.type_ids.push_back(p.type); }
    }
  }
  e.mask.assign(e.ids.size(), 1);
  return e;
}

Encoding Tokenizer::encode(const std::string &a, const std::string *b) const {
  std::vector<int32_t> pa = pieces(a), pb;
  if (b) pb = pieces(*b);
  return assemble(std::move(pa), b ? &pb : nullptr);
}

int Tokenizer::encode_batch(const char *const *a, const char *const *b, int n, std::vector<Encoding> &out) const {
  size_t longest = 0;
  const size_t first = out.size();
  for (int i = 0; i < n; ++i) {
    std::string sb;
    if (b) sb = b[i] ? b[i] : "";
    out.push_back(encode(a[i] ? a[i] : "", b ? &sb : nullptr));
    longest = std::max(longest, out.back().ids.size());
  }
  for (size_t i = first; i < out.size(); ++i) {  // utils/padding.rs, BatchLongest, direction Right
    out[i].ids.resize(longest, pad_id_);
    out[i].type_ids.resize(longest, pad_type_);
    out[i].mask.resize(longest, 0);
  }
  return (int)longest;
}

// ---- tokenizer.json ----------------------------------------------------------------------------------
Tokenizer Tokenizer::from_json(const char *text, size_t len) {
  json::Value root;
  try {
    root = json::parse(text, len);
  } catch (const std::exception &e) {
    throw StatusError(MRK_ERR_PARSE, std::string("tokenizer.json: ") + e.what());
  }
  Tokenizer t;
  try {
    const json::Value &model = root.at("model");
    if (const json::Value *ty = model.find("type"))
      if (!ty->is_null() && ty->as_string() != "WordPiece") unsupported("model type " + ty->as_string());
    for (auto &kv : model.at("vocab").obj) t.vocab_.emplace(kv.first, (int32_t)kv.second.as_int());
    t.vocab_size_ = t.vocab_.size();
    if (const json::Value *p = model.find("continuing_subword_prefix")) if (!p->is_null()) t.prefix_ = p->as_string();
    if (const json::Value *p = model.find("max_input_chars_per_word"))
      if (!p->is_null()) {
        const int64_t v = p->as_int();
        if (v < 0 || v > (1 << 20)) throw StatusError(MRK_ERR_PARSE, "tokenizer.json: max_input_chars_per_word out of range");
        t.max_chars_ = (int)v;
      }
    std::string unk = "[UNK]";
    if (const json::Value *p = model.find("unk_token")) if (!p->is_null()) unk = p->as_string();
    auto u = t.vocab_.find(unk);
    if (u == t.vocab_.end()) throw StatusError(MRK_ERR_PARSE, "tokenizer.json: unk_token " + unk + " is not in the vocabulary");
    t.unk_id_ = u->second;

    if (const json::Value *n = root.find("normalizer")) {
      if (n->is_null()) { t.clean_text_ = t.chinese_ = t.strip_accents_ = t.lowercase_ = false; }
      else {
        if (type_of(*n) != "BertNormalizer") unsupported("normalizer " + type_of(*n));
        auto flag = [&](const char *k, bool dflt) { const json::Value *v = n->find(k); return v && !v->is_null() ? v->as_bool() : dflt; };
        t.clean_text_ = flag("clean_text", true);
        t.chinese_ = flag("handle_chinese_chars", true);
        t.lowercase_ = flag("lowercase", true);
        t.strip_accents_ = flag("strip_accents", t.lowercase_);  // None follows `lowercase`
      }
    }
    if (const json::Value *p = root.find("pre_tokenizer"))
      if (p->is_null() || type_of(*p) != "BertPreTokenizer") unsupported("pre_tokenizer other than BertPreTokenizer");

    if (const json::Value *ad = root.find("added_tokens"))
      if (ad->is_array())
        for (auto &a : ad->arr) {
          const json::Value *norm = a.find("normalized");
          if (norm && !norm->is_null() && norm->as_bool()) continue;  // matched on normalised text: not a BERT special
          const int32_t id = (int32_t)a.at("id").as_int();
          t.added_.emplace_back(a.at("content").as_string(), id);
          t.vocab_size_ = std::max(t.vocab_size_, (size_t)id + 1);
        }
    for (auto &kv : t.vocab_) t.vocab_size_ = std::max(t.vocab_size_, (size_t)kv.second + 1);

    // post-processor -> templates
    auto special = [&](const std::string &tok, int32_t id, int type) { return Piece{true, id, type, 0}; };
    const json::Value *pp = root.find("post_processor");
    if (!pp || pp->is_null()) {
      t.single_ = {Piece{false, 0, 0, 0}};
      t.pair_ = {Piece{false, 0, 0, 0}, Piece{false, 0, 1, 1}};
    } else if (type_of(*pp) == "BertProcessing") {
      const int32_t sep = (int32_t)pp->at("sep").arr.at(1).as_int(), cls = (int32_t)pp->at("cls").arr.at(1).as_int();
      t.single_ = {special("", cls, 0), Piece{false, 0, 0, 0}, special("", sep, 0)};
      t.pair_ = {special("", cls, 0), Piece{false, 0, 0, 0}, special("", sep, 0), Piece{false, 0, 1, 1}, special("", sep, 1)};
    } else if (type_of(*pp) == "TemplateProcessing") {
      const json::Value &sp = pp->at("special_tokens");
      auto tpl = [&](const json::Value &arr, std::vector<Piece> &dst) {
        for (auto &el : arr.arr) {
          if (const json::Value *s = el.find("SpecialToken")) {
            const json::Value &def = sp.at(s->at("id").as_string().c_str());
            const int type = (int)s->at("type_id").as_int();
            for (auto &idv : def.at("ids").arr) dst.push_back(Piece{true, (int32_t)idv.as_int(), type, 0});
          } else if (const json::Value *q = el.find("Sequence")) {
            dst.push_back(Piece{false, 0, (int)q->at("type_id").as_int(), q->at("id").as_string() == "A" ? 0 : 1});
          } else {
            unsupported("template element");
          }
        }
      };
      tpl(pp->at("single"), t.single_);
      tpl(pp->at("pair"), t.pair_);
    } else {
      unsupported("post_processor " + type_of(*pp));
    }

    // DJL: truncation=true -> LongestFirst at the JSON's max_length, else 512; padding=true -> BatchLongest
    if (const json::Value *tr = root.find("truncation"))
      if (!tr->is_null()) {
        const int64_t ml = tr->at("max_length").as_int();
        if (ml < 0 || ml > (1 << 20)) throw StatusError(MRK_ERR_PARSE, "tokenizer.json: truncation.max_length out of range");
        t.max_length_ = (int)ml;
        if (const json::Value *d = tr->find("direction")) if (!d->is_null() && d->as_string() != "Right") unsupported("left truncation");
      }
    auto pad = t.vocab_.find("[PAD]");
    if (pad != t.vocab_.end()) t.pad_id_ = pad->second;
    if (const json::Value *pd = root.find("padding"))
      if (!pd->is_null()) {
        t.pad_id_ = (int32_t)pd->at("pad_id").as_int();
        if (const json::Value *v = pd->find("pad_type_id")) t.pad_type_ = (int32_t)v->as_int();
        if (const json::Value *d = pd->find("direction")) if (!d->is_null() && d->as_string() != "Right") unsupported("left padding");
      }
  } catch (const StatusError &) {
    throw;
  } catch (const std::exception &e) {
    throw StatusError(MRK_ERR_PARSE, std::string("tokenizer.json: ") + e.what());
  }
  return t;
}

}  // namespace mrk
