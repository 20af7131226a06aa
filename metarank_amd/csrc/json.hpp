// Minimal JSON + UBJSON reader producing one DOM.  Host-side only (model / config parsing).
//
// Numbers keep their source token so a caller that wants a float32 (XGBoost split conditions
// and leaf values) parses the decimal text straight to float — decimal -> double -> float can
// double-round, and XGBoost's own reader goes decimal -> float.  UBJSON numbers carry their
// binary value instead.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mrk {
namespace json {

struct Value;
using ValuePtr = std::unique_ptr<Value>;

enum class Type : uint8_t { Null, Bool, Number, String, Array, Object };

struct Value {
  Type type = Type::Null;
  bool b = false;
  // Number: either a text token (tok/toklen) or a binary value (is_bin).
  const char *tok = nullptr;
  uint32_t toklen = 0;
  bool is_bin = false;
  bool bin_is_f32 = false;
  double d = 0.0;
  float f = 0.f;
  int64_t i = 0;
  bool bin_is_int = false;
  std::string str;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;

  bool is_null() const { return type == Type::Null; }
  bool is_array() const { return type == Type::Array; }
  bool is_object() const { return type == Type::Object; }
  bool is_string() const { return type == Type::String; }
  bool is_number() const { return type == Type::Number; }

  const Value *find(const char *key) const {
    if (type != Type::Object) return nullptr;
    for (auto &kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  const Value &at(const char *key) const {
    const Value *v = find(key);
    if (!v) throw std::runtime_error(std::string("json: missing key '") + key + "'");
    return *v;
  }

  double as_double() const {
    if (type == Type::Number) {
      if (is_bin) return bin_is_int ? (double)i : (bin_is_f32 ? (double)f : d);
      char buf[64];
      return strtod(tokz(buf, sizeof buf), nullptr);
    }
    if (type == Type::String) return parse_string_number();
    if (type == Type::Bool) return b ? 1.0 : 0.0;
    throw std::runtime_error("json: value is not a number");
  }
  float as_float() const {
    if (type == Type::Number) {
      if (is_bin) return bin_is_int ? (float)i : (bin_is_f32 ? f : (float)d);
      char buf[64];
      return strtof(tokz(buf, sizeof buf), nullptr);
    }
    if (type == Type::String) return parse_string_float();   // (strtof, not strtod + narrowing: one rounding, as the library's own reader)
    if (type == Type::Bool) return b ? 1.f : 0.f;
    throw std::runtime_error("json: value is not a number");
  }
  int64_t as_int() const {
    if (type == Type::Number) {
      if (is_bin) return bin_is_int ? i : (int64_t)(bin_is_f32 ? (double)f : d);
      char buf[64];
      const char *z = tokz(buf, sizeof buf);
      char *end = nullptr;
      long long v = strtoll(z, &end, 10);
      if (end && *end == 0) return v;
      return (int64_t)strtod(z, nullptr);
    }
    if (type == Type::String) return (int64_t)parse_string_number();
    if (type == Type::Bool) return b ? 1 : 0;
    throw std::runtime_error("json: value is not an integer");
  }
  bool as_bool() const {
    if (type == Type::Bool) return b;
    if (type == Type::Number) return as_double() != 0.0;
    if (type == Type::String) return str == "true" || str == "1";
    throw std::runtime_error("json: value is not a bool");
  }
  const std::string &as_string() const {
    if (type != Type::String) throw std::runtime_error("json: value is not a string");
    return str;
  }

 private:
  const char *tokz(char *buf, size_t cap) const {
    size_t n = toklen < cap - 1 ? toklen : cap - 1;
    memcpy(buf, tok, n);
    buf[n] = 0;
    return buf;
  }
  // XGBoost writes some numbers as strings ("5E-1", "[5E-1]", "127").
  float parse_string_float() const {
    std::string s = str;
    if (!s.empty() && s.front() == '[') s = s.substr(1);
    if (!s.empty() && s.back() == ']') s.pop_back();
    return strtof(s.c_str(), nullptr);
  }
  double parse_string_number() const {
    std::string s = str;
    if (!s.empty() && s.front() == '[') s = s.substr(1);
    if (!s.empty() && s.back() == ']') s.pop_back();
    return strtod(s.c_str(), nullptr);
  }
};

class Parser {
 public:
  Parser(const char *p, size_t n) : p_(p), end_(p + n) {}
  Value parse() {
    Value v;
    ws();
    value(v, 0);
    ws();
    return v;
  }

 private:
  const char *p_, *end_;
  [[noreturn]] void fail(const char *what) {
    throw std::runtime_error(std::string("json: ") + what);
  }
  void ws() {
    while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
  }
  void value(Value &v, int depth) {
    if (depth > 256) fail("nesting too deep");
    if (p_ >= end_) fail("unexpected end");
    char c = *p_;
    if (c == '{') {
      ++p_;
      v.type = Type::Object;
      ws();
      if (p_ < end_ && *p_ == '}') { ++p_; return; }
      for (;;) {
        ws();
        std::string k;
        string(k);
        ws();
        if (p_ >= end_ || *p_ != ':') fail("expected ':'");
        ++p_;
        ws();
        v.obj.emplace_back(std::move(k), Value());
        value(v.obj.back().second, depth + 1);
        ws();
        if (p_ >= end_) fail("unexpected end in object");
        if (*p_ == ',') { ++p_; continue; }
        if (*p_ == '}') { ++p_; return; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      ++p_;
      v.type = Type::Array;
      ws();
      if (p_ < end_ && *p_ == ']') { ++p_; return; }
      for (;;) {
        ws();
        v.arr.emplace_back();
        value(v.arr.back(), depth + 1);
        ws();
        if (p_ >= end_) fail("unexpected end in array");
        if (*p_ == ',') { ++p_; continue; }
        if (*p_ == ']') { ++p_; return; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"') {
      v.type = Type::String;
      string(v.str);
    } else if (c == 't' && end_ - p_ >= 4 && !memcmp(p_, "true", 4)) {
      v.type = Type::Bool; v.b = true; p_ += 4;
    } else if (c == 'f' && end_ - p_ >= 5 && !memcmp(p_, "false", 5)) {
      v.type = Type::Bool; v.b = false; p_ += 5;
    } else if (c == 'n' && end_ - p_ >= 4 && !memcmp(p_, "null", 4)) {
      v.type = Type::Null; p_ += 4;
    } else if (c == 'N' && end_ - p_ >= 3 && !memcmp(p_, "NaN", 3)) {
      // XGBoost / python json emit bare NaN / Infinity
      v.type = Type::Number; v.tok = p_; v.toklen = 3; p_ += 3;
    } else if (c == 'I' && end_ - p_ >= 8 && !memcmp(p_, "Infinity", 8)) {
      v.type = Type::Number; v.tok = p_; v.toklen = 8; p_ += 8;
    } else if (c == '-' || (c >= '0' && c <= '9')) {
      const char *s = p_;
      if (c == '-' && end_ - p_ >= 9 && !memcmp(p_, "-Infinity", 9)) {
        p_ += 9;
      } else {
        ++p_;
        while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' ||
                             *p_ == '+' || *p_ == '-'))
          ++p_;
      }
      v.type = Type::Number;
      v.tok = s;
      v.toklen = (uint32_t)(p_ - s);
    } else {
      fail("unexpected character");
    }
  }
  static void utf8(std::string &out, uint32_t cp) {
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
  uint32_t hex4() {
    if (end_ - p_ < 4) fail("bad \\u escape");
    uint32_t v = 0;
    for (int k = 0; k < 4; ++k) {
      char c = *p_++;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= c - '0';
      else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
      else fail("bad \\u escape");
    }
    return v;
  }
  void string(std::string &out) {
    if (p_ >= end_ || *p_ != '"') fail("expected string");
    ++p_;
    for (;;) {
      if (p_ >= end_) fail("unterminated string");
      char c = *p_++;
      if (c == '"') return;
      if (c != '\\') { out.push_back(c); continue; }
      if (p_ >= end_) fail("bad escape");
      char e = *p_++;
      switch (e) {
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case '/': out.push_back('/'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u': {
          uint32_t cp = hex4();
          if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
            p_ += 2;
            uint32_t lo = hex4();
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          utf8(out, cp);
          break;
        }
        default: fail("bad escape");
      }
    }
  }
};

// ---- UBJSON (the binary form XGBoost >= 1.6 can serialise boosters to) -------------------
class UbjParser {
 public:
  UbjParser(const uint8_t *p, size_t n) : p_(p), end_(p + n) {}
  Value parse() {
    Value v;
    value(v, next(), 0);
    return v;
  }

 private:
  const uint8_t *p_, *end_;
  [[noreturn]] void fail(const char *what) { throw std::runtime_error(std::string("ubjson: ") + what); }
  uint8_t next() {
    if (p_ >= end_) fail("unexpected end");
    return *p_++;
  }
  void need(size_t n) {
    if ((size_t)(end_ - p_) < n) fail("truncated");
  }
  template <typename T>
  T be() {
    need(sizeof(T));
    uint8_t b[sizeof(T)];
    for (size_t k = 0; k < sizeof(T); ++k) b[sizeof(T) - 1 - k] = p_[k];
    p_ += sizeof(T);
    T v;
    memcpy(&v, b, sizeof(T));
    return v;
  }
  int64_t integer(uint8_t m) {
    switch (m) {
      case 'i': return be<int8_t>();
      case 'U': return be<uint8_t>();
      case 'I': return be<int16_t>();
      case 'l': return be<int32_t>();
      case 'L': return be<int64_t>();
      default: fail("expected integer marker");
    }
  }
  void str(std::string &out) {
    int64_t n = integer(next());
    if (n < 0) fail("negative string length");
    need((size_t)n);
    out.assign((const char *)p_, (size_t)n);
    p_ += n;
  }
  void scalar(Value &v, uint8_t m) {
    v.type = Type::Number;
    v.is_bin = true;
    switch (m) {
      case 'i': case 'U': case 'I': case 'l': case 'L':
        v.bin_is_int = true; v.i = integer(m); break;
      case 'd': v.bin_is_f32 = true; v.f = be<float>(); break;
      case 'D': v.d = be<double>(); break;
      default: fail("bad scalar marker");
    }
  }
  void value(Value &v, uint8_t m, int depth) {
    if (depth > 256) fail("nesting too deep");
    switch (m) {
      case 'Z': v.type = Type::Null; return;
      case 'N': v.type = Type::Null; return;
      case 'T': v.type = Type::Bool; v.b = true; return;
      case 'F': v.type = Type::Bool; v.b = false; return;
      case 'i': case 'U': case 'I': case 'l': case 'L': case 'd': case 'D': scalar(v, m); return;
      case 'C': v.type = Type::String; v.str.assign(1, (char)next()); return;
      case 'S': v.type = Type::String; str(v.str); return;
      case 'H': v.type = Type::String; str(v.str); return;
      case '[': {
        v.type = Type::Array;
        uint8_t elem_type = 0;
        int64_t count = -1;
        if (p_ < end_ && *p_ == '$') { ++p_; elem_type = next(); }
        if (p_ < end_ && *p_ == '#') { ++p_; count = integer(next()); }
        if (elem_type && count < 0) fail("typed array without count");
        if (count >= 0) {
          // an element takes at least a byte of input (typed arrays of zero-byte markers are not something a model holds):
          // a declared count beyond the rest of the input is corrupt - and must not size an allocation
          if ((uint64_t)count > (uint64_t)(end_ - p_)) fail("array count exceeds the input");
          v.arr.resize((size_t)count);
          for (int64_t k = 0; k < count; ++k) value(v.arr[(size_t)k], elem_type ? elem_type : next(), depth + 1);
        } else {
          for (;;) {
            uint8_t mm = next();
            if (mm == ']') break;
            v.arr.emplace_back();
            value(v.arr.back(), mm, depth + 1);
          }
        }
        return;
      }
      case '{': {
        v.type = Type::Object;
        uint8_t elem_type = 0;
        int64_t count = -1;
        if (p_ < end_ && *p_ == '$') { ++p_; elem_type = next(); }
        if (p_ < end_ && *p_ == '#') { ++p_; count = integer(next()); }
        if (count >= 0) {
          for (int64_t k = 0; k < count; ++k) {
            std::string key;
            str(key);
            v.obj.emplace_back(std::move(key), Value());
            value(v.obj.back().second, elem_type ? elem_type : next(), depth + 1);
          }
        } else {
          for (;;) {
            if (p_ < end_ && *p_ == '}') { ++p_; break; }
            std::string key;
            str(key);
            v.obj.emplace_back(std::move(key), Value());
            value(v.obj.back().second, next(), depth + 1);
          }
        }
        return;
      }
      default: fail("unknown marker");
    }
  }
};

inline Value parse(const char *p, size_t n) { return Parser(p, n).parse(); }
inline Value parse_ubjson(const uint8_t *p, size_t n) { return UbjParser(p, n).parse(); }

}  // namespace json
}  // namespace mrk
