// Bulk load from Metarank's binary FeatureValue wire format: the bytes the reference
// keeps in the `values` store (Redis / RocksDB / MapDB), decoded straight into the device store's host
// mirror.  Reference writer/reader: fstore/codec/impl/FeatureValueCodec.scala:40-236 (tags 0-13, KeyCodec,
// ScopeCodec, PeriodicValueCodec), ScalarCodec.scala, TimeValueCodec.scala, ListCodec / ArrayCodec / MapCodec
// (varint size + elements), util/VarNum.java (unsigned LEB128), java.io.DataOutput (big-endian, writeUTF =
// u16 length + modified UTF-8).  Records are self-delimiting, so a blob is any concatenation of them.
#include <cstring>
#include <stdexcept>

#include "store.hpp"

namespace mrk {

namespace {

struct In {
  const uint8_t *p, *end;
  void need(size_t n) const {
    if ((size_t)(end - p) < n) throw StatusError(MRK_ERR_PARSE, "feature value blob: truncated");
  }
  uint8_t byte() { need(1); return *p++; }
  int64_t var_long() {  // VarNum.getVarLong
    uint64_t v = 0;
    int idx = 0;
    uint8_t b;
    do {
      b = byte();
      if (idx < 10) v |= (uint64_t)(b & 0x7f) << (7 * idx);
      ++idx;
    } while (b & 0x80);
    return (int64_t)v;
  }
  int32_t var_int() {  // VarNum.getVarInt: at most 5 payload bytes, further continuation bytes are skipped
    uint32_t v = 0;
    int idx = 0;
    uint8_t b;
    do {
      b = byte();
      if (idx < 5) v |= (uint32_t)(b & 0x7f) << (7 * idx);
      ++idx;
    } while (b & 0x80);
    return (int32_t)v;
  }
  double f64() {  // DataInput.readDouble: big-endian IEEE-754
    need(8);
    uint64_t u = 0;
    for (int i = 0; i < 8; ++i) u = (u << 8) | p[i];
    p += 8;
    double d;
    memcpy(&d, &u, 8);
    return d;
  }
  // DataInput.readUTF: u16 byte length + modified UTF-8.  Ids and feature names are returned as the raw bytes,
  // which equals standard UTF-8 for everything in the BMP except U+0000 (C0 80) - the same bytes the JVM host
  // would have to hand to the mrk_store_put_* entry points for that key.
  std::string utf() {
    need(2);
    size_t n = ((size_t)p[0] << 8) | p[1];
    p += 2;
    need(n);
    std::string s((const char *)p, n);
    p += n;
    return s;
  }
};

// Key.encode (model/Key.scala:9) of the key the record carries: "<ScopeCodec.encode(scope)>/<feature>"
std::string read_key(In &in) {
  std::string scope;
  switch (in.byte()) {  // FeatureValueCodec.ScopeCodec :205-235
    case 0: scope = "user=" + in.utf(); break;
    case 1: scope = "item=" + in.utf(); break;
    case 2: scope = "global"; break;
    case 3: scope = "session=" + in.utf(); break;
    case 4: { std::string f = in.utf(); scope = "field=" + f + ":" + in.utf(); break; }
    case 5: { std::string f = in.utf(), v = in.utf(); scope = "irf=" + f + ":" + v + ":" + in.utf(); break; }
    case 6: scope = "ranking=" + in.utf(); break;
    default: throw StatusError(MRK_ERR_PARSE, "feature value blob: cannot parse scope index");
  }
  return scope + "/" + in.utf();
}

struct ScalarV {
  int kind = -1;  // 0 string, 1 double, 2 bool, 3 string list, 4 double list
  std::string s;
  double d = 0;
  bool b = false;
  std::vector<std::string> sl;
  std::vector<double> dl;
};

ScalarV read_scalar(In &in) {  // ScalarCodec.read
  ScalarV v;
  v.kind = in.byte();
  switch (v.kind) {
    case 0: v.s = in.utf(); break;
    case 1: v.d = in.f64(); break;
    case 2: v.b = in.byte() != 0; break;
    case 3: { int n = in.var_int(); for (int i = 0; i < n; ++i) v.sl.push_back(in.utf()); break; }
    case 4: { int n = in.var_int(); for (int i = 0; i < n; ++i) v.dl.push_back(in.f64()); break; }
    default: throw StatusError(MRK_ERR_PARSE, "feature value blob: cannot decode scalar");
  }
  return v;
}

}  // namespace

// decodes every record of the blob into `store`; returns the number of records seen
int load_feature_values(Store &store, const uint8_t *bytes, size_t len) {
  In in{bytes, bytes + len};
  int n = 0;
  while (in.p < in.end) {
    const uint8_t tag = in.byte();
    const bool has_ttl = tag >= 7;  // tags 0-6 are the pre-ttl encodings ("compat")
    if (tag > 13) throw StatusError(MRK_ERR_PARSE, "cannot decode fv index " + std::to_string(tag));
    const std::string key = read_key(in);
    (void)in.var_long();  // timestamp: the read path does not look at it
    switch (has_ttl ? tag - 7 : tag) {
      case 0: {  // ScalarValue
        ScalarV v = read_scalar(in);
        switch (v.kind) {
          case 0: store.put_string(key.c_str(), v.s.c_str()); break;
          case 1: store.put_double(key.c_str(), v.d); break;
          case 2: store.put_bool(key.c_str(), v.b); break;
          case 3: {
            std::vector<const char *> ptrs;
            for (auto &s : v.sl) ptrs.push_back(s.c_str());
            store.put_string_list(key.c_str(), ptrs.data(), (int)ptrs.size());
            break;
          }
          default: store.put_double_list(key.c_str(), v.dl.data(), (int)v.dl.size()); break;
        }
        break;
      }
      case 1: store.put_counter(key.c_str(), in.var_long()); break;  // CounterValue
      case 2: {  // NumStatsValue: not read by /rank
        (void)in.f64(); (void)in.f64();
        int m = in.var_int();
        for (int i = 0; i < m; ++i) { (void)in.var_int(); (void)in.f64(); }
        break;
      }
      case 3: {  // MapValue: not read by /rank
        int m = in.var_int();
        for (int i = 0; i < m; ++i) { (void)in.utf(); (void)read_scalar(in); }
        break;
      }
      case 4: {  // PeriodicCounterValue: Array[PeriodicValue(start, end, periods, value)]
        int m = in.var_int();
        std::vector<int64_t> vals;
        for (int i = 0; i < m; ++i) {
          (void)in.var_long(); (void)in.var_long(); (void)in.var_int();
          vals.push_back(in.var_long());
        }
        store.put_periodic(key.c_str(), vals.data(), (int)vals.size());
        break;
      }
      case 5: {  // FrequencyValue: not read by /rank
        int m = in.var_int();
        for (int i = 0; i < m; ++i) { (void)in.utf(); (void)in.f64(); }
        break;
      }
      default: {  // 6: BoundedListValue: List[TimeValue(ts, scalar)]; /rank reads the SString elements (item ids)
        int m = in.var_int();
        std::vector<std::string> ids;
        for (int i = 0; i < m; ++i) {
          (void)in.var_long();
          ScalarV v = read_scalar(in);
          if (v.kind == 0) ids.push_back(v.s);  // InteractedWithFeature.scala:104-108 collects SString values only
        }
        std::vector<const char *> ptrs;
        for (auto &s : ids) ptrs.push_back(s.c_str());
        store.put_bounded_list(key.c_str(), ptrs.data(), (int)ptrs.size());
        break;
      }
    }
    if (has_ttl) (void)in.var_long();  // expire (ms)
    ++n;
  }
  return n;
}

}  // namespace mrk
