// Bulk load from Metarank's binary FeatureValue wire format: the bytes the reference
// keeps in the `values` store (Redis / RocksDB / MapDB), decoded straight into the device store's host
// mirror.  Reference writer/reader: fstore/codec/impl/FeatureValueCodec.scala:40-236 (tags 0-13, KeyCodec,
// ScopeCodec, PeriodicValueCodec), ScalarCodec.scala, TimeValueCodec.scala, ListCodec / ArrayCodec / MapCodec
// (varint size + elements), util/VarNum.java (unsigned LEB128), java.io.DataOutput (big-endian, writeUTF =
// u16 length + modified UTF-8).  Records are self-delimiting, so a blob is any concatenation of them.
#include <cstring>
#include <memory>
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

// The key a record carries (FeatureValueCodec.KeyCodec / ScopeCodec :205-235), kept structured: ids and field values may
// hold any bytes ('/' included), so they never travel through the Key.encode string form.  `id` is what follows
// "<kind>=" in ScopeCodec.encode (fstore/codec/impl/ScopeCodec.scala:18-26).
struct WireKey {
  ScopeId scope = SC_GLOBAL;
  std::string id, feature;
  KeyRef ref() const { return KeyRef{scope, id, feature}; }
};
WireKey read_key(In &in) {
  WireKey k;
  switch (in.byte()) {
    case 0: k.scope = SC_USER; k.id = in.utf(); break;
    case 1: k.scope = SC_ITEM; k.id = in.utf(); break;
    case 2: k.scope = SC_GLOBAL; break;
    case 3: k.scope = SC_SESSION; k.id = in.utf(); break;
    case 4: { std::string f = in.utf(); k.scope = SC_FIELD; k.id = f + ":" + in.utf(); break; }
    case 5: { std::string f = in.utf(), v = in.utf(); k.scope = SC_IRF; k.id = f + ":" + v + ":" + in.utf(); break; }
    case 6: k.scope = SC_RANKING; k.id = in.utf(); break;
    default: throw StatusError(MRK_ERR_PARSE, "feature value blob: cannot parse scope index");
  }
  k.feature = in.utf();
  return k;
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

// decodes every record of the blob into `store`; returns the number of records seen (records of features the
// configuration does not use are decoded and dropped, like KVStore values nobody reads)
// now_ms >= 0: every applied record's deadline = now_ms + its `expire` is remembered (Store::ttl_note)
int load_feature_values(Store &store, const uint8_t *bytes, size_t len, int64_t now_ms) {
  In in{bytes, bytes + len};
  int n = 0;
  while (in.p < in.end) {
    const uint8_t tag = in.byte();
    const bool has_ttl = tag >= 7;  // tags 0-6 are the pre-ttl encodings ("compat")
    if (tag > 13) throw StatusError(MRK_ERR_PARSE, "cannot decode fv index " + std::to_string(tag));
    const WireKey wk = read_key(in);
    const KeyRef key = wk.ref();
    (void)in.var_long();  // timestamp: the read path does not look at it
    switch (has_ttl ? tag - 7 : tag) {
      case 0: {  // ScalarValue
        ScalarV v = read_scalar(in);
        switch (v.kind) {
          case 0: store.put_string(key, v.s); break;
          case 1: store.put_double(key, v.d); break;
          case 2: store.put_bool(key, v.b); break;
          case 3: {
            std::vector<std::string_view> ptrs(v.sl.begin(), v.sl.end());
            store.put_string_list(key, ptrs.data(), (int)ptrs.size());
            break;
          }
          default: store.put_double_list(key, v.dl.data(), (int)v.dl.size()); break;
        }
        break;
      }
      case 1: store.put_counter(key, in.var_long()); break;  // CounterValue
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
        store.put_periodic(key, vals.data(), (int)vals.size());
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
        std::vector<std::string_view> ptrs(ids.begin(), ids.end());
        store.put_bounded_list(key, ptrs.data(), (int)ptrs.size());
        break;
      }
    }
    int64_t expire_ms = 90ll * 86400 * 1000;   // the pre-ttl encodings decode to 90 days (FeatureValueCodec.scala:66)
    if (has_ttl) expire_ms = in.var_long();
    if (now_ms >= 0 && expire_ms > 0 && expire_ms < (1ll << 53)) store.ttl_note(key, now_ms + expire_ms);
    ++n;
  }
  return n;
}

// ---------------------------------------------------------------------------------------------------------
// RankingEventFormat (util/RankingEventFormat.scala:12-62): the binary form of a `ranking` event the reference
// uses for the warm-up requests stored in the model container (ml/rank/LambdaMARTRanker.scala:223,387):
//   UTF id, i64 ts, [bool + UTF] user, [bool + UTF] session, i32 n + fields, i32 n + items {UTF id, i32 n + fields}
//   field: u8 tag (0 string, 1 boolean, 2 number, 3 string list, 4 number list), UTF name, value
// (DataOutputStream: big-endian, fixed-width ints - not the varints of the FeatureValue codec).
namespace {
struct Fixed {
  In in;
  int64_t i64() { in.need(8); uint64_t u = 0; for (int i = 0; i < 8; ++i) u = (u << 8) | in.p[i]; in.p += 8; return (int64_t)u; }
  int32_t i32() { in.need(4); uint32_t u = 0; for (int i = 0; i < 4; ++i) u = (u << 8) | in.p[i]; in.p += 4; return (int32_t)u; }
};
}  // namespace

static void read_field(Fixed &f, mrk_field &out, std::vector<std::unique_ptr<std::string>> &strs,
                       std::vector<std::unique_ptr<std::vector<const char *>>> &lists, std::vector<std::unique_ptr<std::vector<double>>> &nums) {
  auto keep = [&](std::string s) { strs.emplace_back(new std::string(std::move(s))); return strs.back()->c_str(); };
  memset(&out, 0, sizeof(out));
  const uint8_t tag = f.in.byte();
  out.name = keep(f.in.utf());
  switch (tag) {
    case 0: out.type = MRK_FIELD_STRING; out.str = keep(f.in.utf()); break;
    case 1: out.type = MRK_FIELD_BOOL; out.num = f.in.byte() ? 1.0 : 0.0; break;
    case 2: out.type = MRK_FIELD_NUMBER; out.num = f.in.f64(); break;
    case 3: {
      const int n = f.i32();
      if (n < 0) throw StatusError(MRK_ERR_PARSE, "ranking event: negative list size");
      lists.emplace_back(new std::vector<const char *>());
      for (int i = 0; i < n; ++i) lists.back()->push_back(keep(f.in.utf()));
      out.type = MRK_FIELD_STRING_LIST;
      out.n = n;
      out.strs = lists.back()->data();
      break;
    }
    case 4: {
      const int n = f.i32();
      if (n < 0) throw StatusError(MRK_ERR_PARSE, "ranking event: negative list size");
      nums.emplace_back(new std::vector<double>());
      for (int i = 0; i < n; ++i) nums.back()->push_back(f.in.f64());
      out.type = MRK_FIELD_NUMBER_LIST;
      out.n = n;
      out.nums = nums.back()->data();
      break;
    }
    default: throw StatusError(MRK_ERR_PARSE, "ranking event: unknown field tag");
  }
}

size_t DecodedRequest::decode(const uint8_t *bytes, size_t len) {
  Fixed f{In{bytes, bytes + len}};
  auto keep = [&](std::string s) { strs.emplace_back(new std::string(std::move(s))); return strs.back()->c_str(); };
  memset(&req, 0, sizeof(req));
  req.id = keep(f.in.utf());
  req.timestamp_ms = f.i64();
  req.user = f.in.byte() ? keep(f.in.utf()) : nullptr;
  req.session = f.in.byte() ? keep(f.in.utf()) : nullptr;
  const int nf = f.i32();
  if (nf < 0) throw StatusError(MRK_ERR_PARSE, "ranking event: negative field count");
  // a field is at least a name length, a tag and a payload byte: a count beyond the rest of the input is corrupt and must
  // not size an allocation
  if ((size_t)nf > (size_t)(f.in.end - f.in.p)) throw StatusError(MRK_ERR_PARSE, "ranking event: field count exceeds the input");
  fields.resize(nf);
  for (int i = 0; i < nf; ++i) read_field(f, fields[i], strs, lists, nums);
  const int ni = f.i32();
  if (ni < 0) throw StatusError(MRK_ERR_PARSE, "ranking event: negative item count");
  item_ids.clear();
  item_offsets.assign(1, 0);
  item_fields.clear();
  for (int i = 0; i < ni; ++i) {
    item_ids.push_back(keep(f.in.utf()));
    const int k = f.i32();
    if (k < 0) throw StatusError(MRK_ERR_PARSE, "ranking event: negative field count");
    for (int j = 0; j < k; ++j) {
      item_fields.emplace_back();
      read_field(f, item_fields.back(), strs, lists, nums);
    }
    item_offsets.push_back((int32_t)item_fields.size());
  }
  req.fields = fields.data();
  req.n_fields = nf;
  req.n_items = ni;
  req.item_ids = item_ids.data();
  req.item_field_offsets = item_fields.empty() ? nullptr : item_offsets.data();
  req.item_fields = item_fields.empty() ? nullptr : item_fields.data();
  return (size_t)(f.in.p - bytes);
}

}  // namespace mrk
