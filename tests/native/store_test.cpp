// CPU unit test of the feature store's host mirror (csrc/store.cpp): string lists in the records' inline heaps and in
// the token pool, double lists and bounded lists in their pools - under heavy replacement.  Every value read back from
// the mirror (the bytes the device would gather) equals a reference map, and the pools stay bounded: replaced values are
// recycled through the size-class free lists instead of being appended for ever.  Links libmrk_hip.so; no device call.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "features.hpp"
#include "store.hpp"

using namespace mrk;

static unsigned long long rng_state = 0x9e3779b97f4a7c15ull;
static unsigned long long rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main() {
  const char *cfg = R"({"features": [
    {"name": "genre", "type": "string", "scope": "item", "source": "item.genre", "values": ["a", "b", "c"], "encode": "index"},
    {"name": "tags", "type": "string", "scope": "item", "source": "item.tags", "values": ["a", "b"], "encode": "onehot"},
    {"name": "vec", "type": "vector", "scope": "item", "source": "item.vec", "reduce": ["vector8"]},
    {"name": "pop", "type": "number", "scope": "item", "source": "item.pop"},
    {"name": "profile", "type": "interacted_with", "scope": "session", "interaction": "click", "field": ["item.genres", "item.actors"]}],
    "models": {"m": {"type": "lambdamart", "features": ["genre", "tags", "vec", "pop", "profile"]}}})";
  Store st;
  std::unique_ptr<Registry> reg = load_config(cfg, strlen(cfg), st, false);
  const Table &items = st.tables[SC_ITEM];
  printf("item record: stride %u, heap at %u, %u bytes\n", items.stride, items.heap_off, items.heap_cap);
  if (items.stride % 64 != 0 || items.heap_off + items.heap_cap > items.stride) { printf("bad layout\n"); return 1; }

  const int N_ITEMS = 300, N_SESS = 40;
  const char *list_cols[] = {"genre", "tags", "profile_genres", "profile_actors"};
  std::map<std::string, std::vector<std::string>> ref_lists;   // "item=<i>/<col>" -> tokens
  std::map<std::string, std::vector<double>> ref_vecs;
  std::map<std::string, std::vector<std::string>> ref_sessions;
  std::map<std::string, bool> ref_is_double;                   // a list column overwritten by a number
  int bad = 0;
  auto token = [] { return "t" + std::to_string(rnd() % 500); };
  size_t tok_hi = 0, f64_hi = 0, slot_hi = 0;
  for (int round = 0; round < 60000; ++round) {
    const std::string item = std::to_string(rnd() % N_ITEMS);
    switch (rnd() % 8) {
      case 0: case 1: case 2: case 3: {  // a string list of 0..40 tokens (longer ones overflow the heap into the pool)
        const std::string key = "item=" + item + "/" + list_cols[rnd() % 4];
        const int n = rnd() % 5 == 0 ? (int)(rnd() % 41) : (int)(rnd() % 6);
        std::vector<std::string> v;
        for (int i = 0; i < n; ++i) v.push_back(token());
        std::vector<const char *> p;
        for (auto &s : v) p.push_back(s.c_str());
        st.put_string_list(key.c_str(), p.data(), n);
        ref_lists[key] = v;
        ref_is_double[key] = false;
        break;
      }
      case 4: {  // a double list whose length changes
        const std::string key = "item=" + item + "/vec";
        // short lists stay f64; long ones whose values are all exactly floats (x / 8) go to the f32 pool, the others (x / 7) do not
        std::vector<double> v((size_t)(rnd() % 3 == 0 ? rnd() % 41 : rnd() % 12));
        const double div = rnd() % 2 ? 8.0 : 7.0;
        for (auto &x : v) x = (double)(rnd() % 1000) / div;
        st.put_double_list(key.c_str(), v.data(), (int)v.size());
        ref_vecs[key] = v;
        break;
      }
      case 5: {  // a session's bounded list through the write path (prepend, keep `count`)
        const std::string key = "session=s" + std::to_string(rnd() % N_SESS) + "/profile_interactions";
        auto &l = ref_sessions[key];
        l.insert(l.begin(), item);
        if (l.size() > 100) l.resize(100);
        st.append(key.c_str(), item.c_str(), 1000 + round);
        break;
      }
      case 6: {  // a number put over a column that may hold a list, or an erase
        const std::string key = "item=" + item + "/" + list_cols[rnd() % 4];
        if (rnd() % 2) { st.put_double(key.c_str(), 1.5); ref_is_double[key] = true; ref_lists.erase(key); }
        else { st.erase(key.c_str()); ref_is_double[key] = false; ref_lists.erase(key); }
        break;
      }
      default: st.put_double(("item=" + item + "/pop").c_str(), (double)round); break;
    }
    if (round == 20000) { tok_hi = st.tok_pool.host.size(); f64_hi = st.f64_pool.host.size() + st.f32_pool.host.size(); slot_hi = st.slot_pool.host.size(); }
  }
  // read back through the mirror exactly like the device does
  auto cell_of = [&](ScopeId sc, const std::string &id, const std::string &col, uint8_t &tag, uint64_t &bits, const uint8_t *&rec) {
    const Table &t = st.tables[sc];
    const uint32_t s = st.slot(sc, id, false);
    if (s == Store::NO_SLOT) { tag = TAG_MISSING; return; }
    rec = t.rows.data() + (size_t)s * t.stride;
    const Column &c = t.cols[(size_t)t.col_of.at(col)];
    tag = rec[c.tag_index];
    memcpy(&bits, rec + c.val_off, 8);
  };
  for (auto &kv : ref_lists) {
    const std::string id = kv.first.substr(5, kv.first.find('/') - 5), col = kv.first.substr(kv.first.find('/') + 1);
    uint8_t tag = 0; uint64_t bits = 0; const uint8_t *rec = nullptr;
    cell_of(SC_ITEM, id, col, tag, bits, rec);
    if (tag != TAG_STRING_LIST || (uint32_t)(bits >> 32) != kv.second.size()) { ++bad; continue; }
    const uint32_t off = (uint32_t)bits;
    const uint32_t *toks = (off & LIST_INLINE) ? (const uint32_t *)(rec + (off & ~LIST_INLINE)) : st.tok_pool.host.data() + off;
    if ((off & LIST_INLINE) && ((off & ~LIST_INLINE) < items.heap_off || (off & ~LIST_INLINE) + 4 * kv.second.size() > items.stride)) ++bad;
    for (size_t i = 0; i < kv.second.size(); ++i)
      if (toks[i] != st.find_token(kv.second[i])) { ++bad; break; }
  }
  for (auto &kv : ref_is_double) {
    if (!kv.second) continue;
    const std::string id = kv.first.substr(5, kv.first.find('/') - 5), col = kv.first.substr(kv.first.find('/') + 1);
    uint8_t tag = 0; uint64_t bits = 0; const uint8_t *rec = nullptr;
    cell_of(SC_ITEM, id, col, tag, bits, rec);
    if (tag != TAG_DOUBLE) ++bad;
  }
  for (auto &kv : ref_vecs) {
    const std::string id = kv.first.substr(5, kv.first.find('/') - 5);
    uint8_t tag = 0; uint64_t bits = 0; const uint8_t *rec = nullptr;
    cell_of(SC_ITEM, id, "vec", tag, bits, rec);
    if (tag != TAG_DOUBLE_LIST || (uint32_t)(bits >> 32) != kv.second.size()) { ++bad; continue; }
    const uint32_t off = (uint32_t)bits;
    bool exact = kv.second.size() >= LIST_F32_MIN;
    for (double x : kv.second) exact = exact && (double)(float)x == x;
    if (((off & LIST_F32_BIT) != 0) != exact && !kv.second.empty()) ++bad;   // the right pool
    if ((off & LIST_F32_BIT) && (off & 3u)) ++bad;                           // 16-byte aligned for float4 loads
    for (size_t i = 0; i < kv.second.size(); ++i) {
      const double got = (off & LIST_F32_BIT) ? (double)st.f32_pool.host[(off & ~LIST_F32_BIT) + i] : st.f64_pool.host[off + i];
      if (got != kv.second[i]) { ++bad; break; }
    }
  }
  for (auto &kv : ref_sessions) {
    const std::string id = kv.first.substr(8, kv.first.find('/') - 8);
    uint8_t tag = 0; uint64_t bits = 0; const uint8_t *rec = nullptr;
    cell_of(SC_SESSION, id, "profile_interactions", tag, bits, rec);
    if (tag != TAG_PRESENT || (uint32_t)(bits >> 32) != kv.second.size()) { ++bad; continue; }
    for (size_t i = 0; i < kv.second.size(); ++i)
      if (st.slot_pool.host[(uint32_t)bits + i] != st.slot(SC_ITEM, kv.second[i], false)) { ++bad; break; }
  }
  // ---- FeatureValue.expire (Store::ttl_note / ttl_expire): a value is dropped `expire` after its LAST write, a rewrite moves the
  // deadline, a plain put clears it, unknown keys are no-ops
  {
    auto tag_of = [&](const char *id, const char *col) { uint8_t tag = 0; uint64_t bits = 0; const uint8_t *rec = nullptr; cell_of(SC_ITEM, id, col, tag, bits, rec); return tag; };
    const int64_t T0 = 1700000000000ll;
    st.put_double("item=ttl1/pop", 1.0); st.ttl_note("item=ttl1/pop", T0 + 1000);
    st.put_double("item=ttl2/pop", 2.0); st.ttl_note("item=ttl2/pop", T0 + 5000);
    st.put_double("item=ttl3/pop", 3.0); st.ttl_note("item=ttl3/pop", T0 + 1000);
    { const char *g[1] = {"a"}; st.put_string_list("item=ttl1/genre", g, 1); st.ttl_note("item=ttl1/genre", T0 + 2000); }
    st.ttl_note("item=nobody/pop", T0 + 1);          // no such slot
    st.ttl_note("item=ttl1/unknown_feature", T0 + 1);
    if (st.ttl_tracked() != 4) { printf("ttl: tracked %zu\n", st.ttl_tracked()); ++bad; }
    if (st.ttl_expire(T0 + 999) != 0) ++bad;
    st.put_double("item=ttl3/pop", 3.5);                          // a plain put: its deadline is void
    st.put_double("item=ttl2/pop", 2.5); st.ttl_note("item=ttl2/pop", T0 + 9000);   // rewritten: the deadline moves
    if (st.ttl_expire(T0 + 1000) != 1 || tag_of("ttl1", "pop") != TAG_MISSING || tag_of("ttl3", "pop") != TAG_DOUBLE || tag_of("ttl1", "genre") == TAG_MISSING) { printf("ttl: first sweep\n"); ++bad; }
    if (st.ttl_expire(T0 + 6000) != 1 || tag_of("ttl1", "genre") != TAG_MISSING || tag_of("ttl2", "pop") != TAG_DOUBLE) { printf("ttl: second sweep\n"); ++bad; }
    if (st.ttl_expire(T0 + 9000) != 1 || tag_of("ttl2", "pop") != TAG_MISSING || st.ttl_tracked() != 0) { printf("ttl: third sweep\n"); ++bad; }
    if (st.ttl_expire(T0 + 99000) != 0) ++bad;
  }
  // two thirds of the churn happened after the high-water marks were taken: a store that appended for ever would have
  // tripled; recycling keeps the growth small
  const size_t tok_end = st.tok_pool.host.size(), f64_end = st.f64_pool.host.size() + st.f32_pool.host.size(), slot_end = st.slot_pool.host.size();
  printf("pools after 20000 / 60000 rounds: tokens %zu / %zu, doubles %zu / %zu, slots %zu / %zu\n", tok_hi, tok_end, f64_hi, f64_end, slot_hi, slot_end);
  const bool bounded = tok_end <= tok_hi * 3 / 2 + 256 && f64_end <= f64_hi * 3 / 2 + 256 && slot_end <= slot_hi * 3 / 2 + 512;
  printf("lists %zu vecs %zu sessions %zu bad %d bounded %d\n", ref_lists.size(), ref_vecs.size(), ref_sessions.size(), bad, (int)bounded);
  return bad == 0 && bounded ? 0 : 1;
}
