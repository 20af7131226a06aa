// CPU unit test of SlotMap (csrc/store.hpp): the id -> slot map every request's item ids go through.  An id that was
// never put must never resolve to a slot, whatever its hash; ids are arbitrary byte strings without NUL.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

#include "store.hpp"

using mrk::SlotMap;

static unsigned long long rng_state = 88172645463325252ull;
static unsigned long long rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main() {
  SlotMap m;
  std::unordered_map<std::string, uint32_t> ref;
  std::vector<std::string> ids;
  auto add = [&](const std::string &s) {
    if (ref.count(s)) return;
    const uint32_t slot = (uint32_t)ids.size();
    m.insert(s.data(), s.size(), slot);
    ref.emplace(s, slot);
    ids.push_back(s);
  };
  // empty id, every length around the 8-byte steps of the hash, shared prefixes / suffixes, bytes >= 0x80
  add("");
  for (int len = 1; len <= 40; ++len) {
    add(std::string((size_t)len, 'a'));
    add(std::string((size_t)len - 1, 'a') + "b");
    add("b" + std::string((size_t)len - 1, 'a'));
    add(std::string((size_t)len, (char)0xe9));
  }
  for (int i = 0; i < 200000; ++i) {  // growth through many doublings; numeric ids like the benchmark's
    add(std::to_string(i));
    if (i % 7 == 0) add("item-" + std::to_string(rnd() % 1000003) + "-" + std::string((size_t)(rnd() % 30), 'x'));
  }
  int bad = 0;
  for (size_t s = 0; s < ids.size(); ++s)
    if (m.find(ids[s].data(), ids[s].size()) != (uint32_t)s) ++bad;
  // absent ids: near misses of present ones and random strings
  long long absent = 0;
  for (int i = 0; i < 300000; ++i) {
    std::string q;
    switch (i % 4) {
      case 0: q = std::to_string(200000 + i); break;
      case 1: q = ids[rnd() % ids.size()] + "x"; break;
      case 2: q = ids[rnd() % ids.size()]; if (!q.empty()) q.pop_back(); else q = "?"; break;
      default: q = std::string((size_t)(rnd() % 24), (char)('a' + rnd() % 26)) + std::to_string(rnd()); break;
    }
    const bool present = ref.count(q) != 0;
    const uint32_t got = m.find(q.data(), q.size());
    if (present ? got != ref[q] : got != SlotMap::NONE) ++bad;
    absent += !present;
  }
  // the batched form used by resolve_requests: hash, prefetch, probe
  for (int i = 0; i < 1000; ++i) {
    const std::string &s = ids[rnd() % ids.size()];
    const uint64_t h = SlotMap::hash(s.data(), s.size());
    m.prefetch(h);
    if (m.find_hashed(h, s.data(), s.size()) != ref[s]) ++bad;
  }
  printf("ids %zu absent-queries %lld bad %d load %.3f\n", ids.size(), absent, bad, (double)m.n / (double)m.table.size());
  return bad == 0 && m.n == ids.size() && m.n * 2 <= m.table.size() ? 0 : 1;
}
