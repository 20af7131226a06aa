// CPU: the token -> count hash tables of the pre-pass (csrc/table_device.hpp) with a ONE-LANE stand-in for the wavefront
// primitives: the per-lane algorithm - insert, lookup, their list forms over tables probed by aligned bucket - gives the
// counts a std::map holds, for buckets of 2 / 4 entries (-DMRK_PROBE_W), tables from 8 entries up, from almost empty to
// over-full, tokens that collide, lookups of absent tokens, lanes that only ride along, and several host threads (one-lane
// wavefronts) inserting into one table at once.  What a wavefront adds - 64 lanes sharing one loop - is exercised by the GPU
// parity suites.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <thread>
#include <vector>

#define __device__
#define __forceinline__ inline
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return p ? 1ull : 0ull; }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long c, unsigned long long v) {
  __atomic_compare_exchange_n(p, &c, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return c;   // the value found there (== the expected one on success), like the device's atomicCAS
}
using std::min;

namespace buckets {
#include "table_device.hpp"
}
struct buckets_api {
  static constexpr const char *name = "buckets";
  static bool add(unsigned long long *t, uint32_t cap, uint32_t tok, bool want) { return buckets::mrk::table_add(t, cap, tok, want); }
  static uint32_t get(const unsigned long long *t, uint32_t cap, uint32_t tok, bool want) { return buckets::mrk::table_get(t, cap, tok, want); }
  static double sum(const uint32_t *toks, const unsigned long long *t, uint32_t cap, uint32_t len, double c) { return buckets::mrk::table_sum_list(toks, t, cap, len, c); }
  static uint32_t add_list(const uint32_t *toks, unsigned long long *t, uint32_t cap, uint32_t len) { return buckets::mrk::table_add_list(toks, t, cap, len); }
};

static uint64_t rng_state = 0x243f6a8885a308d3ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

template <typename A>
static void run(long long &checks, long long &bad) {
  rng_state = 0x243f6a8885a308d3ull;
  const uint32_t caps[] = {8, 10, 12, 16, 38, 64, 100, 258, 1000, 4094};   // even, >= 8: what the host hands out
  for (uint32_t cap : caps) {
    const uint32_t usable = cap / MRK_PROBE_W * MRK_PROBE_W;
    for (int fill_pct : {0, 10, 50, 75, 90, 100, 130}) {
      for (int rep = 0; rep < 6; ++rep) {
        std::vector<unsigned long long> tab(cap, 0ull);
        std::map<uint32_t, uint32_t> ref;
        const uint32_t distinct = std::max<uint32_t>(fill_pct ? 1u : 0u, (uint32_t)((uint64_t)cap * fill_pct / 100));
        std::vector<uint32_t> universe;   // a small universe (neighbouring ids) or a wide one
        for (uint32_t i = 0; i < distinct; ++i) universe.push_back(rep % 2 ? 1u + i : 1u + (uint32_t)(rnd() % 0x7fffffffu));
        for (uint32_t i = 0; i < distinct * 3; ++i) {
          const uint32_t tok = universe[rnd() % universe.size()];
          const bool ok = i % 5 == 4 ? A::add_list(&tok, tab.data(), cap, 1u) == 0u : A::add(tab.data(), cap, tok, true);
          if (ok) ref[tok] += 1;   // refused: the table is full of other keys - the device flags the request (ST_TABLE_FULL)
          ++checks;
          if (!ok && ref.size() < usable) { ++bad; printf("%s: table_add refused with room left: cap %u keys %zu\n", A::name, cap, ref.size()); }
          if (!ok && ref.count(tok)) { ++bad; printf("%s: table_add refused a key it holds\n", A::name); }
        }
        (void)A::add(tab.data(), cap, 12345u, false);   // a lane that rides along inserts nothing
        size_t used = 0;   // every entry is a key of the reference with its count, once
        for (unsigned long long e : tab)
          if ((uint32_t)e) { ++used; if (!ref.count((uint32_t)e) || ref[(uint32_t)e] != ((uint32_t)(e >> 32) & 0x7fffffffu /* bit 63: the bucket's walked-past mark */)) { ++bad; printf("%s: entry mismatch\n", A::name); } }
        if (used != ref.size()) { ++bad; printf("%s: cap %u: %zu entries for %zu keys\n", A::name, cap, used, ref.size()); }
        std::vector<uint32_t> q;   // lookups: present keys, absent keys, riding lanes
        for (auto &kv : ref) q.push_back(kv.first);
        for (int i = 0; i < 64; ++i) q.push_back(1u + (uint32_t)(rnd() % 0x7fffffffu));
        for (size_t i = 0; i < q.size(); ++i) {
          const uint32_t tok = q[i], exp = ref.count(tok) ? ref[tok] : 0u;
          const uint32_t a = A::get(tab.data(), cap, tok, true), ride = A::get(tab.data(), cap, tok, false);
          checks += 2;
          if (a != exp || ride != 0u) { ++bad; if (bad < 20) printf("%s: cap %u fill %d: tok %u expected %u got %u (riding %u)\n", A::name, cap, fill_pct, tok, exp, a, ride); }
        }
        std::vector<uint32_t> list;   // the list form: sums in list order (integers: exact)
        for (int i = 0; i < 21; ++i) list.push_back(q[rnd() % q.size()]);
        for (uint32_t len : {0u, 1u, 7u, 8u, 9u, 21u}) {
          double exp = 0.5;
          for (uint32_t i = 0; i < len; ++i) exp += ref.count(list[i]) ? ref[list[i]] : 0u;
          const double s0 = A::sum(list.data(), tab.data(), cap, len, 0.5);
          ++checks;
          if (s0 != exp) { ++bad; if (bad < 20) printf("%s: sum_list len %u: %g expected %g\n", A::name, len, s0, exp); }
        }
      }
    }
  }
}

// Several host threads - each a one-lane wavefront - inserting into ONE table at the same time: what the wavefronts of a
// workgroup do in the pre-pass (compare-and-swap races for an empty entry, the same new key from two sides at once).  Afterwards every key's count is the number of its inserts.
template <typename A>
static void hammer(long long &checks, long long &bad) {
  int twice = 0;   // rounds that left a key in two entries: must not happen
  for (int round = 0; round < 300; ++round) {
    const uint32_t cap = (uint32_t[]){16, 24, 40, 64, 130}[round % 5];
    const int n_thr = 6, per = (int)(cap * 3 / 4 / n_thr) * 4;      // occurrences: up to 3 x the distinct keys the table is sized for
    const uint32_t universe = std::max(2u, cap * 3 / 4 - 2);          // distinct keys <= 75 % of the capacity - 2
    std::vector<unsigned long long> tab(cap, 0ull);
    std::vector<std::vector<uint32_t>> toks(n_thr);
    std::vector<int> refused(n_thr, 0);
    for (int t = 0; t < n_thr; ++t)
      for (int i = 0; i < per; ++i) toks[t].push_back(1000u + (uint32_t)(rnd() % universe));
    std::vector<std::thread> th;
    int go = 0;   // all threads start together: the inserts really overlap
    for (int t = 0; t < n_thr; ++t)
      th.emplace_back([&, t] {
        while (!__atomic_load_n(&go, __ATOMIC_ACQUIRE)) {}
        for (uint32_t tok : toks[t]) if (!A::add(tab.data(), cap, tok, true)) refused[t] += 1;
      });
    __atomic_store_n(&go, 1, __ATOMIC_RELEASE);
    for (auto &x : th) x.join();
    size_t used = 0;
    for (unsigned long long e : tab) used += (uint32_t)e ? 1 : 0;
    { std::map<uint32_t, uint32_t> d; for (auto &v : toks) for (uint32_t tok : v) d[tok] = 1; if (used > d.size()) ++twice; }
    std::map<uint32_t, uint32_t> ref;
    for (auto &v : toks) for (uint32_t tok : v) ref[tok] += 1;
    for (int t = 0; t < n_thr; ++t) { ++checks; if (refused[t]) { ++bad; printf("%s: concurrent insert refused (cap %u, %zu keys)\n", A::name, cap, ref.size()); } }
    for (auto &kv : ref) {
      ++checks;
      const uint32_t got = A::get(tab.data(), cap, kv.first, true);
      if (got != kv.second) { ++bad; if (bad < 20) printf("%s: concurrent inserts of %u: count %u, expected %u (cap %u)\n", A::name, kv.first, got, kv.second, cap); }
    }
  }
  printf("%s: concurrent rounds with a key in two entries: %d\n", A::name, twice);
  if (twice) { ++bad; printf("%s: a key twice in a table with one home per key\n", A::name); }
}

int main() {
  long long checks = 0, bad = 0;
  hammer<buckets_api>(checks, bad);
  run<buckets_api>(checks, bad);
  printf("PROBE_W %d: %lld checks, %lld bad\n", (int)MRK_PROBE_W, checks, bad);
  return bad ? 1 : 0;
}
