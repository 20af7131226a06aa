// CPU: the token -> count hash tables of the pre-pass (csrc/table_device.hpp) with a ONE-LANE stand-in for the wavefront
// primitives: the per-lane algorithm of every variant - the default lookup, MRK_LEAN_GET (bookkeeping per window; relies on
// "no empty entry precedes a key on its probe sequence"), MRK_GET_PAIR (two home windows per trip) and both together - gives
// the counts a std::map holds, for window widths 2 / 3 / 4 / 8 (-DMRK_PROBE_W), tables from 8 entries up, from almost empty
// to completely full, tokens that collide, lookups of absent tokens, lanes that only ride along; and MRK_TABLE_BUCKETS (the
// tables probed by aligned bucket; widths 2 / 4) and MRK_TABLE_2CHOICE (two home buckets per key), each variant building its own tables.  What a wavefront adds -
// 64 lanes sharing one loop - is exercised by the GPU parity suites.
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
static inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long c, unsigned long long v) {
  __atomic_compare_exchange_n(p, &c, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return c;   // the value found there (== the expected one on success), like the device's atomicCAS
}
using std::min;

// one namespace per variant (the header has no `#pragma once`); VARIANT wraps its entry points in a struct the checks template over
#define VARIANT(NS, PAIR)                                                                                                          \
  struct NS##_api {                                                                                                                  \
    static constexpr bool has_pair = PAIR;                                                                                           \
    static constexpr bool two_homes = sizeof(#NS) == sizeof("two_choice");                                                            \
    static constexpr const char *name = #NS;                                                                                         \
    static bool add(unsigned long long *t, uint32_t cap, uint32_t tok, bool want) { return NS::mrk::table_add(t, cap, tok, want); } \
    static uint32_t get(const unsigned long long *t, uint32_t cap, uint32_t tok, bool want) { return NS::mrk::table_get(t, cap, tok, want); } \
    static double sum(const uint32_t *toks, const unsigned long long *t, uint32_t cap, uint32_t len, double c) { return NS::mrk::table_sum_list(toks, t, cap, len, c); } \
    static uint32_t add_list(const uint32_t *toks, unsigned long long *t, uint32_t cap, uint32_t len) { return NS::mrk::table_add_list(toks, t, cap, len); } \
  };

namespace plain {
#include "table_device.hpp"
}
#undef MRK_TABLE_DEVICE_HPP
#define MRK_LEAN_GET 1
namespace lean {
#include "table_device.hpp"
}
#undef MRK_TABLE_DEVICE_HPP
#define MRK_GET_PAIR 1
namespace lean_pair {
#include "table_device.hpp"
}
#undef MRK_TABLE_DEVICE_HPP
#undef MRK_LEAN_GET
namespace pair_only {
#include "table_device.hpp"
}
#undef MRK_TABLE_DEVICE_HPP
#undef MRK_GET_PAIR
#if MRK_PROBE_W == 2 || MRK_PROBE_W == 4
#define HAVE_BUCKETS 1
#define MRK_TABLE_BUCKETS 1
namespace buckets {
#include "table_device.hpp"
}
#undef MRK_TABLE_DEVICE_HPP
#define MRK_GET_PAIR 1
namespace buckets_pair {
#include "table_device.hpp"
}
#undef MRK_TABLE_DEVICE_HPP
#undef MRK_GET_PAIR
#define MRK_TABLE_2CHOICE 1
namespace two_choice {
#include "table_device.hpp"
}
#endif
VARIANT(plain, false)
VARIANT(lean, false)
VARIANT(lean_pair, true)
VARIANT(pair_only, true)
#ifdef HAVE_BUCKETS
VARIANT(buckets, false)
VARIANT(buckets_pair, true)
VARIANT(two_choice, false)
#endif
template <typename NSAPI> struct Pair;   // table_get2 exists only in the pair variants
#define PAIR_OF(NS) template <> struct Pair<NS##_api> { static void get2(const unsigned long long *t, uint32_t cap, uint32_t a, bool wa, uint32_t b, bool wb, uint32_t &ra, uint32_t &rb) { NS::mrk::table_get2(t, cap, a, wa, b, wb, ra, rb); } };
PAIR_OF(lean_pair)
PAIR_OF(pair_only)
#ifdef HAVE_BUCKETS
PAIR_OF(buckets_pair)
#endif

static uint64_t rng_state = 0x243f6a8885a308d3ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

template <typename A>
static void run(long long &checks, long long &bad, bool bucketed) {
  rng_state = 0x243f6a8885a308d3ull;
  const uint32_t caps[] = {8, 10, 12, 16, 38, 64, 100, 258, 1000, 4094};   // even, >= 8: what the host hands out
  for (uint32_t cap : caps) {
    const uint32_t usable = bucketed ? cap / MRK_PROBE_W * MRK_PROBE_W : cap;
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
          if ((uint32_t)e) { ++used; if (!ref.count((uint32_t)e) || ref[(uint32_t)e] != (uint32_t)(e >> 32)) { ++bad; printf("%s: entry mismatch\n", A::name); } }
        if (used != ref.size()) { ++bad; printf("%s: cap %u: %zu entries for %zu keys\n", A::name, cap, used, ref.size()); }
        std::vector<uint32_t> q;   // lookups: present keys, absent keys, riding lanes
        for (auto &kv : ref) q.push_back(kv.first);
        for (int i = 0; i < 64; ++i) q.push_back(1u + (uint32_t)(rnd() % 0x7fffffffu));
        for (size_t i = 0; i < q.size(); ++i) {
          const uint32_t tok = q[i], exp = ref.count(tok) ? ref[tok] : 0u;
          const uint32_t a = A::get(tab.data(), cap, tok, true), ride = A::get(tab.data(), cap, tok, false);
          checks += 2;
          if (a != exp || ride != 0u) { ++bad; if (bad < 20) printf("%s: cap %u fill %d: tok %u expected %u got %u (riding %u)\n", A::name, cap, fill_pct, tok, exp, a, ride); }
          if constexpr (A::has_pair) {
            const uint32_t tok2 = q[(i * 7 + 3) % q.size()], exp2 = ref.count(tok2) ? ref[tok2] : 0u;
            const bool w2 = i % 3 != 0;
            uint32_t c0, c1;
            Pair<A>::get2(tab.data(), cap, tok, true, tok2, w2, c0, c1);
            checks += 2;
            if (c0 != exp || c1 != (w2 ? exp2 : 0u)) { ++bad; if (bad < 20) printf("%s: pair %u/%u expected %u/%u\n", A::name, c0, c1, exp, w2 ? exp2 : 0u); }
          }
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
// workgroup do in the pre-pass (compare-and-swap races for an empty entry, the same new key from two sides at once, and - two
// home buckets - a key ending up in both).  Afterwards every key's count is the number of its inserts.
template <typename A>
static void hammer(long long &checks, long long &bad) {
  int twice = 0;   // rounds that left a key in two entries (two home buckets only)
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
  if (twice && !A::two_homes) { ++bad; printf("%s: a key twice in a table with one home per key\n", A::name); }
}

int main() {
  long long checks = 0, bad = 0;
  hammer<plain_api>(checks, bad);
  hammer<lean_api>(checks, bad);
#ifdef HAVE_BUCKETS
  hammer<buckets_api>(checks, bad);
  hammer<two_choice_api>(checks, bad);
#endif
  run<plain_api>(checks, bad, false);
  run<lean_api>(checks, bad, false);
  run<lean_pair_api>(checks, bad, false);
  run<pair_only_api>(checks, bad, false);
#ifdef HAVE_BUCKETS
  run<buckets_api>(checks, bad, true);
  run<buckets_pair_api>(checks, bad, true);
  run<two_choice_api>(checks, bad, true);
#endif
#ifdef HAVE_BUCKETS
  {  // what two lanes inserting the same NEW key at the same moment can leave behind: the key in both of its buckets
    const uint32_t cap = 64, nb = cap / MRK_PROBE_W, tok = 4242u;
    std::vector<unsigned long long> tab(cap, 0ull);
    const uint32_t ba = two_choice::mrk::tok_home(tok, nb), bb = two_choice::mrk::tok_alt(tok, ba, nb);
    tab[ba * MRK_PROBE_W + 1] = (unsigned long long)tok | (3ull << 32);
    tab[bb * MRK_PROBE_W + 0] = (unsigned long long)tok | (2ull << 32);
    checks += 3;
    if (ba == bb) { ++bad; printf("two_choice: the second bucket is the first\n"); }
    if (two_choice::mrk::table_get(tab.data(), cap, tok, true) != 5u) { ++bad; printf("two_choice: a key in both buckets is not summed\n"); }
    two_choice::mrk::table_add(tab.data(), cap, tok, true);
    if (two_choice::mrk::table_get(tab.data(), cap, tok, true) != 6u) { ++bad; printf("two_choice: insert of a key in both buckets\n"); }
    for (uint32_t t = 1; t < 5000; ++t) {   // the second bucket is never the first, for any table size
      for (uint32_t n : {2u, 3u, 5u, 16u, 50u, 1023u}) {
        const uint32_t a = two_choice::mrk::tok_home(t, n), b = two_choice::mrk::tok_alt(t, a, n);
        ++checks;
        if (a >= n || b >= n || a == b) { ++bad; printf("two_choice: buckets %u %u of %u\n", a, b, n); }
      }
    }
  }
#endif
  printf("PROBE_W %d: %lld checks, %lld bad\n", (int)MRK_PROBE_W, checks, bad);
  return bad ? 1 : 0;
}
