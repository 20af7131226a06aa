// CPU: the token -> count hash tables of the pre-pass (csrc/table_device.hpp) with a ONE-LANE stand-in for the wavefront
// primitives: the per-lane algorithm of every variant - the default lookup, MRK_LEAN_GET (bookkeeping per window; relies on
// "no empty entry precedes a key on its probe sequence"), MRK_GET_PAIR (two home windows per trip) and both together - gives
// the counts a std::map holds, for window widths 2 / 3 / 4 / 8 (-DMRK_PROBE_W), tables from 8 entries up, from almost empty
// to completely full, tokens that collide, lookups of absent tokens, lanes that only ride along.  What a wavefront adds -
// 64 lanes sharing one loop - is exercised by the GPU parity suites.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define __device__
#define __forceinline__ inline
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return p ? 1ull : 0ull; }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
static inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long c, unsigned long long v) { const unsigned long long o = *p; if (o == c) *p = v; return o; }
using std::min;

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

static uint64_t rng_state = 0x243f6a8885a308d3ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main() {
  long long checks = 0, bad = 0;
  const uint32_t caps[] = {8, 9, 11, 16, 37, 64, 100, 257, 1000, 4093};
  for (uint32_t cap : caps) {
    for (int fill_pct : {0, 10, 50, 75, 90, 100, 130}) {
      for (int rep = 0; rep < 6; ++rep) {
        std::vector<unsigned long long> tab(cap, 0ull);
        std::map<uint32_t, uint32_t> ref;
        const uint32_t distinct = std::max<uint32_t>(fill_pct ? 1u : 0u, (uint32_t)((uint64_t)cap * fill_pct / 100));
        // tokens from a small universe (collisions in the low bits of the hash) or a wide one
        std::vector<uint32_t> universe;
        for (uint32_t i = 0; i < distinct; ++i) universe.push_back(rep % 2 ? 1u + i : 1u + (uint32_t)(rnd() % 0x7fffffffu));
        bool overflowed = false;
        for (uint32_t i = 0; i < distinct * 3; ++i) {
          const uint32_t tok = universe[rnd() % universe.size()];
          const bool ok = plain::mrk::table_add(tab.data(), cap, tok, true);
          if (ok) ref[tok] += 1;
          else overflowed = true;   // the table is full of other keys: the device flags the request (ST_TABLE_FULL)
          ++checks;
          if (!ok && ref.size() < cap) { ++bad; printf("table_add refused with room left: cap %u keys %zu\n", cap, ref.size()); }
          if (!ok && ref.count(tok)) { ++bad; printf("table_add refused a key it holds\n"); }
        }
        (void)overflowed;
        (void)plain::mrk::table_add(tab.data(), cap, 12345u, false);   // a lane that rides along inserts nothing
        // every entry is a key of the reference with its count
        size_t used = 0;
        for (unsigned long long e : tab)
          if ((uint32_t)e) { ++used; if (!ref.count((uint32_t)e) || ref[(uint32_t)e] != (uint32_t)(e >> 32)) { ++bad; printf("entry mismatch\n"); } }
        if (used != ref.size()) { ++bad; printf("cap %u: %zu entries for %zu keys\n", cap, used, ref.size()); }
        // lookups: present keys, absent keys, riding lanes
        std::vector<uint32_t> q;
        for (auto &kv : ref) q.push_back(kv.first);
        for (int i = 0; i < 64; ++i) q.push_back(1u + (uint32_t)(rnd() % 0x7fffffffu));
        for (size_t i = 0; i < q.size(); ++i) {
          const uint32_t tok = q[i], exp = ref.count(tok) ? ref[tok] : 0u;
          const uint32_t a = plain::mrk::table_get(tab.data(), cap, tok, true);
          const uint32_t b = lean::mrk::table_get(tab.data(), cap, tok, true);
          const uint32_t tok2 = q[(i * 7 + 3) % q.size()], exp2 = ref.count(tok2) ? ref[tok2] : 0u;
          uint32_t c0, c1, d0, d1;
          lean_pair::mrk::table_get2(tab.data(), cap, tok, true, tok2, true, c0, c1);
          pair_only::mrk::table_get2(tab.data(), cap, tok, true, tok2, i % 3 != 0, d0, d1);
          const uint32_t ride = lean::mrk::table_get(tab.data(), cap, tok, false) + plain::mrk::table_get(tab.data(), cap, tok, false);
          checks += 6;
          if (a != exp || b != exp || c0 != exp || c1 != exp2 || d0 != exp || d1 != (i % 3 != 0 ? exp2 : 0u) || ride != 0u) {
            ++bad;
            if (bad < 20) printf("cap %u fill %d: tok %u expected %u: plain %u lean %u pair %u/%u (exp2 %u) pair_only %u/%u ride %u\n", cap, fill_pct, tok, exp, a, b, c0, c1, exp2, d0, d1, ride);
          }
        }
        // the list forms: sums in list order (integers: exact)
        std::vector<uint32_t> list;
        for (int i = 0; i < 21; ++i) list.push_back(q[rnd() % q.size()]);
        for (uint32_t len : {0u, 1u, 7u, 8u, 9u, 21u}) {
          double exp = 0.5;
          for (uint32_t i = 0; i < len; ++i) exp += ref.count(list[i]) ? ref[list[i]] : 0u;
          const double s0 = plain::mrk::table_sum_list(list.data(), tab.data(), cap, len, 0.5);
          const double s1 = lean::mrk::table_sum_list(list.data(), tab.data(), cap, len, 0.5);
          const double s2 = lean_pair::mrk::table_sum_list(list.data(), tab.data(), cap, len, 0.5);
          const double s3 = pair_only::mrk::table_sum_list(list.data(), tab.data(), cap, len, 0.5);
          checks += 4;
          if (s0 != exp || s1 != exp || s2 != exp || s3 != exp) { ++bad; if (bad < 20) printf("sum_list len %u: %g %g %g %g expected %g\n", len, s0, s1, s2, s3, exp); }
        }
      }
    }
  }
  printf("PROBE_W %d: %lld checks, %lld bad\n", (int)plain::mrk::PROBE_W, checks, bad);
  return bad ? 1 : 0;
}
