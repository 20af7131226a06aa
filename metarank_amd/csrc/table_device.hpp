// Device code of the token -> count hash tables of the pre-pass (interacted_with session profiles, diversity histograms):
// insert (table_add), lookup (table_get) and their list forms.  Part of the translation unit of every assembly kernel
// (rank_device.hpp includes it; the run-time specialised kernels get it through jit_embed.inc) - and, with a one-lane
// stand-in for the wavefront primitives, compiled for the HOST by tests/native/table_test.cpp, which checks every variant
// (default, MRK_LEAN_GET, MRK_GET_PAIR, window widths) against a std::map.
// (No `#pragma once`: the host test includes it more than once, with different macros, inside different namespaces.)
#ifndef MRK_TABLE_DEVICE_HPP
#define MRK_TABLE_DEVICE_HPP

namespace mrk {

namespace {

// ---------------------------------------------------------------- token -> count hash tables
// entry = key (token id, >= 1) in the low 32 bits, count in the high 32 bits; 0 = empty.  Open addressing
// with linear probing; the capacity is any number > the number of tokens inserted (not a power of two:
// the tables of a request live in LDS and their size sets the occupancy), the home slot is the
// multiply-high range reduction of a multiplicative hash.
// (homes lie in [0, cap - PROBE_W]: the first window of a probe sequence - the one most lookups end in - never wraps, so its
// entries are read at constant offsets from ONE address; later windows wrap around `cap`.  The host sizes every table >= 8.)
__device__ __forceinline__ uint32_t tok_home(uint32_t tok, uint32_t span) { return __umulhi(tok * 2654435761u, span); }

// Both primitives are written for the wavefront, not for the lane: the probe loop runs while ANY active lane is
// still looking (one scalar branch per round), lanes that are done - or that never wanted anything (`want`
// false) - ride along on selects.  A per-lane `while` costs ~25 scalar exec-mask instructions per probe.
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// A probe is a dependent trip to LDS, and a wave-uniform loop runs as long as its SLOWEST lane.  Most lookups of the
// assembly phase are misses (a candidate's token that the session profile does not hold), a miss ends at the first empty
// entry, and at the 75 % load the tables are sized for a miss walks 8.5 entries on average - the longest walk among 64
// lanes is 20 - 30.  Round 2 read 4 entries together and then walked the rest ONE entry per trip: 20 dependent LDS round
// trips per lookup.  Both primitives now take EVERY trip PROBE_W entries wide: independent reads, one wait, the whole window
// examined in registers.  Width measured same-box on c2 / c3 (gpurun_out r03_za; MRK_JIT_DEFINES="MRK_PROBE_W=n" compiles the
// specialised kernels with another width): 2 -> 0.269 / 0.415 ms, 3 -> 0.270 / 0.416, **4 -> 0.265 / 0.412**, 6 -> 0.276 / 0.420,
// 8 -> 0.288 / 0.423 (a window costs three VALU per entry for the wrap-around and two for the compare, and its registers are
// live next to the candidate's record); round 2's 4-then-1 loop: 0.282 / 0.431.
#ifndef MRK_PROBE_W
#define MRK_PROBE_W 4
#endif
constexpr int PROBE_W = MRK_PROBE_W;

#ifdef MRK_TABLE_BUCKETS
// ===== EXPERIMENT (-DMRK_TABLE_BUCKETS; compiled, checked on the host against a map, not yet run on a device) =====
// The same tables probed by BUCKET: a table is cap / PROBE_W aligned buckets of PROBE_W entries, a key lives in the first
// bucket of its probe sequence (home bucket, then the following ones, wrapping) that had room when it was inserted.  What
// the code object says about the default (tools: a build with -gline-tables-only): table_get is 30 % and table_add 25 % of
// the stock kernel's instructions, almost all of it in the LATER windows - a miss walks 8.5 entries on average at 75 % load
// (primary clustering), the slowest of 64 lanes 20 - 30, so the wave-uniform loop runs 5 - 7 times per lookup, and every
// entry of a later window pays its own wrap-around test.  Buckets end a miss at the first bucket with an EMPTY entry (a key
// never lies beyond one: entries are never emptied) - 1.4 buckets on average, about 4 for the slowest of 64 lanes - read every
// bucket at constant offsets from one address (a wrap-around test per bucket, not per entry) and examine it with the lean
// lookup's per-window bookkeeping.  The host needs no change: cap is even and >= 8, cap / PROBE_W * PROBE_W >= the tokens
// the table was sized for (features.cpp table_capacity: tokens / 0.75 + 2) as long as PROBE_W <= 4.
static_assert(PROBE_W == 2 || PROBE_W == 4, "bucketed tables: 2 or 4 entries per bucket");

// (the walk of one insert from bucket `bkt` on; `seen` buckets were found full of other keys already)
__device__ __forceinline__ bool table_add_from(unsigned long long *tab, uint32_t nb, uint32_t tok, bool open, uint32_t bkt, uint32_t seen) {
  const unsigned long long fresh = (unsigned long long)tok | (1ull << 32);
  bool full = false;
  while (wave_any(open)) {
    unsigned long long *bp = tab + bkt * (uint32_t)PROBE_W;
    unsigned long long e[PROBE_W];
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) e[k] = bp[k];
    uint32_t stop = PROBE_W;  // the FIRST entry of the bucket that holds tok or is empty
    bool at_key = false;
#pragma unroll
    for (int k = PROBE_W - 1; k >= 0; --k) {
      const uint32_t key = (uint32_t)e[k];
      if (key == tok || key == 0u) { stop = (uint32_t)k; at_key = key == tok; }
    }
    const bool here = open && stop < (uint32_t)PROBE_W;
    unsigned long long *at = bp + (stop < (uint32_t)PROBE_W ? stop : 0u);
    if (here && at_key) atomicAdd(at, 1ull << 32);
    unsigned long long prev = ~0ull;
    if (here && !at_key) prev = atomicCAS(at, 0ull, fresh);
    const bool took = here && !at_key && prev == 0ull;
    const bool same = here && !at_key && (uint32_t)prev == tok;    // another lane put tok there in the meantime
    if (same) atomicAdd(at, 1ull << 32);
    const bool done = here && (at_key || took || same);
    // not done: the bucket holds other keys only (on to the next one), or its empty entry went to another key (the SAME
    // bucket once more: it may have another empty entry)
    const bool next = open && !here;
    seen += next ? 1u : 0u;
    bkt = next ? (bkt + 1u == nb ? 0u : bkt + 1u) : bkt;
    full = full || (open && !done && seen >= nb);
    open = open && !done && seen < nb;
  }
  return !full;
}

// (the walk of one lookup from bucket `bkt` on; `seen` buckets were looked at already)
__device__ __forceinline__ uint32_t table_get_from(const unsigned long long *tab, uint32_t nb, uint32_t tok, bool open, uint32_t bkt, uint32_t seen,
                                                   uint32_t res) {
  for (; wave_any(open); ++seen) {
    const unsigned long long *bp = tab + bkt * (uint32_t)PROBE_W;
    unsigned long long e[PROBE_W];
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) e[k] = bp[k];
    uint32_t r2 = 0u, lo = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) {
      const uint32_t key = (uint32_t)e[k];
      r2 = key == tok ? (uint32_t)(e[k] >> 32) : r2;
      lo = min(lo, key);
    }
    res = open ? r2 : res;
    open = open && r2 == 0u && lo != 0u && seen + 1u < nb;   // found, or an empty entry, or every bucket seen: done
    bkt = bkt + 1u == nb ? 0u : bkt + 1u;
  }
  return res;
}

#ifdef MRK_TABLE_2CHOICE
#ifdef MRK_GET_PAIR
#error "MRK_TABLE_2CHOICE reads two buckets per lookup already: not together with MRK_GET_PAIR"
#endif
// ===== EXPERIMENT on top of the buckets (-DMRK_TABLE_BUCKETS -DMRK_TABLE_2CHOICE) =====
// TWO home buckets per key.  With one home, a miss at the 75 % load the tables are sized for walks on while buckets are
// full - the slowest of a wavefront's 64 lookups needs ~8 trips (simulated; the same for unaligned windows).  A key that may
// go to the emptier of two buckets leaves far fewer buckets full: both are read in ONE trip, and only if both are full of
// other keys does the lookup walk on from behind the second one - ~3 trips for the slowest of 64 at 75 % load, 1.1 at 55 %.
// An insert looks for its key in both, else takes the first empty entry of the emptier one; two lanes inserting the same
// new key at the same moment may put it into one bucket each, so a lookup ADDS what it finds in the two.
__device__ __forceinline__ uint32_t tok_alt(uint32_t tok, uint32_t home, uint32_t nb) {   // the second bucket: never the first (nb >= 2)
  const uint32_t b = home + 1u + __umulhi((tok ^ 0x9e3779b9u) * 0x85ebca6bu, nb - 1u);
  return b >= nb ? b - nb : b;
}

__device__ __forceinline__ bool table_add(unsigned long long *tab, uint32_t cap, uint32_t tok, bool want) {
  const uint32_t nb = cap / (uint32_t)PROBE_W;
  const uint32_t ba = tok_home(tok, nb), bb = tok_alt(tok, ba, nb);
  unsigned long long *pa = tab + ba * (uint32_t)PROBE_W, *pb = tab + bb * (uint32_t)PROBE_W;
  const unsigned long long fresh = (unsigned long long)tok | (1ull << 32);
  bool open = want;
  bool overflow = false;   // both buckets full of other keys: on to the buckets behind the second one
  while (wave_any(open)) {
    unsigned long long ea[PROBE_W], eb[PROBE_W];
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) ea[k] = pa[k];
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) eb[k] = pb[k];
    uint32_t key_a = PROBE_W, key_b = PROBE_W, emp_a = PROBE_W, emp_b = PROBE_W, n_emp_a = 0, n_emp_b = 0;  // first entry with tok / first empty entry / empties
#pragma unroll
    for (int k = PROBE_W - 1; k >= 0; --k) {
      const uint32_t ka = (uint32_t)ea[k], kb = (uint32_t)eb[k];
      if (ka == tok) key_a = (uint32_t)k;
      if (kb == tok) key_b = (uint32_t)k;
      if (ka == 0u) { emp_a = (uint32_t)k; n_emp_a += 1u; }
      if (kb == 0u) { emp_b = (uint32_t)k; n_emp_b += 1u; }
    }
    const bool at_key = key_a < (uint32_t)PROBE_W || key_b < (uint32_t)PROBE_W;
    const bool room = n_emp_a + n_emp_b > 0u;
    const bool in_a = at_key ? key_a < (uint32_t)PROBE_W : n_emp_a >= n_emp_b;   // (room: the emptier bucket, the first on a tie)
    const uint32_t slot = at_key ? (in_a ? key_a : key_b) : (in_a ? emp_a : emp_b);
    const bool here = open && (at_key || room);
    unsigned long long *at = (in_a ? pa : pb) + (slot < (uint32_t)PROBE_W ? slot : 0u);
    if (here && at_key) atomicAdd(at, 1ull << 32);
    unsigned long long prev = ~0ull;
    if (here && !at_key) prev = atomicCAS(at, 0ull, fresh);
    const bool took = here && !at_key && prev == 0ull;
    const bool same = here && !at_key && (uint32_t)prev == tok;
    if (same) atomicAdd(at, 1ull << 32);
    const bool done = here && (at_key || took || same);
    overflow = overflow || (open && !here);
    open = open && here && !done;   // the empty entry went to another key: look at the two buckets again
  }
  if (nb <= 2u) return !overflow;   // (two buckets are the whole table)
  return table_add_from(tab, nb, tok, overflow, bb + 1u == nb ? 0u : bb + 1u, 1u);   // (the walk passes the first bucket again: every bucket but the second)
}

__device__ __forceinline__ uint32_t table_get(const unsigned long long *tab, uint32_t cap, uint32_t tok_in, bool want) {
  if (!wave_any(want)) return 0u;
  const uint32_t tok = want ? tok_in : 0u;   // a lane that rides along looks for key 0: an empty entry, count 0
  const uint32_t nb = cap / (uint32_t)PROBE_W;
  const uint32_t ba = tok_home(tok, nb), bb = tok_alt(tok, ba, nb);
  const unsigned long long *pa = tab + ba * (uint32_t)PROBE_W, *pb = tab + bb * (uint32_t)PROBE_W;
  unsigned long long ea[PROBE_W], eb[PROBE_W];
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) ea[k] = pa[k];
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) eb[k] = pb[k];
  uint32_t ra = 0u, rb = 0u, lo = 0xffffffffu;
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) {
    const uint32_t ka = (uint32_t)ea[k], kb = (uint32_t)eb[k];
    ra = ka == tok ? (uint32_t)(ea[k] >> 32) : ra;
    rb = kb == tok ? (uint32_t)(eb[k] >> 32) : rb;
    lo = min(lo, min(ka, kb));
  }
  const uint32_t res = ra + rb;
  // not found and no empty entry in either: the key may lie behind the second bucket
  return table_get_from(tab, nb, tok, want && res == 0u && lo != 0u && nb > 2u, bb + 1u == nb ? 0u : bb + 1u, 1u, res);
}
#else  // one home bucket
__device__ __forceinline__ bool table_add(unsigned long long *tab, uint32_t cap, uint32_t tok, bool want) {
  const uint32_t nb = cap / (uint32_t)PROBE_W;
  return table_add_from(tab, nb, tok, want, tok_home(tok, nb), 0u);
}

__device__ __forceinline__ uint32_t table_get(const unsigned long long *tab, uint32_t cap, uint32_t tok_in, bool want) {
  const uint32_t tok = want ? tok_in : 0u;   // a lane that rides along looks for key 0: an empty entry, count 0
  const uint32_t nb = cap / (uint32_t)PROBE_W;
  return table_get_from(tab, nb, tok, want, tok_home(tok, nb), 0u, 0u);
}

#ifdef MRK_GET_PAIR
// two lookups whose home buckets travel together (see the linear-probing form below)
__device__ __forceinline__ void table_get2(const unsigned long long *tab, uint32_t cap, uint32_t tok0_in, bool want0, uint32_t tok1_in, bool want1,
                                           uint32_t &res0, uint32_t &res1) {
  res0 = 0u;
  res1 = 0u;
  if (!wave_any(want0 || want1)) return;
  const uint32_t tok0 = want0 ? tok0_in : 0u, tok1 = want1 ? tok1_in : 0u;
  const uint32_t nb = cap / (uint32_t)PROBE_W;
  uint32_t b0 = tok_home(tok0, nb), b1 = tok_home(tok1, nb);
  const unsigned long long *p0 = tab + b0 * (uint32_t)PROBE_W, *p1 = tab + b1 * (uint32_t)PROBE_W;
  unsigned long long e0[PROBE_W], e1[PROBE_W];
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) e0[k] = p0[k];
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) e1[k] = p1[k];
  uint32_t lo0 = 0xffffffffu, lo1 = 0xffffffffu;
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) {
    const uint32_t key = (uint32_t)e0[k];
    res0 = key == tok0 ? (uint32_t)(e0[k] >> 32) : res0;
    lo0 = min(lo0, key);
  }
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) {
    const uint32_t key = (uint32_t)e1[k];
    res1 = key == tok1 ? (uint32_t)(e1[k] >> 32) : res1;
    lo1 = min(lo1, key);
  }
  b0 = b0 + 1u == nb ? 0u : b0 + 1u;
  b1 = b1 + 1u == nb ? 0u : b1 + 1u;
  res0 = table_get_from(tab, nb, tok0, want0 && res0 == 0u && lo0 != 0u && nb > 1u, b0, 1u, res0);
  res1 = table_get_from(tab, nb, tok1, want1 && res1 == 0u && lo1 != 0u && nb > 1u, b1, 1u, res1);
}
#endif  // MRK_GET_PAIR
#endif  // MRK_TABLE_2CHOICE

#else  // linear probing (the default)

__device__ __forceinline__ bool table_add(unsigned long long *tab, uint32_t cap, uint32_t tok, bool want) {
  // tab / cap may differ between lanes (item-parallel kernel: lanes of several requests in one wavefront)
  uint32_t idx = tok_home(tok, cap - (uint32_t)(PROBE_W - 1));
  const unsigned long long fresh = (unsigned long long)tok | (1ull << 32);
  bool open = want;          // still looking for tok's entry
  bool full = false;
  uint32_t walked = 0;       // entries known to hold other keys
  bool first = true;         // (uniform) the window at the home entry: no wrap-around
  while (wave_any(open)) {
    // a window of the probe sequence: entries that hold OTHER keys can be skipped for good (a key, once set, never changes)
    unsigned long long e[PROBE_W];
    uint32_t pos[PROBE_W];
    if (first) {
#pragma unroll
      for (int k = 0; k < PROBE_W; ++k) {
        pos[k] = idx + (uint32_t)k;
        e[k] = tab[idx + (uint32_t)k];
      }
    } else {
      uint32_t ix = idx;
#pragma unroll
      for (int k = 0; k < PROBE_W; ++k) {
        pos[k] = ix;
        e[k] = tab[ix];  // every lane reads: ix stays inside its table
        ix = ix + 1 == cap ? 0 : ix + 1;
      }
    }
    first = false;
    uint32_t stop = PROBE_W;  // the FIRST entry of the window that holds tok or is empty
    bool at_key = false;
#pragma unroll
    for (int k = PROBE_W - 1; k >= 0; --k) {
      const uint32_t key = (uint32_t)e[k];
      if (key == tok || key == 0u) { stop = (uint32_t)k; at_key = key == tok; }
    }
    const bool here = open && stop < (uint32_t)PROBE_W && walked + stop < cap;
    const uint32_t at = pos[stop < (uint32_t)PROBE_W ? stop : 0];
    if (here && at_key) atomicAdd(&tab[at], 1ull << 32);  // the key is there already: one atomic, no compare-and-swap
    unsigned long long prev = ~0ull;
    if (here && !at_key) prev = atomicCAS(&tab[at], 0ull, fresh);  // empty -> {tok, 1}
    const bool took = here && !at_key && prev == 0ull;
    const bool same = here && !at_key && (uint32_t)prev == tok;    // another lane put tok there in the meantime
    if (same) atomicAdd(&tab[at], 1ull << 32);
    const bool done = here && (at_key || took || same);
    // not done: either the whole window holds other keys (walk on behind it), or the empty entry went to another key
    // (walk on behind that entry)
    const uint32_t adv = here ? stop + 1 : (uint32_t)PROBE_W;
    walked += adv;
    idx = idx + adv;
    idx = idx >= cap ? idx - cap : idx;
    full = full || (open && !done && walked >= cap);  // every entry holds another key
    open = open && !done && walked < cap;
  }
  return !full;
}

#ifdef MRK_LEAN_GET
// ===== EXPERIMENT (-DMRK_LEAN_GET; compiled, not yet run on a device) =====
// The same lookup with the bookkeeping per WINDOW instead of per entry.  The default examines every entry with
// `hit = open && key == tok; open = open && key != tok && key != 0` - 4 VALU + 3 SALU (64-bit lane masks) per entry.  In an
// insert-only table with linear probing no empty entry ever precedes a key on its probe sequence (the key went into the
// FIRST empty entry of that sequence, and entries are never emptied), and a key occurs once: the count is simply that of
// whichever entry holds the key, and the walk ends when the window held the key or an empty entry - decided once per
// window from `res != 0` (counts are >= 1) and the minimum of the window's keys.  A lane that only rides along looks for
// key 0: an empty entry's count is 0.
__device__ __forceinline__ uint32_t table_get(const unsigned long long *tab, uint32_t cap, uint32_t tok_in, bool want) {
  if (!wave_any(want)) return 0u;
  const uint32_t tok = want ? tok_in : 0u;
  uint32_t idx = tok_home(tok, cap - (uint32_t)(PROBE_W - 1));
  uint32_t res = 0u, lo = 0xffffffffu;
  {
    unsigned long long e[PROBE_W];
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) e[k] = tab[idx + (uint32_t)k];
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) {
      const uint32_t key = (uint32_t)e[k];
      res = key == tok ? (uint32_t)(e[k] >> 32) : res;
      lo = min(lo, key);
    }
    idx += (uint32_t)PROBE_W;
    idx = idx >= cap ? idx - cap : idx;
  }
  bool open = want && res == 0u && lo != 0u;
  for (uint32_t walked = PROBE_W; wave_any(open); walked += PROBE_W) {   // (a wrapped walk may look at the first entries twice: harmless)
    unsigned long long e[PROBE_W];
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) {
      e[k] = tab[idx];
      idx = idx + 1 == cap ? 0 : idx + 1;
    }
    uint32_t r2 = 0u;
    lo = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) {
      const uint32_t key = (uint32_t)e[k];
      r2 = key == tok ? (uint32_t)(e[k] >> 32) : r2;
      lo = min(lo, key);
    }
    res = open ? r2 : res;
    open = open && r2 == 0u && lo != 0u && walked + (uint32_t)PROBE_W < cap;
  }
  return res;
}
#else
__device__ __forceinline__ uint32_t table_get(const unsigned long long *tab, uint32_t cap, uint32_t tok, bool want) {
  if (!wave_any(want)) return 0u;  // (a diversity column of string LISTS asks for no single-string lookup at all)
  uint32_t idx = tok_home(tok, cap - (uint32_t)(PROBE_W - 1));
  uint32_t res = 0;
  bool open = want;
  {  // the window at the home entry: one trip, no wrap-around, serves most lanes
    unsigned long long e[PROBE_W];
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) e[k] = tab[idx + (uint32_t)k];
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) {
      const uint32_t key = (uint32_t)e[k];
      res = open && key == tok ? (uint32_t)(e[k] >> 32) : res;
      open = open && key != tok && key != 0u;  // keys are token ids >= 1: key 0 = empty entry
    }
    idx += (uint32_t)PROBE_W;
    idx = idx >= cap ? idx - cap : idx;
  }
  for (uint32_t walked = PROBE_W; wave_any(open); walked += PROBE_W) {
    unsigned long long e[PROBE_W];
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) {
      e[k] = tab[idx];  // every lane reads: idx stays inside its table
      idx = idx + 1 == cap ? 0 : idx + 1;
    }
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) {
      const uint32_t key = (uint32_t)e[k];
      res = open && key == tok ? (uint32_t)(e[k] >> 32) : res;
      open = open && key != tok && key != 0u && walked + (uint32_t)(k + 1) < cap;
    }
  }
  return res;
}
#endif  // MRK_LEAN_GET

#ifdef MRK_GET_PAIR
// ===== EXPERIMENT (-DMRK_GET_PAIR / MRK_JIT_DEFINES="MRK_GET_PAIR=1"; compiled, not yet run on a device) =====
// TWO lookups whose home windows travel together: a lookup is one dependent trip to LDS (the window is read, waited for,
// examined), and the lookups of a candidate's tokens do not depend on each other - the per-item phase spends 30 k cycles
// of a request's 113 k in `profile`'s lookups and 10 k in each string-diversity column, one trip after the other.  Both
// home windows are requested before the first wait; the (rare) later windows of either are walked as before.
#ifdef MRK_LEAN_GET   // (both experiments: the pair of lookups with the per-window bookkeeping of the lean lookup)
__device__ __forceinline__ uint32_t table_get_rest(const unsigned long long *tab, uint32_t cap, uint32_t tok, bool open, uint32_t idx, uint32_t res) {
  for (uint32_t walked = PROBE_W; wave_any(open); walked += PROBE_W) {
    unsigned long long e[PROBE_W];
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) {
      e[k] = tab[idx];
      idx = idx + 1 == cap ? 0 : idx + 1;
    }
    uint32_t r2 = 0u, lo = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) {
      const uint32_t key = (uint32_t)e[k];
      r2 = key == tok ? (uint32_t)(e[k] >> 32) : r2;
      lo = min(lo, key);
    }
    res = open ? r2 : res;
    open = open && r2 == 0u && lo != 0u && walked + (uint32_t)PROBE_W < cap;
  }
  return res;
}

__device__ __forceinline__ void table_get2(const unsigned long long *tab, uint32_t cap, uint32_t tok0_in, bool want0, uint32_t tok1_in, bool want1,
                                           uint32_t &res0, uint32_t &res1) {
  res0 = 0u;
  res1 = 0u;
  if (!wave_any(want0 || want1)) return;
  const uint32_t tok0 = want0 ? tok0_in : 0u, tok1 = want1 ? tok1_in : 0u;
  uint32_t idx0 = tok_home(tok0, cap - (uint32_t)(PROBE_W - 1)), idx1 = tok_home(tok1, cap - (uint32_t)(PROBE_W - 1));
  unsigned long long e0[PROBE_W], e1[PROBE_W];
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) e0[k] = tab[idx0 + (uint32_t)k];
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) e1[k] = tab[idx1 + (uint32_t)k];
  uint32_t lo0 = 0xffffffffu, lo1 = 0xffffffffu;
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) {
    const uint32_t key = (uint32_t)e0[k];
    res0 = key == tok0 ? (uint32_t)(e0[k] >> 32) : res0;
    lo0 = min(lo0, key);
  }
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) {
    const uint32_t key = (uint32_t)e1[k];
    res1 = key == tok1 ? (uint32_t)(e1[k] >> 32) : res1;
    lo1 = min(lo1, key);
  }
  idx0 += (uint32_t)PROBE_W;
  idx0 = idx0 >= cap ? idx0 - cap : idx0;
  idx1 += (uint32_t)PROBE_W;
  idx1 = idx1 >= cap ? idx1 - cap : idx1;
  res0 = table_get_rest(tab, cap, tok0, want0 && res0 == 0u && lo0 != 0u, idx0, res0);
  res1 = table_get_rest(tab, cap, tok1, want1 && res1 == 0u && lo1 != 0u, idx1, res1);
}
#else
__device__ __forceinline__ uint32_t table_get_rest(const unsigned long long *tab, uint32_t cap, uint32_t tok, bool open, uint32_t idx, uint32_t res) {
  for (uint32_t walked = PROBE_W; wave_any(open); walked += PROBE_W) {
    unsigned long long e[PROBE_W];
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) {
      e[k] = tab[idx];
      idx = idx + 1 == cap ? 0 : idx + 1;
    }
#pragma unroll
    for (int k = 0; k < PROBE_W; ++k) {
      const uint32_t key = (uint32_t)e[k];
      res = open && key == tok ? (uint32_t)(e[k] >> 32) : res;
      open = open && key != tok && key != 0u && walked + (uint32_t)(k + 1) < cap;
    }
  }
  return res;
}

__device__ __forceinline__ void table_get2(const unsigned long long *tab, uint32_t cap, uint32_t tok0, bool want0, uint32_t tok1, bool want1,
                                           uint32_t &res0, uint32_t &res1) {
  res0 = 0u;
  res1 = 0u;
  if (!wave_any(want0 || want1)) return;
  uint32_t idx0 = tok_home(tok0, cap - (uint32_t)(PROBE_W - 1)), idx1 = tok_home(tok1, cap - (uint32_t)(PROBE_W - 1));
  bool open0 = want0, open1 = want1;
  unsigned long long e0[PROBE_W], e1[PROBE_W];
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) e0[k] = tab[idx0 + (uint32_t)k];
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) e1[k] = tab[idx1 + (uint32_t)k];
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) {
    const uint32_t key = (uint32_t)e0[k];
    res0 = open0 && key == tok0 ? (uint32_t)(e0[k] >> 32) : res0;
    open0 = open0 && key != tok0 && key != 0u;
  }
#pragma unroll
  for (int k = 0; k < PROBE_W; ++k) {
    const uint32_t key = (uint32_t)e1[k];
    res1 = open1 && key == tok1 ? (uint32_t)(e1[k] >> 32) : res1;
    open1 = open1 && key != tok1 && key != 0u;
  }
  idx0 += (uint32_t)PROBE_W;
  idx0 = idx0 >= cap ? idx0 - cap : idx0;
  idx1 += (uint32_t)PROBE_W;
  idx1 = idx1 >= cap ? idx1 - cap : idx1;
  res0 = table_get_rest(tab, cap, tok0, open0, idx0, res0);
  res1 = table_get_rest(tab, cap, tok1, open1, idx1, res1);
}
#endif  // MRK_LEAN_GET
#endif  // MRK_GET_PAIR
#endif  // MRK_TABLE_BUCKETS

// The tokens of a list are fetched TOK_BATCH at a time (independent loads in flight together) before the
// probes start: one trip to memory per batch instead of one per token.
#ifndef MRK_TOK_BATCH
#define MRK_TOK_BATCH 8
#endif
constexpr int TOK_BATCH = MRK_TOK_BATCH;

// every token of toks[0, len) -> table (pre-pass); len = 0 for lanes without a list.  Returns the
// number of tokens this lane could not insert (table full).
__device__ __forceinline__ uint32_t table_add_list(const uint32_t *toks, unsigned long long *tab, uint32_t cap, uint32_t len) {
  uint32_t failed = 0;
  for (uint32_t j0 = 0; wave_any(j0 < len); j0 += TOK_BATCH) {
    uint32_t tk[TOK_BATCH];
#pragma unroll
    for (int t = 0; t < TOK_BATCH; ++t) tk[t] = j0 + t < len ? toks[j0 + t] : 0u;
#pragma unroll
    for (int t = 0; t < TOK_BATCH; ++t) {
      if (!wave_any(j0 + t < len)) break;
      failed += table_add(tab, cap, tk[t], j0 + t < len) ? 0u : 1u;
    }
  }
  return failed;
}

// sum over the tokens toks[0, len) of their table counts, added as doubles in list order
// (InteractedWithFeature.scala:150-160 / DiversityFeature.scala:112-122: integers, exact)
__device__ __forceinline__ double table_sum_list(const uint32_t *toks, const unsigned long long *tab, uint32_t cap, uint32_t len,
                                                 double cnt = 0.0) {
  for (uint32_t j0 = 0; wave_any(j0 < len); j0 += TOK_BATCH) {
    uint32_t tk[TOK_BATCH];
#pragma unroll
    for (int t = 0; t < TOK_BATCH; ++t) tk[t] = j0 + t < len ? toks[j0 + t] : 0u;
#ifdef MRK_GET_PAIR
    static_assert(TOK_BATCH % 2 == 0, "pairs");
#pragma unroll
    for (int t = 0; t < TOK_BATCH; t += 2) {
      if (!wave_any(j0 + t < len)) break;
      uint32_t g0, g1;
      table_get2(tab, cap, tk[t], j0 + t < len, tk[t + 1], j0 + t + 1 < len, g0, g1);
      cnt = cnt + (double)g0;
      cnt = cnt + (double)g1;
    }
#else
#pragma unroll
    for (int t = 0; t < TOK_BATCH; ++t) {
      if (!wave_any(j0 + t < len)) break;
      cnt = cnt + (double)table_get(tab, cap, tk[t], j0 + t < len);  // riding lanes add 0.0
    }
#endif
  }
  return cnt;
}

}  // namespace

}  // namespace mrk

#endif  // MRK_TABLE_DEVICE_HPP
