// Device code of the token -> count hash tables of the pre-pass (interacted_with session profiles, diversity histograms):
// insert (table_add), lookup (table_get) and their list forms.  Part of the translation unit of every assembly kernel
// (rank_device.hpp includes it; the run-time specialised kernels get it through jit_embed.inc) - and, with a one-lane
// stand-in for the wavefront primitives, compiled for the HOST by tests/native/table_test.cpp, which checks it against a
// std::map (bucket widths 2 / 4, concurrent inserts).
// (No `#pragma once`: the host test includes it inside a namespace of its own.)
#ifndef MRK_TABLE_DEVICE_HPP
#define MRK_TABLE_DEVICE_HPP

namespace mrk {

namespace {

// ---------------------------------------------------------------- token -> count hash tables
// entry = key (token id, >= 1) in the low 32 bits, count in the high 32 bits; 0 = empty.  A table is cap / PROBE_W ALIGNED
// BUCKETS of PROBE_W entries (open addressing by bucket): a key lives in the first bucket of its probe sequence - home
// bucket, then the following ones, wrapping - that had room when it was inserted; entries are never emptied, so a miss ends
// at the first bucket with an empty entry.  The capacity is any even number >= 8 (not a power of two: the tables of a
// request live in LDS and their size sets the occupancy; features.cpp table_capacity sizes them tokens / 0.75 + 2, and
// cap / PROBE_W * PROBE_W still holds those tokens for PROBE_W <= 4); the home bucket is the multiply-high range reduction
// of a multiplicative hash.
// Round 6: bit 63 of a bucket's FIRST entry says "an insert walked past this bucket" (it was full of other keys then).  A key
// lives in the first bucket of its probe sequence that had room when it came, and every bucket before that one was walked
// past - so a lookup ends at the first bucket without the mark, found or not.  Without it a miss ends only at a bucket with an
// empty entry: at load 0.75 four of ten buckets are full, a miss walks 1.6 buckets on average - and a wavefront's lookup loop
// runs as long as its SLOWEST lane: 4 - 5 trips for 64 lanes, where the mark leaves ~2 (tools/phase_clocks.py: the lookups
// were 55 % of an unloaded request's assembly phase).  Counts are < 2^31 (bits 32..62).
constexpr unsigned long long TABLE_WALKED = 1ull << 63;
#ifndef MRK_TABLE_WALKED
#define MRK_TABLE_WALKED 1   // 0 (A/B through MRK_JIT_DEFINES): nothing is marked, a miss ends at an empty entry only
#endif
__device__ __forceinline__ uint32_t tok_home(uint32_t tok, uint32_t span) { return __umulhi(tok * 2654435761u, span); }

// Both primitives are written for the wavefront, not for the lane: the probe loop runs while ANY active lane is
// still looking (one scalar branch per round), lanes that are done - or that never wanted anything (`want`
// false) - ride along on selects.  A per-lane `while` costs ~25 scalar exec-mask instructions per probe.
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// A probe is a dependent trip to LDS, and a wave-uniform loop runs as long as its SLOWEST lane.  Most lookups of the assembly
// phase are misses (a candidate's token that the session profile does not hold).  Every trip reads one whole bucket -
// PROBE_W independent reads at constant offsets from ONE address, one wait - and examines it in registers with per-bucket
// bookkeeping: the count of whichever entry holds the key, the minimum of the keys (0 = an empty entry: stop), one wrap-around
// test per bucket.  History (same-box A/Bs, LOG.md): one entry per trip -> 4-entry unaligned windows (c2 assembly -10 %) ->
// aligned buckets (round 4: -34 % static instructions, c2 -7.5 %, c3 -3.5 %).  Measured and dropped in round 4
// (gpurun_out/r04_first): two home buckets per key (+10 % on c2: two buckets in registers spill), two lookups per LDS trip
// (neutral on c2, +2 % on c3), per-window bookkeeping on unaligned windows (-3 %: superseded by the buckets).
#ifndef MRK_PROBE_W
#define MRK_PROBE_W 4
#endif
constexpr int PROBE_W = MRK_PROBE_W;

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
    if (MRK_TABLE_WALKED && next && !(e[0] & TABLE_WALKED)) atomicOr(bp, TABLE_WALKED);   // (the first entry of a full bucket is never empty)
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
      r2 = key == tok ? (uint32_t)(e[k] >> 32) & 0x7fffffffu : r2;
      lo = min(lo, key);
    }
    res = open ? r2 : res;
    // found, or an empty entry, or no insert ever walked past this bucket, or every bucket seen: done
    open = open && r2 == 0u && lo != 0u && (!MRK_TABLE_WALKED || (e[0] & TABLE_WALKED) != 0ull) && seen + 1u < nb;
    bkt = bkt + 1u == nb ? 0u : bkt + 1u;
  }
  return res;
}

__device__ __forceinline__ bool table_add(unsigned long long *tab, uint32_t cap, uint32_t tok, bool want) {
  const uint32_t nb = cap / (uint32_t)PROBE_W;
  return table_add_from(tab, nb, tok, want, tok_home(tok, nb), 0u);
}

__device__ __forceinline__ uint32_t table_get(const unsigned long long *tab, uint32_t cap, uint32_t tok_in, bool want) {
  const uint32_t tok = want ? tok_in : 0u;   // a lane that rides along looks for key 0: an empty entry, count 0
  const uint32_t nb = cap / (uint32_t)PROBE_W;
  return table_get_from(tab, nb, tok, want, tok_home(tok, nb), 0u, 0u);
}

// (Measured and removed, round 6 - profiles/r06_l_get_k_ab.txt: K lookups of one table sharing their trips - the K buckets of a trip read
// together, one loop for all K.  K = 2 / 4 against one at a time, same box: c2 assembly 0.212 / 0.221 vs 0.206 ms, c4x 0.828 / 0.907 vs
// 0.775 ms - K buckets in registers spill (6 -> 17 -> 53 VGPRs spilled in the c2 kernel), and with the walked-past mark a lookup
// is 1 - 2 trips anyway.)
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

// cnt + the table counts of tk[0, min(len, N)), added as doubles in token order (lanes past their list add 0.0)
template <int N>
__device__ __forceinline__ double table_sum_tokens(const unsigned long long *tab, uint32_t cap, const uint32_t (&tk)[N], uint32_t len, double cnt) {
#pragma unroll
  for (int t = 0; t < N; ++t) {
    if (!wave_any((uint32_t)t < len)) break;
    cnt = cnt + (double)table_get(tab, cap, tk[t], (uint32_t)t < len);
  }
  return cnt;
}

// sum over the tokens toks[0, len) of their table counts, added as doubles in list order
// (InteractedWithFeature.scala:150-160 / DiversityFeature.scala:112-122: integers, exact)
__device__ __forceinline__ double table_sum_list(const uint32_t *toks, const unsigned long long *tab, uint32_t cap, uint32_t len,
                                                 double cnt = 0.0) {
  for (uint32_t j0 = 0; wave_any(j0 < len); j0 += TOK_BATCH) {
    uint32_t tk[TOK_BATCH];
#pragma unroll
    for (int t = 0; t < TOK_BATCH; ++t) tk[t] = j0 + t < len ? toks[j0 + t] : 0u;
    cnt = table_sum_tokens<TOK_BATCH>(tab, cap, tk, len > j0 ? len - j0 : 0u, cnt);
  }
  return cnt;
}

}  // namespace

}  // namespace mrk

#endif  // MRK_TABLE_DEVICE_HPP
