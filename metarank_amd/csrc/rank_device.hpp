// Device code of the feature assembly (pre-pass + per-item program), shared by the kernels of rank.hip - which run
// ANY model program from device memory (ProgramDev) - and by the run-time specialised kernel jit.cpp builds for one
// model with hiprtc: there `Prog` is a type whose ops / prep / aux are compile-time constants, the op loop unrolls,
// every switch folds and the record loads of all ops are issued together.
// Reference: ml/Ranker.scala:27-83,97-106; feature/*.scala (cited per op in rank.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "qs_device.hpp"
#include "rank.hpp"
#include "sort_device.hpp"
#include "table_device.hpp"
#include "wave_device.hpp"

#ifdef MRK_PHASE_CLOCKS
// measurement builds only (MRK_DEFINES=MRK_PHASE_CLOCKS): core-clock cycles thread 0 of every workgroup spends per phase
// of the fused kernel, summed over workgroups: [0] table sweep, [1] interacted_with histograms, [2] diversity find-first +
// type, [3] diversity strings, [4] diversity medians, [5] per-item assembly, [6] workgroups; [16 + op] per op (lane 0)
__device__ unsigned long long mrk_phase_clocks[64];
// (kept in registers and flushed with atomics once, at the end: an atomic per boundary would sit in front of
// every later load in the in-order memory pipeline)
#define MRK_PHASE(prev, acc)                                                   \
  do {                                                                         \
    const unsigned long long t_ = clock64();                                   \
    (acc) += t_ - (prev);                                                      \
    (prev) = t_;                                                               \
  } while (0)
#else
#define MRK_PHASE(prev, acc) do { } while (0)
#endif

namespace mrk {

namespace {

__device__ __forceinline__ double d_nan() { return __longlong_as_double(0x7ff8000000000000LL); }

struct Cell {
  uint32_t tag;
  uint64_t bits;
  __device__ __forceinline__ double f64() const { return __longlong_as_double((long long)bits); }
  __device__ __forceinline__ long long i64() const { return (long long)bits; }
  __device__ __forceinline__ uint32_t lo() const { return (uint32_t)bits; }
  __device__ __forceinline__ uint32_t hi() const { return (uint32_t)(bits >> 32); }
};

__device__ __forceinline__ const uint8_t *record(const StoreDev &st, int scope, int slot) {
  if (slot < 0) return nullptr;
  return st.tab[scope].rows + (size_t)slot * st.tab[scope].stride;
}

__device__ __forceinline__ Cell load_cell(const uint8_t *rec, ColRef c, int idx = 0) {
  Cell out;
  if (rec == nullptr || c.tag < 0) {
    out.tag = TAG_MISSING;
    out.bits = 0;
    return out;
  }
  out.tag = rec[c.tag];
  out.bits = *(const uint64_t *)(rec + c.val + idx * 8);
  return out;
}

// the tokens of a string list cell of record `rec`: in the record's own inline heap (same 128-byte lines as the cells:
// no second trip to memory) or, for lists that did not fit, in the token pool
__device__ __forceinline__ const uint32_t *list_tokens(const StoreDev &st, const uint8_t *rec, uint32_t off) {
  return (off & LIST_INLINE_BIT) ? (const uint32_t *)(rec + (off & ~LIST_INLINE_BIT)) : st.tok_pool + off;
}

// element k of a double list cell {off, len}: from the f32 pool (LIST_F32_BIT: every value is exactly a float) or the f64 pool
__device__ __forceinline__ double list_f64(const StoreDev &st, uint32_t off, uint32_t k) {
  return (off & LIST_F32_BIT) ? (double)st.f32_pool[(off & ~LIST_F32_BIT) + k] : st.f64_pool[off + k];
}

__device__ __forceinline__ int scoped_slot(const ReqDev &rq, int scope, int item_slot) {
  switch (scope) {
    case SC_GLOBAL: return 0;
    case SC_ITEM: return item_slot;
    case SC_USER: return rq.user_slot;
    case SC_SESSION: return rq.session_slot;
    case SC_RANKING: return rq.ranking_slot;
    default: return -1;
  }
}

// ---------------------------------------------------------------- pre-pass
constexpr int PREP_THREADS = 256;
constexpr int FUSED_MAX_PREP = 32;  // pre-pass entries of one model (PrepOut copies + `first` slots kept in LDS)
constexpr int PREP_GROUP = 4;       // entries handled by one merged pass (their loads are issued together)
constexpr int PREP_WAVES = 8;       // most wavefronts of a workgroup that runs a pre-pass (512 lanes: a single request split over op groups)
#ifndef MRK_DIV_GROUP
#define MRK_DIV_GROUP 4
#endif
constexpr int DIV_GROUP = MRK_DIV_GROUP;   // diversity entries the wave-local section handles in one pass (8 - the five of the Ranklens model in one pass - measured no faster under load and 12 k cycles slower unloaded: profiles/r06_m_prepass_ab.txt)
// (Round 6, measured and removed: the first 6 tokens of every entry's / field's list fetched together ahead of the inserts - one
//  trip per group instead of one per entry.  No gain under load (c2 assembly 0.206 vs 0.208 ms), the diversity section 12 k cycles
//  SLOWER on an unloaded request, +25 KB of unrolled insert loops: profiles/r06_m_prepass_ab.txt.)
constexpr int PREP_INTS = 96;       // ints of LDS scratch: wave_tot[PREP_WAVES][PREP_GROUP] | first[FUSED_MAX_PREP] | misc[4] | tokens[DIV_GROUP]

struct PrepScratch {   // LDS scratch of one workgroup
  double *vals;        // vals_cap doubles: diversity median
  int vals_cap;
  int *ints;           // PREP_INTS
  mutable unsigned long long clk;  // MRK_PHASE_CLOCKS builds: time of the previous phase boundary
  mutable unsigned long long acc[6];
  __device__ __forceinline__ int *wave_tot() const { return ints; }                              // [PREP_WAVES][PREP_GROUP]
  __device__ __forceinline__ int *first() const { return ints + PREP_WAVES * PREP_GROUP; }       // [FUSED_MAX_PREP]
  __device__ __forceinline__ int *misc() const { return ints + PREP_WAVES * PREP_GROUP + FUSED_MAX_PREP; }   // [4]
  __device__ __forceinline__ int *tokens() const { return misc() + 4; }                          // [DIV_GROUP]
};

// exclusive prefix sum of a 0/1 flag over the workgroup + total (blockDim.x <= 64 * PREP_WAVES)
__device__ __forceinline__ int block_scan_flag(bool flag, int *s_wave_tot, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_waves = (blockDim.x + 63) >> 6;
  const unsigned long long ball = __ballot(flag);
  const int within = __popcll(ball & ((1ull << lane) - 1ull));
  if (lane == 0) s_wave_tot[wave] = __popcll(ball);
  __syncthreads();
  int before = 0;
  total = 0;
  for (int w = 0; w < n_waves; ++w) {
    int t = s_wave_tot[w];
    if (w < wave) before += t;
    total += t;
  }
  __syncthreads();
  return before + within;
}

// the same for PREP_GROUP flags at once (two barriers for the whole group)
__device__ __forceinline__ void block_scan_flags(const bool (&flag)[PREP_GROUP], int n, int *s_wave_tot, int (&excl)[PREP_GROUP],
                                                 int (&total)[PREP_GROUP]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_waves = (blockDim.x + 63) >> 6;
  int within[PREP_GROUP];
#pragma unroll
  for (int u = 0; u < PREP_GROUP; ++u) {
    const unsigned long long ball = __ballot(u < n && flag[u]);
    within[u] = __popcll(ball & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_tot[wave * PREP_GROUP + u] = __popcll(ball);
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PREP_GROUP; ++u) {
    int before = 0, tot = 0;
    for (int w = 0; w < n_waves; ++w) {
      const int t = s_wave_tot[w * PREP_GROUP + u];
      if (w < wave) before += t;
      tot += t;
    }
    excl[u] = before + within[u];
    total[u] = tot;
  }
  __syncthreads();
}

// commons-math Percentile (LEGACY estimation, NaN removed) .evaluate(50) of s_vals[0, n_raw); whole workgroup
__device__ double median_of(double *s_vals, int n_raw, int *s_misc) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (n_raw == 1) return s_vals[0];
  if (n_raw <= nthr) {
    // one value per thread: its rank among the others is its place in the sorted array (two barriers instead of a
    // compare-exchange network's log^2 n); NaNs are counted and left out (NaNStrategy.REMOVED)
    const bool mine = tid < n_raw;
    const double v = mine ? s_vals[tid] : 0.0;
    const bool isn = v != v;
    if (mine && isn) atomicAdd(&s_misc[1], 1);
    int rank = 0;
    if (mine && !isn)
      for (int j = 0; j < n_raw; ++j) {
        const double w = s_vals[j];  // the same address in every lane: a broadcast read
        rank += (w < v || (w == v && j < tid)) ? 1 : 0;
      }
    __syncthreads();
    if (mine && !isn) s_vals[rank] = v;
    __syncthreads();
  } else {
    // NaN -> +inf placeholder (sorts last), counted
    for (int i = tid; i < n_raw; i += nthr) {
      double v = s_vals[i];
      if (v != v) { s_vals[i] = __longlong_as_double(0x7ff0000000000000LL); atomicAdd(&s_misc[1], 1); }
    }
    int p2 = 1;
    while (p2 < n_raw) p2 <<= 1;
    for (int i = n_raw + tid; i < p2; i += nthr) s_vals[i] = __longlong_as_double(0x7ff0000000000000LL);
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < p2; i += nthr) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const double a = s_vals[i], c2 = s_vals[ixj];
            const bool up = (i & k) == 0;
            if ((a > c2) == up) { s_vals[i] = c2; s_vals[ixj] = a; }
          }
        }
        __syncthreads();
      }
    }
  }
  const int m = n_raw - s_misc[1];
  if (m <= 0) return d_nan();
  const double pos = 0.5 * (double)(m + 1);
  const double fpos = floor(pos);
  const int ipos = (int)fpos;
  const double dif = pos - fpos;
  if (pos < 1.0) return s_vals[0];
  if (pos >= (double)m) return s_vals[m - 1];
  const double lower = s_vals[ipos - 1], upper = s_vals[ipos];
  return __dadd_rn(lower, __dmul_rn(dif, __dsub_rn(upper, lower)));
}

// ---- The sections of a request's pre-pass on DIFFERENT wavefronts of its workgroup (round 4: c2 assembly -2 %, and
// -11 % on a single request's latency: the sections are most of an unloaded request's critical path).
// Measured on the round-3 kernel (tools/phase_clocks.py c2 32): the interacted_with histograms are 37 k cycles of a request's
// 225 k, the diversity sections 28 k + 37 k + 8 k, one after the other - and in both only ONE wavefront has work (50
// interacted items, the first `top` = 20 candidates) while the other waits at the section's barriers.  Here wavefront 0
// runs the whole diversity section with wave-local scans (ballots instead of LDS totals + two barriers per scan; lanes
// of one wavefront talk through LDS in program order) while the other wavefronts build the interacted_with tables; one
// barrier at the end (same box, round 4: 244 k -> 205 k cycles).  Tables of different entries are disjoint, the histograms do
// not depend on insertion order: same results.  Taken when every diversity entry looks at no more than 64 candidates' values
// (`top` <= 64: the default is 20); otherwise the workgroup-wide code below runs (a program property: in a specialised kernel
// only one of the two is compiled).

// (wave_lds_sync, wave_scan_flag, wave_median_of: wave_device.hpp - compiled for the host too, tests/native/wave_test.cpp)

// (a function of the program alone: in a specialised kernel the other pre-pass is not even compiled)
template <typename Prog>
__device__ __forceinline__ bool prepass_waves_ok(const Prog &prog) {
  bool ok = true;
  for (int e = 0; e < prog.n_prep; ++e)
    if (prog.prep[e].kind == PREP_DIVERSITY && prog.prep[e].top > 64) ok = false;
  return ok;
}

// the interacted_with section of prepass_request, run by lanes `lane_id` of `n_lanes` (nothing in it synchronises)
template <typename Prog>
__device__ __forceinline__ void prepass_interacted_with(const StoreDev &st, const Prog &prog, const BatchDev &b, int r, const ReqDev &rq,
                                                        unsigned long long *tab_base, uint32_t tab_sub, PrepOut *po_out, int lane_id, int n_lanes) {
  const int n_prep = prog.n_prep;
  for (int e0 = 0; e0 < n_prep;) {
    const PrepEntry pe0 = prog.prep[e0];
    if (pe0.kind != PREP_IW_FIELD) { ++e0; continue; }
    int n = 1;  // consecutive entries on the same bounded list
    while (n < PREP_GROUP && e0 + n < n_prep) {
      const PrepEntry q = prog.prep[e0 + n];
      if (q.kind != PREP_IW_FIELD || q.list_scope != pe0.list_scope || q.list_col.tag != pe0.list_col.tag || q.list_col.val != pe0.list_col.val) break;
      ++n;
    }
    ColRef col[PREP_GROUP];
    unsigned long long *tab[PREP_GROUP];
    uint32_t cap[PREP_GROUP];
#pragma unroll
    for (int u = 0; u < PREP_GROUP; ++u) {
      const int e = e0 + (u < n ? u : 0);
      col[u] = prog.prep[e].item_col;
      tab[u] = tab_base + (po_out[e].tab_off - tab_sub);
      cap[u] = po_out[e].tab_cap;
    }
    const int vslot = pe0.list_scope == SC_SESSION ? rq.session_slot : rq.user_slot;
    const Cell lc = load_cell(record(st, pe0.list_scope, vslot), pe0.list_col);
    if (lc.tag != TAG_MISSING) {
      const uint32_t off = lc.lo(), len = lc.hi();
      for (uint32_t k = (uint32_t)lane_id; k < len; k += (uint32_t)n_lanes) {
        const uint8_t *irec = record(st, SC_ITEM, (int)st.slot_pool[off + k]);
        Cell ic[PREP_GROUP];
#pragma unroll
        for (int u = 0; u < PREP_GROUP; ++u) {  // the field cells of all entries: independent loads
          ic[u].tag = TAG_MISSING;
          ic[u].bits = 0;
          if (u < n) ic[u] = load_cell(irec, col[u]);
        }
#pragma unroll
        for (int u = 0; u < PREP_GROUP; ++u) {
          if (u < n) {  // uniform
            const bool list = ic[u].tag == TAG_STRING_LIST;
            if (table_add_list(list_tokens(st, irec, ic[u].lo()), tab[u], cap[u], list ? ic[u].hi() : 0u)) atomicOr(&b.status[r], ST_TABLE_FULL);
          }
        }
      }
    }
    e0 += n;
  }


}

// the diversity section of prepass_request, run by ONE wavefront (all 64 lanes)
template <typename Prog>
__device__ __forceinline__ void prepass_diversity_wave(const StoreDev &st, const Prog &prog, const BatchDev &b, int r, const ReqDev &rq,
                                                       unsigned long long *tab_base, uint32_t tab_sub, PrepOut *po_out, const PrepScratch &sc) {
  const int lane = threadIdx.x & 63;
  const int n_prep = prog.n_prep;
  int *s_first = sc.first();
  int *s_tokens = sc.tokens();
#ifdef MRK_PHASE_CLOCKS
  unsigned long long dv_t = clock64(), dv_acc[3] = {0, 0, 0};
#endif
  for (int e0 = 0; e0 < n_prep;) {
    if (prog.prep[e0].kind != PREP_DIVERSITY || po_out[e0].preset) { ++e0; continue; }
    int ent[DIV_GROUP];
    int n = 0, e1 = e0;
    for (; e1 < n_prep && n < DIV_GROUP; ++e1) {
      if (prog.prep[e1].kind != PREP_DIVERSITY || po_out[e1].preset) continue;
#pragma unroll
      for (int u = 0; u < DIV_GROUP; ++u) if (u == n) ent[u] = e1;
      ++n;
    }
    ColRef col[DIV_GROUP];
    unsigned long long *tab[DIV_GROUP];
    uint32_t cap[DIV_GROUP];
    int top[DIV_GROUP];
#pragma unroll
    for (int u = 0; u < DIV_GROUP; ++u) {
      if (u >= n) ent[u] = e0;
      col[u] = prog.prep[ent[u]].item_col;
      top[u] = prog.prep[ent[u]].top;
      tab[u] = tab_base + (po_out[ent[u]].tab_off - tab_sub);
      cap[u] = po_out[ent[u]].tab_cap;
    }
    // (a) first candidate with state, per entry; the cells of the first 64 candidates stay in registers for (b) and (c)
    const uint8_t *keep_rec = nullptr;
    Cell keep[DIV_GROUP];
#pragma unroll
    for (int u = 0; u < DIV_GROUP; ++u) { keep[u].tag = TAG_MISSING; keep[u].bits = 0; }
    for (int base = 0; base < rq.n_items; base += 64) {
      const int i = base + lane;
      if (i < rq.n_items) {
        const uint8_t *irec = record(st, SC_ITEM, b.item_slot[rq.item_begin + i]);
        Cell c[DIV_GROUP];
#pragma unroll
        for (int u = 0; u < DIV_GROUP; ++u) {
          c[u].tag = TAG_MISSING;
          c[u].bits = 0;
          if (u < n) c[u] = load_cell(irec, col[u]);
        }
#pragma unroll
        for (int u = 0; u < DIV_GROUP; ++u) {
          if (u < n && c[u].tag != TAG_MISSING) atomicMin(&s_first[ent[u]], (i << 8) | (int)c[u].tag);
          if (base == 0) keep[u] = c[u];
        }
        if (base == 0) keep_rec = irec;
      }
      wave_lds_sync();
      bool all = true;
#pragma unroll
      for (int u = 0; u < DIV_GROUP; ++u) all = all && (u >= n || s_first[ent[u]] != 0x7fffffff);
      wave_lds_sync();
      if (all) break;
    }
    int mode[DIV_GROUP];
#pragma unroll
    for (int u = 0; u < DIV_GROUP; ++u) {
      const int first = u < n ? s_first[ent[u]] : 0x7fffffff;
      const int htag = first != 0x7fffffff ? (first & 255) : (int)TAG_MISSING;
      mode[u] = (htag == TAG_STRING || htag == TAG_STRING_LIST) ? DIV_STRING : (htag == TAG_DOUBLE ? DIV_DOUBLE : DIV_EMPTY);
    }
    MRK_PHASE(dv_t, dv_acc[0]);
    // (b) string entries: the first `top` candidates of that type, in request order
    bool any_string = false;
#pragma unroll
    for (int u = 0; u < DIV_GROUP; ++u) any_string = any_string || (u < n && mode[u] == DIV_STRING);
    if (lane < DIV_GROUP) s_tokens[lane] = 0;
    wave_lds_sync();
    if (any_string) {
      int running[DIV_GROUP];
#pragma unroll
      for (int u = 0; u < DIV_GROUP; ++u) running[u] = 0;
      for (int base = 0; base < rq.n_items; base += 64) {
        bool more = false;
#pragma unroll
        for (int u = 0; u < DIV_GROUP; ++u) more = more || (u < n && mode[u] == DIV_STRING && running[u] < top[u]);
        if (!more) break;
        const int i = base + lane;
        Cell c[DIV_GROUP];
        bool cand[DIV_GROUP];
        const uint8_t *irec = base == 0 ? keep_rec : (i < rq.n_items ? record(st, SC_ITEM, b.item_slot[rq.item_begin + i]) : nullptr);
#pragma unroll
        for (int u = 0; u < DIV_GROUP; ++u) {
          c[u].tag = TAG_MISSING;
          c[u].bits = 0;
          if (u < n && mode[u] == DIV_STRING) c[u] = base == 0 ? keep[u] : load_cell(irec, col[u]);
          cand[u] = c[u].tag == TAG_STRING || c[u].tag == TAG_STRING_LIST;
        }
#pragma unroll
        for (int u = 0; u < DIV_GROUP; ++u) {
          int total = 0;
          const int excl = wave_scan_flag(u < n && cand[u], total);
          if (u < n && mode[u] == DIV_STRING) {  // uniform
            const bool take = cand[u] && running[u] + excl < top[u];
            const bool one = take && c[u].tag == TAG_STRING;
            const uint32_t tlen = take && !one ? c[u].hi() : 0u;
            uint32_t failed = table_add(tab[u], cap[u], c[u].lo(), one) ? 0u : 1u;
            failed += table_add_list(list_tokens(st, irec, c[u].lo()), tab[u], cap[u], tlen);
            if (failed) atomicOr(&b.status[r], ST_TABLE_FULL);
            if (take) atomicAdd(&s_tokens[u], one ? 1 : (int)tlen);
          }
          running[u] += total;
        }
      }
      wave_lds_sync();
    }
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < DIV_GROUP; ++u)
        if (u < n && mode[u] != DIV_DOUBLE) {
          po_out[ent[u]].mode = mode[u];
          po_out[ent[u]].scalar = mode[u] == DIV_STRING ? (double)s_tokens[u] : 0.0;
        }
    }
    wave_lds_sync();   // (the next group zeroes s_tokens)
    MRK_PHASE(dv_t, dv_acc[1]);
    // (c) numeric entries: the first `top` (<= 64) present values in request order, then their median
#pragma unroll
    for (int u = 0; u < DIV_GROUP; ++u) {
      if (u >= n || mode[u] != DIV_DOUBLE) continue;
      int running = 0;
      for (int base = 0; base < rq.n_items && running < top[u]; base += 64) {
        const int i = base + lane;
        Cell c;
        c.tag = TAG_MISSING;
        c.bits = 0;
        if (base == 0) c = keep[u];
        else if (i < rq.n_items) c = load_cell(record(st, SC_ITEM, b.item_slot[rq.item_begin + i]), col[u]);
        const bool cand = c.tag == TAG_DOUBLE;
        int total;
        const int rank = running + wave_scan_flag(cand, total);
        if (cand && rank < top[u]) {
          if (rank < sc.vals_cap) sc.vals[rank] = c.f64();
          else atomicOr(&b.status[r], ST_TOO_MANY);
        }
        running += total;
      }
      wave_lds_sync();
      const double scalar = wave_median_of(sc.vals, min(min(running, top[u]), min(sc.vals_cap, 64)));
      if (lane == 0) {
        po_out[ent[u]].mode = DIV_DOUBLE;
        po_out[ent[u]].scalar = scalar;
      }
      wave_lds_sync();
    }
    MRK_PHASE(dv_t, dv_acc[2]);
    e0 = e1;
  }
#ifdef MRK_PHASE_CLOCKS
  if (threadIdx.x == 0)
    for (int i = 0; i < 3; ++i) atomicAdd(&mrk_phase_clocks[8 + i], dv_acc[i]);
#endif
}

// The pre-pass of request r, run by one whole workgroup.  Tables live at tab_base + (po.tab_off - tab_sub)
// (HBM arena: tab_sub = 0; LDS: tab_sub = the request's first arena entry); mode / scalar go to po_out[e].
// The entries are not processed one by one: every global load is a trip to the Infinity Cache, so the loads
// of up to PREP_GROUP entries are issued together -
//   * all tables of the request are zeroed in one sweep;
//   * the interacted_with entries that read the same bounded list (one per field) share ONE pass over the
//     interacted items: slot -> the field cells of all entries -> their tokens;
//   * the diversity entries share the pass that finds each one's first candidate with state, the load that
//     decides string vs number, and - for the string ones - the pass over the first `top` candidates
//     (one multi-flag prefix scan keeps request order); numeric ones then take their median one by one.
template <typename Prog>
__device__ __forceinline__ void prepass_request(const StoreDev &st, const Prog &prog, const BatchDev &b, int r, const ReqDev &rq,
                                unsigned long long *tab_base, uint32_t tab_sub, PrepOut *po_out, const PrepScratch &sc) {
  const int tid = threadIdx.x;
  const int nthr = blockDim.x;
  const int n_prep = prog.n_prep;
  int *s_first = sc.first();
  int *s_misc = sc.misc();
  int *s_tokens = sc.tokens();

  // ---- all tables of the request, one sweep (they are contiguous: host assigns them in entry order)
  {
    uint32_t lo = 0xffffffffu, hi = 0;
    for (int e = 0; e < n_prep; ++e) {
      const uint32_t o = po_out[e].tab_off - tab_sub;
      lo = min(lo, o);
      hi = max(hi, o + po_out[e].tab_cap);
    }
    for (uint32_t i = lo + tid; i < hi; i += nthr) tab_base[i] = 0ull;
    for (int e = tid; e < n_prep; e += nthr) s_first[e] = 0x7fffffff;
  }
  __syncthreads();
  MRK_PHASE(sc.clk, sc.acc[0]);
  if (prepass_waves_ok(prog)) {   // (uniform) the sections side by side on different wavefronts
    const bool lone = nthr <= 64;   // a one-wavefront workgroup runs both, one after the other
    const int wave = tid >> 6;
    if (lone || wave != 0) prepass_interacted_with(st, prog, b, r, rq, tab_base, tab_sub, po_out, lone ? tid : tid - 64, lone ? nthr : nthr - 64);
    if (wave == 0) prepass_diversity_wave(st, prog, b, r, rq, tab_base, tab_sub, po_out, sc);
#ifdef MRK_PHASE_CLOCKS
    if (tid == 64) atomicAdd(&mrk_phase_clocks[2], clock64() - sc.clk);   // the interacted_with section alone (wavefront 1)
    if (tid == 0) atomicAdd(&mrk_phase_clocks[3], clock64() - sc.clk);    // the diversity section alone (wavefront 0)
#endif
    __syncthreads();
    MRK_PHASE(sc.clk, sc.acc[1]);
    return;
  }

  // Per-group state lives in registers indexed at COMPILE time (every loop over the group is fully unrolled
  // and predicated on u < n): a run-time index into a register array costs a select chain per access.

  // ---- interacted_with (InteractedWithFeature.scala:134-147): histogram of the field tokens of every interacted item
  for (int e0 = 0; e0 < n_prep;) {
    const PrepEntry pe0 = prog.prep[e0];
    if (pe0.kind != PREP_IW_FIELD) { ++e0; continue; }
    int n = 1;  // consecutive entries on the same bounded list
    while (n < PREP_GROUP && e0 + n < n_prep) {
      const PrepEntry q = prog.prep[e0 + n];
      if (q.kind != PREP_IW_FIELD || q.list_scope != pe0.list_scope || q.list_col.tag != pe0.list_col.tag || q.list_col.val != pe0.list_col.val) break;
      ++n;
    }
    ColRef col[PREP_GROUP];
    unsigned long long *tab[PREP_GROUP];
    uint32_t cap[PREP_GROUP];
#pragma unroll
    for (int u = 0; u < PREP_GROUP; ++u) {
      const int e = e0 + (u < n ? u : 0);
      col[u] = prog.prep[e].item_col;
      tab[u] = tab_base + (po_out[e].tab_off - tab_sub);
      cap[u] = po_out[e].tab_cap;
    }
    const int vslot = pe0.list_scope == SC_SESSION ? rq.session_slot : rq.user_slot;
    const Cell lc = load_cell(record(st, pe0.list_scope, vslot), pe0.list_col);
    if (lc.tag != TAG_MISSING) {
      const uint32_t off = lc.lo(), len = lc.hi();
      for (uint32_t k = tid; k < len; k += nthr) {
        const uint8_t *irec = record(st, SC_ITEM, (int)st.slot_pool[off + k]);
        Cell ic[PREP_GROUP];
#pragma unroll
        for (int u = 0; u < PREP_GROUP; ++u) {  // the field cells of all entries: independent loads
          ic[u].tag = TAG_MISSING;
          ic[u].bits = 0;
          if (u < n) ic[u] = load_cell(irec, col[u]);
        }
#pragma unroll
        for (int u = 0; u < PREP_GROUP; ++u) {
          if (u < n) {  // uniform
            const bool list = ic[u].tag == TAG_STRING_LIST;
            if (table_add_list(list_tokens(st, irec, ic[u].lo()), tab[u], cap[u], list ? ic[u].hi() : 0u)) atomicOr(&b.status[r], ST_TABLE_FULL);
          }
        }
      }
    }
    e0 += n;
  }

  MRK_PHASE(sc.clk, sc.acc[1]);
  // ---- diversity (DiversityFeature.scala:72-103), PREP_GROUP entries at a time
  for (int e0 = 0; e0 < n_prep;) {
    if (prog.prep[e0].kind != PREP_DIVERSITY || po_out[e0].preset) { ++e0; continue; }  // (preset: the host took this entry's median)
    int ent[PREP_GROUP];
    int n = 0, e1 = e0;
    for (; e1 < n_prep && n < PREP_GROUP; ++e1) {
      if (prog.prep[e1].kind != PREP_DIVERSITY || po_out[e1].preset) continue;
#pragma unroll
      for (int u = 0; u < PREP_GROUP; ++u) if (u == n) ent[u] = e1;
      ++n;
    }
    ColRef col[PREP_GROUP];
    unsigned long long *tab[PREP_GROUP];
    uint32_t cap[PREP_GROUP];
    int top[PREP_GROUP];
#pragma unroll
    for (int u = 0; u < PREP_GROUP; ++u) {
      if (u >= n) ent[u] = e0;
      col[u] = prog.prep[ent[u]].item_col;
      top[u] = prog.prep[ent[u]].top;
      tab[u] = tab_base + (po_out[ent[u]].tab_off - tab_sub);
      cap[u] = po_out[ent[u]].tab_cap;
    }
    // (a) the first candidate that has a ScalarValue decides string vs number - for every entry of the group:
    // atomicMin over (index << 8 | tag) finds the first one AND what it holds.  A request that fits one round of the
    // workgroup (the usual case) keeps its cells in registers for passes (b) and (c).
    const bool single = rq.n_items <= nthr;
    const uint8_t *keep_rec = nullptr;  // ... and the record they came from (inline string lists are read from it)
    Cell keep[PREP_GROUP];
#pragma unroll
    for (int u = 0; u < PREP_GROUP; ++u) { keep[u].tag = TAG_MISSING; keep[u].bits = 0; }
    for (int base = 0; base < rq.n_items; base += nthr) {
      const int i = base + tid;
      if (i < rq.n_items) {
        const uint8_t *irec = record(st, SC_ITEM, b.item_slot[rq.item_begin + i]);
        Cell c[PREP_GROUP];
#pragma unroll
        for (int u = 0; u < PREP_GROUP; ++u) {
          c[u].tag = TAG_MISSING;
          c[u].bits = 0;
          if (u < n) c[u] = load_cell(irec, col[u]);
        }
#pragma unroll
        for (int u = 0; u < PREP_GROUP; ++u) {
          if (u < n && c[u].tag != TAG_MISSING) atomicMin(&s_first[ent[u]], (i << 8) | (int)c[u].tag);
          if (single) keep[u] = c[u];
        }
        if (single) keep_rec = irec;
      }
      __syncthreads();
      bool all = true;
#pragma unroll
      for (int u = 0; u < PREP_GROUP; ++u) all = all && (u >= n || s_first[ent[u]] != 0x7fffffff);
      __syncthreads();  // nobody may start the next round's atomicMin before everyone has read
      if (all) break;
    }
    int mode[PREP_GROUP];
#pragma unroll
    for (int u = 0; u < PREP_GROUP; ++u) {
      const int first = u < n ? s_first[ent[u]] : 0x7fffffff;
      const int htag = first != 0x7fffffff ? (first & 255) : (int)TAG_MISSING;
      mode[u] = (htag == TAG_STRING || htag == TAG_STRING_LIST) ? DIV_STRING : (htag == TAG_DOUBLE ? DIV_DOUBLE : DIV_EMPTY);
    }
    MRK_PHASE(sc.clk, sc.acc[2]);
    // (b) string entries: the first `top` candidates of that type, in request order, all entries in one pass
    bool any_string = false;
#pragma unroll
    for (int u = 0; u < PREP_GROUP; ++u) any_string = any_string || (u < n && mode[u] == DIV_STRING);
    if (tid < PREP_GROUP) s_tokens[tid] = 0;
    __syncthreads();
    if (any_string) {
      int running[PREP_GROUP];
#pragma unroll
      for (int u = 0; u < PREP_GROUP; ++u) running[u] = 0;
      for (int base = 0; base < rq.n_items; base += nthr) {
        bool more = false;
#pragma unroll
        for (int u = 0; u < PREP_GROUP; ++u) more = more || (u < n && mode[u] == DIV_STRING && running[u] < top[u]);
        if (!more) break;
        const int i = base + tid;
        Cell c[PREP_GROUP];
        bool cand[PREP_GROUP];
        const uint8_t *irec = single ? keep_rec : (i < rq.n_items ? record(st, SC_ITEM, b.item_slot[rq.item_begin + i]) : nullptr);
#pragma unroll
        for (int u = 0; u < PREP_GROUP; ++u) {
          c[u].tag = TAG_MISSING;
          c[u].bits = 0;
          if (u < n && mode[u] == DIV_STRING) c[u] = single ? keep[u] : load_cell(irec, col[u]);
          cand[u] = c[u].tag == TAG_STRING || c[u].tag == TAG_STRING_LIST;
        }
        int excl[PREP_GROUP], total[PREP_GROUP];
        block_scan_flags(cand, n, sc.wave_tot(), excl, total);
#pragma unroll
        for (int u = 0; u < PREP_GROUP; ++u) {
          if (u < n && mode[u] == DIV_STRING) {  // uniform
            const bool take = cand[u] && running[u] + excl[u] < top[u];
            const bool one = take && c[u].tag == TAG_STRING;
            const uint32_t tlen = take && !one ? c[u].hi() : 0u;
            uint32_t failed = table_add(tab[u], cap[u], c[u].lo(), one) ? 0u : 1u;
            failed += table_add_list(list_tokens(st, irec, c[u].lo()), tab[u], cap[u], tlen);
            if (failed) atomicOr(&b.status[r], ST_TABLE_FULL);
            if (take) atomicAdd(&s_tokens[u], one ? 1 : (int)tlen);
          }
          running[u] += total[u];
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
#pragma unroll
      for (int u = 0; u < PREP_GROUP; ++u)
        if (u < n && mode[u] != DIV_DOUBLE) {
          po_out[ent[u]].mode = mode[u];
          // stringCounts.values.foldLeft(0.0)(_ + _): integers, exact in f64
          po_out[ent[u]].scalar = mode[u] == DIV_STRING ? (double)s_tokens[u] : 0.0;
        }
    }
    MRK_PHASE(sc.clk, sc.acc[3]);
    // (c) numeric entries, one at a time: the first `top` present values in request order, then their median
#pragma unroll
    for (int u = 0; u < PREP_GROUP; ++u) {
      if (u >= n || mode[u] != DIV_DOUBLE) continue;
      if (tid == 0) { s_misc[0] = 0; s_misc[1] = 0; }
      __syncthreads();
      int running = 0;
      for (int base = 0; base < rq.n_items && running < top[u]; base += nthr) {
        const int i = base + tid;
        Cell c;
        c.tag = TAG_MISSING;
        c.bits = 0;
        if (single) c = keep[u];
        else if (i < rq.n_items) c = load_cell(record(st, SC_ITEM, b.item_slot[rq.item_begin + i]), col[u]);
        const bool cand = c.tag == TAG_DOUBLE;
        int total;
        const int rank = running + block_scan_flag(cand, sc.wave_tot(), total);
        if (cand && rank < top[u]) {
          if (rank < sc.vals_cap) sc.vals[rank] = c.f64();
          else atomicOr(&b.status[r], ST_TOO_MANY);
        }
        running += total;
      }
      __syncthreads();
      const double scalar = median_of(sc.vals, min(min(running, top[u]), sc.vals_cap), s_misc);
      if (tid == 0) {
        po_out[ent[u]].mode = DIV_DOUBLE;
        po_out[ent[u]].scalar = scalar;
      }
      __syncthreads();
    }
    MRK_PHASE(sc.clk, sc.acc[4]);
    e0 = e1;
  }
  __syncthreads();
}

// The pre-pass alone, one workgroup per request (requests too large for one workgroup's assembly - config 4 - or whose tables
// exceed the fused kernel's LDS budget).  The item-parallel kernels that follow read the tables from the HBM arena, but a
// request's tables depend on its session and `top`, not on its candidate count: when they fit the `lds_entries` of dynamic LDS
// they are BUILT there (every insert / probe a `ds_cmpst` instead of a global atomic round trip) and copied out once.
// Two shapes: prepass_kernel (rank.hip) interprets the program; mrk_jit_prepass (jit.cpp) has it as constants.
template <typename Prog>
__device__ __forceinline__ void prepass_body(const StoreDev &st, const Prog &prog, const BatchDev &b, uint32_t lds_entries) {
  __shared__ double s_vals[PREP_MAX_VALUES];
  __shared__ int s_ints[PREP_INTS];
  extern __shared__ __align__(16) unsigned long long s_tables[];
  const int r = blockIdx.x;
  const ReqDev rq = b.reqs[r];
  if (rq.item_begin >= b.item_hi || rq.item_begin + rq.n_items <= b.item_lo) return;  // not in this shard
  PrepScratch sc{s_vals, PREP_MAX_VALUES, s_ints, 0ull, {0, 0, 0, 0, 0, 0}};
  PrepOut *po = &b.prep_out[(size_t)r * prog.n_prep];
  uint32_t n_ent = 0;  // this request's table entries: [arena_begin, arena_begin + n_ent)
  for (int e = 0; e < prog.n_prep; ++e) n_ent = max(n_ent, po[e].tab_off - rq.arena_begin + po[e].tab_cap);
  const bool in_lds = n_ent <= lds_entries;  // uniform
  if (!in_lds) {
    prepass_request(st, prog, b, r, rq, b.arena, 0u, po, sc);
  } else {  // (two instantiations: with a selected pointer the table accesses would be flat instead of ds operations)
    prepass_request(st, prog, b, r, rq, s_tables, rq.arena_begin, po, sc);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_ent; i += blockDim.x) b.arena[(size_t)rq.arena_begin + i] = s_tables[i];
  }
}

// ---------------------------------------------------------------- assemble
constexpr int ASM_THREADS = 256;

// java.lang.Math.round(double)
__device__ __forceinline__ long long java_round(double a) {
  if (a != a) return 0;
  if (a >= 9223372036854775807.0) return 0x7fffffffffffffffLL;
  if (a <= -9223372036854775808.0) return (long long)0x8000000000000000ULL;
  if (fabs(a) >= 4503599627370496.0) return (long long)a;
  const double fl = floor(a);
  return (long long)fl + ((a - fl) >= 0.5 ? 1 : 0);
}

// Scala Long / Long (truncating; Long.MinValue / -1 wraps); the zero divisor is reported by the caller
__device__ __forceinline__ long long long_div(long long a, long long b) {
  if (b == -1) return (long long)(0ull - (unsigned long long)a);
  return a / b;
}

// (Long / Long).toDouble - what the normalised rate does with the GLOBAL counters (RateFeature.scala:334-350).  The operands
// are request-level, so the compiler scalarises the 64-bit division: ~300 scalar instructions per period and WAVEFRONT (3 % of
// the c2 kernel's scalar instructions).  Counters are counts: for 0 <= a < 2^52, 0 < b < 2^52 the quotient is
// trunc(fl(a / b)) exactly - a division the vector unit has.  Proof: let k = floor(a / b) <= a < 2^52; if b divides a the
// quotient is exact; else a / b <= k + 1 - 1 / b, and fl() could reach k + 1 only if 1 / b <= half the spacing of doubles
// below k + 1, i.e. b >= 2^(53 - e) with 2^e <= k + 1 - which puts a >= k b >= 2^52.  Anything else takes the integer path.
__device__ __forceinline__ double long_div_to_double(long long a, long long b) {
  if ((unsigned long long)a < (1ull << 52) && (unsigned long long)b < (1ull << 52)) return trunc((double)a / (double)b);
  return (double)long_div(a, b);
}

// ---- sinks: where an assembled value goes.  A sink is driven by whole wavefronts: lanes without an item
// (`active` false) run the same program on a missing record and write nothing.
struct MatrixSink {   // row-major f64 matrix, ClickthroughQuery's layout
  double *row;
  bool active;
  __device__ __forceinline__ void begin() const {}
  __device__ __forceinline__ void finish() const {}
  __device__ __forceinline__ void put(int col, double v) const {
    if (active) row[col] = v;
  }
};

// What a sink knows about the forest at compile time.  QsDyn: nothing - every column's descriptor (QsFeature) is read from
// memory, a column ahead, by scalar loads.  A kernel specialised for a model (jit.cpp) is also keyed by the forest's view
// signature and passes a type with `is_static`, `n_feats`, `n_views`, `thr_cap` and a constexpr `QsSig operator[](col)`:
// descriptor loads, view-kind decoding, the view loop, the staging loop and the in-order wait all fold into straight-line
// code with immediate offsets (they were 9 k of the stock kernel's 54 k instructions - and, being run once per column
// and wavefront, as many dynamic ones, two thirds of them scalar).
struct QsDyn { static constexpr bool is_static = false; };
template <typename QS> __device__ __forceinline__ uint32_t qs_thr_cap(const QsDev &q) {
  if constexpr (QS::is_static) return QS::thr_cap;
  else return q.thr_cap;
}
template <typename QS> __device__ __forceinline__ int qs_n_views(const QsDev &q) {
  if constexpr (QS::is_static) return QS::n_views;
  else return q.n_views;
}

template <bool F64, typename QS = QsDyn>
struct CellSink {     // the scorer's binned tile: [tile of 128 rows][view][row] u16
  QsDev q;
  uint16_t *dst;      // &cells[tile][0][row]
  int32_t *status;    // the request's status word
  qs_lds_double *thr_lds;  // two staging buffers of thr_cap() doubles, private to this wavefront
  bool active;
  // false when `dst` is LDS (the one-launch kernels keep the request's tile there): its stores are ds_writes, which vmcnt does
  // not count at all - the in-order wait below must then never count them (round 4: the one-launch kernel waited vmcnt(1 | 2)
  // for a table that was still on its way and binned against a half-landed table once in a few runs - found when the
  // signature-keyed sink shortened the time between request and search; the batch kernels, whose cells go to global memory,
  // were never affected)
  bool vm_stores = true;
  // A column's threshold table is searched in LDS (a per-lane binary search in global memory would be log2(T)
  // scattered wave-loads per column).  Asking for the descriptor, then for the table, then searching costs two trips
  // to memory per column - measured: 70 % of the assembly phase - so the tables travel one column ahead of the search:
  // columns arrive in increasing order; while column c is searched in buffer c & 1, the table of c + 1 is on its way
  // into the other buffer (global_load_lds: no registers, no ds_write) and the descriptor of c + 2 into scalar
  // registers (constant address space: s_load).
  mutable QsFeature ft_cur = {}, ft_next = {};   // feats[next_col], feats[next_col + 1]   (QsDyn only)
  mutable int next_col = -1;                     // the column whose table has been requested
  mutable uint32_t newer = 0;                    // cell stores this wavefront issued AFTER that request (vmcnt retires in order)

  __device__ __forceinline__ uint32_t thr_cap() const { return qs_thr_cap<QS>(q); }
  __device__ __forceinline__ int n_feats() const {
    if constexpr (QS::is_static) return QS::n_feats;
    else return q.n_feats;
  }
  __device__ __forceinline__ bool staged(const QsFeature &ft) const { return ft.thr_len <= thr_cap() && ft.view_begin != ft.view_end; }
  __device__ __forceinline__ QsFeatureK desc(int col) const { return (QsFeatureK)(unsigned long long)q.feats + col; }
  __device__ __forceinline__ QsFeature feature(int col) const {
    QsFeature ft = {};
    if (col < n_feats()) {
      if constexpr (QS::is_static) {
        // constants; thr_len is the table's length in whole chunks when it is staged (the staged search and the staging loop
        // treat the +inf padding as the table's end) - the exact length is read where it matters (real_len below)
        const QsSig s = QS{}[col];
        ft.thr_off = s.thr_off;
        ft.view_begin = s.view_begin;
        ft.view_end = s.view_end;
        ft.view_kinds = s.view_kinds;
        const uint32_t padded = (uint32_t)s.chunks * QS_STAGE_CHUNK;
        ft.thr_len = (padded <= QS::thr_cap || s.view_begin == s.view_end) ? (uint16_t)padded : desc(col)->thr_len;
      } else {
        const QsFeatureK f = desc(col);
        ft.thr_off = f->thr_off;
        ft.thr_len = f->thr_len;
        ft.zero_bin = f->zero_bin;
        ft.view_begin = f->view_begin;
        ft.view_end = f->view_end;
        ft.view_kinds = f->view_kinds;
      }
    }
    return ft;
  }
  __device__ __forceinline__ void request(const QsFeature &ft, int col) const {  // table of `col` -> buffer col & 1
    if (!staged(ft)) return;
    // lane l: entries 2 l, 2 l + 1 of the chunk (runs past the table's end into the next one / the slack after the last).
    // Address = a uniform base (kernel argument + the column's offset) + the lane's byte offset, the latter behind an empty
    // `asm volatile`: with every offset a constant the addresses are invariant across the rounds of the item loop, and the
    // compiler computed all of them up front - 2 registers per column, 44 spilled VGPRs - instead of one add per request.
    uint32_t lane_off = (threadIdx.x & 63u) * 16u;
    if constexpr (QS::is_static) asm volatile("" : "+v"(lane_off));
    qs_lds_double *buf = thr_lds + (size_t)(col & 1) * thr_cap();
    if constexpr (QS::is_static) {
#pragma unroll
      for (uint32_t k0 = 0; k0 < ft.thr_len; k0 += QS_STAGE_CHUNK)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)(q.thr + ft.thr_off + k0) + lane_off),
                                         (__attribute__((address_space(3))) void *)(buf + k0), 16, 0, 0);
    } else {
#pragma unroll 1
      for (uint32_t k0 = 0; k0 < ft.thr_len; k0 += QS_STAGE_CHUNK)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)(q.thr + ft.thr_off + k0) + lane_off),
                                         (__attribute__((address_space(3))) void *)(buf + k0), 16, 0, 0);
    }
  }
  // before the first put of an item / after its last
  __device__ __forceinline__ void begin() const { restart(0); }
  __device__ __forceinline__ void finish() const {}
  __device__ __forceinline__ void restart(int col) const {
    ft_cur = feature(col);
    if constexpr (!QS::is_static) ft_next = feature(col + 1);
    next_col = col;
    newer = 0;
    request(ft_cur, col);
  }
  // The requested table is in its buffer.  vmcnt counts loads, stores and LDS-DMA alike and retires them IN ORDER: once at
  // most `newer` operations are outstanding, everything issued before the last `newer` - the request among it - is
  // complete.  Waiting for vmcnt(0) instead (round 2) made every column wait for the previous column's cell stores as
  // well: a store's round trip per matrix column on the critical path of the workgroup.  (Loads the ops issue between two
  // columns only make this wait stricter, never too weak.)
  __device__ __forceinline__ void wait_landed() const {
    if (newer == 1u) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (newer == 2u) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (newer == 3u) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler does not track LDS-DMA
    __builtin_amdgcn_wave_barrier();
  }

  // `col` is uniform across the wavefront; every lane of the wavefront takes part (lanes without an item write nothing)
  __device__ __forceinline__ void put(int col, double v) const {
    if (col >= n_feats()) {  // a column the forest does not know: nothing to bin, but still part of XGBoost's DMatrix row
      if constexpr (!F64) {
        bool fin;
        (void)qs_prep<F64>(v, fin);
        if (!fin && active) atomicOr(status, ST_XGB_INF);
      }
      return;
    }
    if (col != next_col) restart(col);  // a column out of order
    const QsFeature ft = QS::is_static ? feature(col) : ft_cur;
    if (!QS::is_static || staged(ft)) wait_landed();   // this column's table (requested one column ago, before that column's cell stores)
    // the next column's table into the other buffer: the search that used it is over.  `newer` restarts with every request
    // issued (a static signature knows which columns have no staged table: nothing is requested, nothing waited for)
    if constexpr (QS::is_static) {
      const QsFeature nx = feature(col + 1);
      request(nx, col + 1);
      if (staged(nx)) newer = 0;
    } else {
      ft_cur = ft_next;
      request(ft_cur, col + 1);
      ft_next = feature(col + 2);
      newer = 0;
    }
    next_col = col + 1;
    bool ok;
    const double x = qs_prep<F64>(v, ok);
    // XGBoost's DMatrix rejects the whole row for an inf in ANY column, split on or not - every scorer path does the same
    if (!ok && active) atomicOr(status, ST_XGB_INF);
    if (ft.view_begin == ft.view_end) return;  // the forest never splits on this column
    const QsFeatureK f = desc(col);
    const uint32_t pos = staged(ft) ? qs_bin_search_staged<F64>(thr_lds + (size_t)(col & 1) * thr_cap(), ft.thr_len, x,
                                                                [&]() -> uint32_t { return QS::is_static ? f->thr_len : ft.thr_len; })
                                    : qs_bin_search<F64>(q.thr + ft.thr_off, ft.thr_len, x);
    if (vm_stores) newer += (uint32_t)(ft.view_end - ft.view_begin);   // one store per view below (at least one lane of the wavefront has an item)
    uint16_t *d = dst;
    const bool act = active;
    auto emit = [d, act](uint32_t view, uint32_t cell) { if (act) d[view * QS_TILE_ROWS] = (uint16_t)cell; };
    if constexpr (QS::is_static)
      qs_emit_views_sig<F64>(x, pos, QS{}[col], [&]() -> uint32_t { return f->zero_bin; }, emit);
    else
      qs_emit_views<F64>(x, pos, ft, q.views, emit);
  }
};

// ---- the same tile written by a workgroup that holds EVERY threshold table of the forest in LDS (the item-parallel kernel
// of a model whose signature is known at compile time and whose tables fit: assemble_cells_rt_body).  Nothing is staged per
// column, so nothing is waited for, and a value need not be binned the moment it is produced: the sink keeps up to RT_Q
// values back and bins them TOGETHER - RT_Q independent lower-bound searches whose LDS reads are in flight at the same
// time.  The staging sink's search is a chain of 7-8 dependent LDS trips per column, 24 columns one after the other; here
// the chain of a group is as long as ONE search.  Same lower bound, same cells.
// (The queue is three named slots and a count, all of which fold in a compile-time program: every put happens at a fixed
// point of straight-line code.  Were the count not to fold, the switch below would be a uniform branch - never a run-time
// index into registers.)
#ifndef MRK_RT_Q
#define MRK_RT_Q 4
#endif
constexpr int RT_Q = MRK_RT_Q;   // 2 .. 4
static_assert(RT_Q >= 2 && RT_Q <= 4, "the resident-table sink holds 1 - 3 values back");
// The workgroup-per-request kernels keep the compact tables resident too when they are no bigger than this (the benchmark's
// Ranklens model - ~50 distinct thresholds on its continuous columns, 740 in all: 6 KB - where two wavefronts' staging buffers take
// 4 - 8 KB): the same LDS class, so the same residency, no staging traffic (every wavefront staged every column's table for every 64 candidates), no counted waits, 4 searches at a time.
// (the split / sliced kernel's workgroups are up to 8 wavefronts - 16 KB of staging buffers at 128-entry tables: its cap is 20 KB: the 64-column c3 model is 16.4 KB)
#ifndef MRK_FUSED_RT_MAX
#define MRK_FUSED_RT_MAX 8192
#endif
#ifndef MRK_FUSED_RT_MAX_SPLIT
#define MRK_FUSED_RT_MAX_SPLIT 20480
#endif
// (A/B through MRK_JIT_DEFINES: 0 = the staging sink; never above the library's own values - the host sizes the region by those)
constexpr size_t FUSED_RT_MAX_BYTES = MRK_FUSED_RT_MAX, FUSED_RT_MAX_BYTES_SPLIT = MRK_FUSED_RT_MAX_SPLIT;
template <typename QS, bool SPLIT> __device__ __forceinline__ constexpr bool qs_fused_rt() {
  if constexpr (QS::is_static) return QS::rt_total > 0u && (size_t)QS::rt_total * 8 <= (SPLIT ? FUSED_RT_MAX_BYTES_SPLIT : FUSED_RT_MAX_BYTES);
  else return false;
}
// QUEUE = false: every value is binned at once (the kernels whose workgroups share out the ops at RUN time - op split - reach a
// put under a run-time condition: the count of values held back would not fold, and every put would carry every group shape).
template <bool F64, typename QS, bool QUEUE = true>
struct CellSinkRT {
  static_assert(QS::is_static, "the resident-table sink needs the forest's signature at compile time");
  QsDev q;
  uint16_t *dst;           // &cells[tile][0][row]
  int32_t *status;
  qs_lds_double *thr_all;  // the forest's COMPACT threshold tables (QsDev::thr_rt, QsSig::rt_off / rt_len)
  bool active;
  mutable double v0 = 0.0, v1 = 0.0, v2 = 0.0;
  mutable int c0 = 0, c1 = 0, c2 = 0, qn = 0;

  __device__ __forceinline__ void begin() const { qn = 0; }
  __device__ __forceinline__ QsFeatureK desc(int col) const { return (QsFeatureK)(unsigned long long)q.feats + col; }

  template <int N>
  __device__ __forceinline__ void bin_group(const int (&col)[4], const double (&val)[4]) const {
    double x[N];
    qs_lds_double *T[N], *p[N];
    uint32_t pos[N];
    auto below = [](double t, double xx) { return F64 ? (t < xx) : (t <= xx); };
    // entries of column c's compact table in LDS (0: none there)
    auto span = [](int c) -> uint32_t { return QS{}[c].rt_len <= 256u ? QS{}[c].rt_len : 0u; };
    uint32_t n[N];   // what is left of each column's range (compile time: the lengths are constants of the signature)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      bool ok;
      x[i] = qs_prep<F64>(val[i], ok);
      if (!ok && active) atomicOr(status, ST_XGB_INF);
      T[i] = thr_all + QS{}[col[i]].rt_off;
      p[i] = T[i];
      n[i] = span(col[i]);
    }
    // the branch-free lower bound of qs_bin_search - halve the range, keep the half the value lies in - for every column of the
    // group TOGETHER, step by step: a step's N reads are in flight at the same time.  (The scheduler, short of registers, would
    // run the N chains one after the other again: the barriers keep a step's reads together, ahead of the step's compares.)
    double t[N];
#pragma unroll
    for (int s = 0; s < 8; ++s) {   // 256 entries: 8 halvings
      bool any = false;
#pragma unroll
      for (int i = 0; i < N; ++i) any = any || n[i] > 1u;
      if (!any) break;   // (compile time)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < N; ++i) t[i] = n[i] > 1u ? p[i][(n[i] >> 1) - 1u] : 0.0;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (n[i] > 1u) {
          const uint32_t half = n[i] >> 1;
          p[i] += below(t[i], x[i]) ? half : 0u;
          n[i] -= half;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = span(col[i]) > 0u ? p[i][0] : 0.0;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const QsSig sg = QS{}[col[i]];
      if (sg.rt_len == 0u) pos[i] = 0u;
      else if (sg.rt_len <= 256u) pos[i] = (uint32_t)(p[i] - T[i]) + (below(t[i], x[i]) ? 1u : 0u);   // (no padding to walk through: pos <= the table's length)
      else pos[i] = qs_bin_search<F64>(q.thr + sg.thr_off, desc(col[i])->thr_len, x[i]);  // a table too long for LDS: searched where it lies
    }
    uint16_t *d = dst;
    const bool act = active;
    auto emit = [d, act](uint32_t view, uint32_t cell) { if (act) d[view * QS_TILE_ROWS] = (uint16_t)cell; };
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const QsFeatureK f = desc(col[i]);
      qs_emit_views_sig<F64>(x[i], pos[i], QS{}[col[i]], [&]() -> uint32_t { return f->zero_bin; }, emit);
    }
  }

  __device__ __forceinline__ void put(int col, double v) const {
    bool splits = false;
    if (col < QS::n_feats) splits = QS{}[col].view_begin != QS{}[col].view_end;
    if (!splits) {  // a column the forest does not know or never splits on: nothing to bin, but still part of XGBoost's DMatrix row
      if constexpr (!F64) {
        bool fin;
        (void)qs_prep<F64>(v, fin);
        if (!fin && active) atomicOr(status, ST_XGB_INF);
      }
      return;
    }
    if constexpr (!QUEUE) {
      const int cols[4] = {col, 0, 0, 0};
      const double vals[4] = {v, 0.0, 0.0, 0.0};
      bin_group<1>(cols, vals);
      return;
    }
    if (qn == RT_Q - 1) {   // the group is complete
      int cols[4] = {c0, c1, c2, 0};
      double vals[4] = {v0, v1, v2, 0.0};
      cols[RT_Q - 1] = col;
      vals[RT_Q - 1] = v;
      bin_group<RT_Q>(cols, vals);
      qn = 0;
      return;
    }
    switch (qn) {
      case 0: v0 = v; c0 = col; qn = 1; break;
      case 1: v1 = v; c1 = col; qn = 2; break;
      default: v2 = v; c2 = col; qn = 3; break;
    }
  }
  __device__ __forceinline__ void finish() const {
    const int cols[4] = {c0, c1, c2, 0};
    const double vals[4] = {v0, v1, v2, 0.0};
    if (qn == 1) bin_group<1>(cols, vals);
    else if (qn == 2) bin_group<2>(cols, vals);
    else if (qn == 3) bin_group<3>(cols, vals);
    qn = 0;
  }
};

// a program whose ops / prep / aux are compile-time constants (the run-time specialised kernel, jit.cpp) says so
template <typename P> __device__ __forceinline__ constexpr auto prog_is_static(int) -> decltype(P::is_static) { return P::is_static; }
template <typename P> __device__ __forceinline__ constexpr bool prog_is_static(...) { return false; }

// f(IntC<I>{}) for I in [I0, I1): a loop whose index is a constant expression in the body
template <int I> struct IntC { static constexpr int value = I; };
template <int I0, int I1, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I0 < I1) {
    f(IntC<I0>{});
    static_for<I0 + 1, I1>(f);
  }
}

// ---- the candidate's own record, seen two ways.
// PtrRec reads it from memory cell by cell (the kernels of rank.hip, any program).  RegRec holds its fixed part - tag
// bytes and value cells, NP 16-byte pieces - in registers: the kernel specialised for one model (jit.cpp) fetches every
// line of the record exactly once, with all loads in flight together, and every cell an op touches later is a register
// read with a compile-time offset.  (Measured on 4 M candidates over an 8 M-item table, profiles/r02_b: cell-by-cell
// loads cost 23 L2 misses per candidate - 1.4 KB for a 384-byte record - because the lines are evicted between ops.)
struct PtrRec {
  static constexpr bool in_regs = false;
  const uint8_t *p;
  __device__ __forceinline__ uint32_t tag(int tag_index) const { return p[tag_index]; }
  __device__ __forceinline__ uint64_t u64(int byte_off) const { return *(const uint64_t *)(p + byte_off); }
};

template <int NP>
struct RegRec {
  static constexpr bool in_regs = true;
  const uint8_t *p;   // still needed: inline string lists are read from the record's heap
  uint4 r[NP];
  __device__ __forceinline__ uint32_t tag(int tag_index) const {
    const uint4 v = r[tag_index >> 4];
    const int w = (tag_index >> 2) & 3;
    const uint32_t word = w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w;
    return (word >> (8 * (tag_index & 3))) & 0xffu;
  }
  __device__ __forceinline__ uint64_t u64(int byte_off) const {
    const uint4 v = r[byte_off >> 4];
    return (byte_off & 8) ? ((uint64_t)v.w << 32) | v.z : ((uint64_t)v.y << 32) | v.x;
  }
};

template <int NP>
__device__ __forceinline__ RegRec<NP> load_record_regs(const uint8_t *p) {
  RegRec<NP> R;
  R.p = p;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    R.r[i] = make_uint4(0u, 0u, 0u, 0u);          // no record: every tag reads TAG_MISSING
    if (p != nullptr) R.r[i] = ((const uint4 *)p)[i];
  }
  return R;
}

template <typename IR>
__device__ __forceinline__ Cell rec_cell(const IR &ir, ColRef c, int idx = 0) {
  Cell out;
  out.tag = TAG_MISSING;
  out.bits = 0;
  if (ir.p == nullptr || c.tag < 0) return out;
  out.tag = ir.tag(c.tag);
  out.bits = ir.u64(c.val + idx * 8);
  return out;
}

// ops whose first load is the cell of their primary column op.c0
__device__ __forceinline__ constexpr bool op_has_primary(const Op &op) {
  const int kind = op.kind;
  if (kind == OP_RATE) return op.i0 == RATE_ITEM_FIELD;  // the link cell
  return kind == OP_SCALAR_DOUBLE || kind == OP_SCALAR_BOOL || kind == OP_VECTOR || kind == OP_STRING_INDEX || kind == OP_STRING_ONEHOT ||
         kind == OP_COUNTER || kind == OP_WINDOW || kind == OP_DIVERSITY || kind == OP_ITEM_AGE || kind == OP_BIENCODER;
}
// ... and whether that column lives in the candidate's own record
__device__ __forceinline__ constexpr bool op_primary_in_item(const Op &op) {
  const int kind = op.kind;
  if (kind == OP_DIVERSITY || kind == OP_BIENCODER || kind == OP_RATE) return true;
  return op.scope == SC_ITEM;
}

#ifndef MRK_IW_TOK
#define MRK_IW_TOK 5   // (round 6: 4 -> 5, the benchmark's tag lists have 5 tokens - the sixth-token path is a trip to memory inside the op; c2 assembly -1.5 %, c4x -2 %, r06_p)
#endif
constexpr int IW_BATCH = 4, IW_TOK = MRK_IW_TOK;   // interacted_with: fields handled together, tokens per field fetched ahead
constexpr int RATE_BATCH = 4;             // rate: periods fetched together
constexpr int PRE_TOK = IW_BATCH * IW_TOK;
constexpr int PRE_F64 = 8;

// What an op needs from memory BEYOND the candidate's record, fetched ahead of the arithmetic (second trip): the first
// tokens of its string lists, the counters of the record a scoped rate points to, the global counters of a normalised
// rate, the head of a stored vector.  In the specialised kernel the fetches of ALL ops are issued together, before the
// first op computes anything; the interpreting kernels fetch op by op.  Fields an op kind does not use are never
// assigned and cost nothing.
struct OpPre {
  Cell fc[IW_BATCH];                    // interacted_with: the field cells
  uint32_t tok[PRE_TOK];                // interacted_with: token t of field u at [u * IW_TOK + t]; string / diversity: the list's first tokens
  long long tv[RATE_BATCH], bv[RATE_BATCH], gtv[RATE_BATCH], gbv[RATE_BATCH];
  uint32_t ttag, btag, gttag, gbtag;
  double d[PRE_F64];                    // vector: the first stored values
};

// Registers (32-bit) the fetched-ahead state of one op occupies, roughly: what decides how many ops share a group
__device__ __forceinline__ constexpr int op_pre_weight(const Op &op) {
  switch (op.kind) {
    case OP_INTERACTED: return 8 * (op.dim < IW_BATCH ? op.dim : IW_BATCH) + IW_TOK * (op.dim < IW_BATCH ? op.dim : IW_BATCH);
    case OP_DIVERSITY: return TOK_BATCH;
    case OP_RATE: return 4 * (op.dim < RATE_BATCH ? op.dim : RATE_BATCH) * (op.i3 != 0 ? 2 : 1) + 4;
    case OP_VECTOR: return 2 * (op.dim < PRE_F64 ? op.dim : PRE_F64);
    case OP_STRING_INDEX: return 1;
    default: return 0;
  }
}
#ifndef MRK_PRE_GROUP_BUDGET
#define MRK_PRE_GROUP_BUDGET 48   // (round 6: 72 -> 48 - c2's 18 ops in 3 groups instead of 2: neutral on c2, the item-parallel kernel -5.7 % (fewer spills), 96: +10 %; r06_p)
#endif
constexpr int PRE_GROUP_BUDGET = MRK_PRE_GROUP_BUDGET;  // registers of fetched-ahead state per group (c2's 19 ops: 2 groups)
// end of the group of ops that starts at `lo` (in_regs: the candidate's record is held in registers, so a primary cell
// that lives in it costs nothing more)
template <typename Prog, bool IN_REGS>
__device__ __forceinline__ constexpr int op_group_end(int lo) {
  int w = 0, hi = lo;
  while (hi < Prog::n_ops) {
    const Op op = Prog{}.ops[hi];
    const int wi = op_pre_weight(op) + (op_has_primary(op) && !(IN_REGS && op_primary_in_item(op)) ? 3 : 0);
    if (hi > lo && w + wi > PRE_GROUP_BUDGET) break;
    w += wi;
    ++hi;
  }
  return hi;
}
// A small batch (a single request) has too few workgroups to fill a CU, let alone the chip: its workgroups then carry
// OP_SPLIT_MAX copies of the item lanes and every copy evaluates a quarter of the program ("op split": the assembly phase
// of a request shortens by the split factor; the pre-pass runs once, on all lanes).  Which copy owns an op: greedy
// balancing of a rough cost, fixed per program.
constexpr int OP_SPLIT_MAX = 4;
__device__ __forceinline__ constexpr int op_cost(const Op &op) {
  switch (op.kind) {
    case OP_INTERACTED: return 5 * op.dim;
    case OP_DIVERSITY: return 7;
    case OP_RATE: return (op.i0 == RATE_ITEM ? 3 : 6) * op.dim;
    case OP_BIENCODER: return 40;
    default: return op.dim > 0 ? op.dim : 1;
  }
}
template <typename Prog>
__device__ __forceinline__ constexpr int op_owner(int oi) {
  int load[OP_SPLIT_MAX] = {0, 0, 0, 0};
  int owner = 0;
  for (int k = 0; k <= oi && k < Prog::n_ops; ++k) {
    owner = 0;
    for (int g = 1; g < OP_SPLIT_MAX; ++g)
      if (load[g] < load[owner]) owner = g;
    load[owner] += op_cost(Prog{}.ops[k]);
  }
  return owner;
}

template <typename Prog, bool IN_REGS, int LO, typename F>
__device__ __forceinline__ void run_groups(F &&f) {
  if constexpr (LO < Prog::n_ops) {
    constexpr int HI = op_group_end<Prog, IN_REGS>(LO);
    f(IntC<LO>{}, IntC<HI>{});
    run_groups<Prog, IN_REGS, HI>(f);
  }
}

// Evaluates the model program for batch item gi of request r.  Hash tables: tab_base + (po.tab_off - tab_sub).
// (og, G): op split - this wavefront evaluates the ops whose owner & (G - 1) == og; G = 1: all of them
template <bool SPLIT, typename Prog, typename Sink, typename IR>
__device__ __forceinline__ void assemble_item_rec(const StoreDev &st, const Prog &prog, const BatchDev &b, int gi, int r,
                                                  const ReqDev &rq, const unsigned long long *tab_base, uint32_t tab_sub,
                                                  const PrepOut *pos, const Sink &sink, int islot, const IR &ir, int og, int G_arg) {
  const int G = SPLIT ? G_arg : 1;  // the hot kernels are instantiated without the split: every guard below folds away
  const uint8_t *irec = ir.p;
  const double NaN = d_nan();

  // the record an op's primary column (op.c0) lives in, when that is not the candidate's own
  auto other_record = [&](const Op &op) -> const uint8_t * {
    if (op.kind == OP_ITEM_AGE) return nullptr;
    return record(st, op.scope, scoped_slot(rq, op.scope, islot));
  };
  auto primary_cell = [&](const Op &op) -> Cell {
    Cell pc;
    pc.tag = TAG_MISSING;
    pc.bits = 0;
    if (!op_has_primary(op)) return pc;
    return op_primary_in_item(op) ? rec_cell(ir, op.c0) : load_cell(other_record(op), op.c0);
  };
  // the record the tokens of the op's primary string list live in
  auto primary_list_record = [&](const Op &op) -> const uint8_t * { return op_primary_in_item(op) ? irec : other_record(op); };

  // ---- second trip to memory of one op (see OpPre)
  auto prefetch_op = [&](const Op &op, const Cell &pc, OpPre &pre) __attribute__((always_inline)) {
    switch (op.kind) {
      case OP_VECTOR: {
        const bool has = pc.tag == TAG_DOUBLE_LIST;
#pragma unroll
        for (int k = 0; k < PRE_F64; ++k) {
          pre.d[k] = 0.0;
          if (k < op.dim && has && (uint32_t)k < pc.hi()) pre.d[k] = list_f64(st, pc.lo(), (uint32_t)k);
        }
        break;
      }
      case OP_STRING_INDEX: {
        pre.tok[0] = 0u;
        if (pc.tag == TAG_STRING_LIST && pc.hi() > 0) pre.tok[0] = list_tokens(st, primary_list_record(op), pc.lo())[0];
        break;
      }
      case OP_DIVERSITY: {
        const uint32_t len = pc.tag == TAG_STRING_LIST ? pc.hi() : 0u;
        const uint32_t *toks = list_tokens(st, irec, pc.lo());
#pragma unroll
        for (int t = 0; t < TOK_BATCH; ++t) pre.tok[t] = (uint32_t)t < len ? toks[t] : 0u;
        break;
      }
      case OP_INTERACTED: {
#pragma unroll
        for (int u = 0; u < IW_BATCH; ++u) {
          pre.fc[u].tag = TAG_MISSING;
          pre.fc[u].bits = 0;
          if (u < op.dim) {
            ColRef col;
            col.tag = (int32_t)prog.aux[op.i0 + 2 * u];
            col.val = (int32_t)prog.aux[op.i0 + 2 * u + 1];
            pre.fc[u] = rec_cell(ir, col);
          }
        }
#pragma unroll
        for (int u = 0; u < IW_BATCH; ++u) {
          const uint32_t len = pre.fc[u].tag == TAG_STRING_LIST ? pre.fc[u].hi() : 0u;
          const uint32_t *toks = list_tokens(st, irec, pre.fc[u].lo());
#pragma unroll
          for (int t = 0; t < IW_TOK; ++t) pre.tok[u * IW_TOK + t] = (uint32_t)t < len ? toks[t] : 0u;
        }
        break;
      }
      case OP_RATE: {
        // every value present with exactly `dim` periods, else NaN x dim (RateFeature.scala:318-350).  All counters of
        // the op are requested together; the value cells of a column exist whatever its tag says.
        ColRef top = op.c0, bot = op.c1;
        const uint8_t *trec = nullptr;  // the record holding the target-scope counters, when it is not the candidate's
        if (op.i0 == RATE_ITEM_FIELD) {
          if (pc.tag == TAG_STRING && pc.hi() != 0) trec = record(st, SC_FIELD, (int)pc.hi() - 1);  // item=<id>/<name>_field -> field slot
          top = op.c4;
          bot = op.c5;
        } else if (op.i0 == RATE_RANKING_FIELD) {
          trec = record(st, SC_IRF, b.irf[(size_t)op.i2 * b.total_items + gi]);
          top = op.c4;
          bot = op.c5;
        }
        const bool cols = top.tag >= 0 && bot.tag >= 0;
        pre.ttag = pre.btag = pre.gttag = pre.gbtag = TAG_MISSING;
#pragma unroll
        for (int u = 0; u < RATE_BATCH; ++u) pre.tv[u] = pre.bv[u] = pre.gtv[u] = pre.gbv[u] = 0;
        if (op.i0 == RATE_ITEM) {
          if (irec != nullptr && cols) {
            pre.ttag = ir.tag(top.tag);
            pre.btag = ir.tag(bot.tag);
#pragma unroll
            for (int u = 0; u < RATE_BATCH; ++u)
              if (u < op.dim) { pre.tv[u] = (long long)ir.u64(top.val + u * 8); pre.bv[u] = (long long)ir.u64(bot.val + u * 8); }
          }
        } else if (trec != nullptr && cols) {
          pre.ttag = trec[top.tag];
          pre.btag = trec[bot.tag];
#pragma unroll
          for (int u = 0; u < RATE_BATCH; ++u)
            if (u < op.dim) { pre.tv[u] = *(const long long *)(trec + top.val + u * 8); pre.bv[u] = *(const long long *)(trec + bot.val + u * 8); }
        }
        if (op.i3 != 0 && op.c2.tag >= 0 && op.c3.tag >= 0) {
          const uint8_t *grec = record(st, SC_GLOBAL, 0);
          if (grec != nullptr) {
            pre.gttag = grec[op.c2.tag];
            pre.gbtag = grec[op.c3.tag];
#pragma unroll
            for (int u = 0; u < RATE_BATCH; ++u)
              if (u < op.dim) { pre.gtv[u] = *(const long long *)(grec + op.c2.val + u * 8); pre.gbv[u] = *(const long long *)(grec + op.c3.val + u * 8); }
          }
        }
        break;
      }
      default: break;
    }
  };

  // ---- one op, given its primary cell `pc` (missing for the kinds without one) and what prefetch_op fetched
  auto run_op = [&](const Op &op, const Cell &pc, const OpPre &pre) __attribute__((always_inline)) {
    const int dst = op.dst;
    switch (op.kind) {
      case OP_SCALAR_DOUBLE: {
        const Cell c = pc;
        sink.put(dst + 0, c.tag == TAG_DOUBLE ? c.f64() : NaN);
        break;
      }
      case OP_SCALAR_BOOL: {
        const Cell c = pc;
        sink.put(dst + 0, c.tag == TAG_BOOL ? c.f64() : NaN);
        break;
      }
      case OP_VECTOR: {
        const Cell c = pc;
        const bool has = c.tag == TAG_DOUBLE_LIST;
        const uint32_t off = c.lo(), len = c.hi();
#pragma unroll
        for (int k = 0; k < PRE_F64; ++k) {  // every put at a wavefront-uniform point (CellSink stages tables cooperatively)
          if (k >= op.dim) break;
          sink.put(dst + k, has ? pre.d[k] : NaN);
        }
        for (int k = PRE_F64; k < op.dim; ++k) {
          double v = NaN;
          if (has) v = (uint32_t)k < len ? list_f64(st, off, (uint32_t)k) : 0.0;
          sink.put(dst + k, v);
        }
        break;
      }
      case OP_STRING_INDEX: {
        const Cell c = pc;
        double idx = 0.0;
        if (c.tag == TAG_STRING_LIST && c.hi() > 0) {
          const uint32_t first = pre.tok[0];
          for (int k = 0; k < op.i1; ++k)
            if (prog.aux[op.i0 + k] == first) idx = (double)(k + 1);  // zipWithIndex.toMap: last duplicate wins
        }
        sink.put(dst + 0, idx);
        break;
      }
      case OP_STRING_ONEHOT: {
        // OneHotEncoder.fromValues: every token sets the FIRST position whose value equals it (indexOf)
        const Cell c = pc;
        const bool has = c.tag == TAG_STRING_LIST;
        const uint32_t len = has ? c.hi() : 0u;
        const uint32_t *toks = list_tokens(st, primary_list_record(op), c.lo());
        for (int k = 0; k < op.dim; ++k) {
          double v = 0.0;
          if (k < op.i1) {
            const uint32_t want = prog.aux[op.i0 + k];
            bool first = true;  // a duplicate possible value is never reached by indexOf
            for (int k2 = 0; k2 < k; ++k2) first = first && prog.aux[op.i0 + k2] != want;
            if (first)
              for (uint32_t j = 0; j < len; ++j)
                if (toks[j] == want) { v = 1.0; break; }
          }
          sink.put(dst + k, v);
        }
        break;
      }
      case OP_COUNTER: {
        const Cell c = pc;
        sink.put(dst + 0, c.tag != TAG_MISSING ? (double)c.i64() : 0.0);
        break;
      }
      case OP_WINDOW: {
        const Cell c = pc;
        const bool ok = c.tag != TAG_MISSING && (int)c.tag - 1 == op.dim;
        const bool in_item = op_primary_in_item(op);
        const uint8_t *rec = in_item ? nullptr : other_record(op);
#pragma unroll 8
        for (int k = 0; k < op.dim; ++k) {
          const Cell ck = in_item ? rec_cell(ir, op.c0, k) : load_cell(rec, op.c0, k);
          sink.put(dst + k, ok ? (double)ck.i64() : NaN);
        }
        break;
      }
      case OP_RATE: {
        const bool norm = op.i3 != 0;
        const uint32_t ttag = pre.ttag, btag = pre.btag, gttag = pre.gttag, gbtag = pre.gbtag;
        bool valid = ttag != TAG_MISSING && btag != TAG_MISSING && (int)ttag - 1 == op.dim && (int)btag - 1 == op.dim;
        if (norm) valid = valid && gttag != TAG_MISSING && gbtag != TAG_MISSING && (int)gttag - 1 == op.dim && (int)gbtag - 1 == op.dim;
        bool thrown = false;  // java.lang.ArithmeticException: / by zero aborts the request
        // periods beyond the first RATE_BATCH: fetched here, batch by batch (the record pointers are re-derived)
        ColRef top = op.c0, bot = op.c1;
        const uint8_t *trec = nullptr;
        if (op.i0 == RATE_ITEM_FIELD) {
          if (pc.tag == TAG_STRING && pc.hi() != 0) trec = record(st, SC_FIELD, (int)pc.hi() - 1);
          top = op.c4;
          bot = op.c5;
        } else if (op.i0 == RATE_RANKING_FIELD) {
          if (op.dim > RATE_BATCH) trec = record(st, SC_IRF, b.irf[(size_t)op.i2 * b.total_items + gi]);
          top = op.c4;
          bot = op.c5;
        }
        const uint8_t *grec = norm && op.dim > RATE_BATCH ? record(st, SC_GLOBAL, 0) : nullptr;
        for (int k0 = 0; k0 < op.dim; k0 += RATE_BATCH) {
          long long tv[RATE_BATCH], bv[RATE_BATCH], gtv[RATE_BATCH], gbv[RATE_BATCH];
#pragma unroll
          for (int u = 0; u < RATE_BATCH; ++u) {
            tv[u] = pre.tv[u]; bv[u] = pre.bv[u]; gtv[u] = pre.gtv[u]; gbv[u] = pre.gbv[u];
            if (k0 > 0) {
              tv[u] = bv[u] = gtv[u] = gbv[u] = 0;
              if (k0 + u < op.dim && ttag != TAG_MISSING) {
                if (op.i0 == RATE_ITEM) {
                  tv[u] = (long long)ir.u64(top.val + (k0 + u) * 8);
                  bv[u] = (long long)ir.u64(bot.val + (k0 + u) * 8);
                } else if (trec != nullptr) {
                  tv[u] = *(const long long *)(trec + top.val + (k0 + u) * 8);
                  bv[u] = *(const long long *)(trec + bot.val + (k0 + u) * 8);
                }
              }
              if (k0 + u < op.dim && grec != nullptr && gttag != TAG_MISSING) {
                gtv[u] = *(const long long *)(grec + op.c2.val + (k0 + u) * 8);
                gbv[u] = *(const long long *)(grec + op.c3.val + (k0 + u) * 8);
              }
            }
          }
#pragma unroll
          for (int u = 0; u < RATE_BATCH; ++u) {
            if (k0 + u >= op.dim) break;
            double v = NaN;
            if (valid && !thrown) {
              if (!norm) {
                v = (double)tv[u] / (double)bv[u];
              } else if (gtv[u] == 0) {
                if (sink.active) atomicOr(&b.status[r], ST_ARITHMETIC);
                thrown = true;
              } else {
                const double ratio = long_div_to_double(gbv[u], gtv[u]);
                const double num = __dadd_rn(op.d0, (double)tv[u]);
                const double den = __dadd_rn(__dmul_rn(op.d0, ratio), (double)bv[u]);
                v = num / den;
              }
            }
            sink.put(dst + k0 + u, v);
          }
        }
        break;
      }
      case OP_INTERACTED: {
        // per field: sum over the candidate's tokens of the session histogram.  IW_BATCH fields at a time: their
        // cells are requested together, then the first tokens of all their lists, then the tables are probed.
        for (int f0 = 0; f0 < op.dim; f0 += IW_BATCH) {
          Cell fc[IW_BATCH];
          uint32_t tk[IW_BATCH][IW_TOK];
#pragma unroll
          for (int u = 0; u < IW_BATCH; ++u) {
            fc[u] = pre.fc[u];
#pragma unroll
            for (int t = 0; t < IW_TOK; ++t) tk[u][t] = pre.tok[u * IW_TOK + t];
          }
          if (f0 > 0) {  // later groups of fields: fetched here
#pragma unroll
            for (int u = 0; u < IW_BATCH; ++u) {
              fc[u].tag = TAG_MISSING;
              fc[u].bits = 0;
              if (f0 + u < op.dim) {
                ColRef col;
                col.tag = (int32_t)prog.aux[op.i0 + 2 * (f0 + u)];
                col.val = (int32_t)prog.aux[op.i0 + 2 * (f0 + u) + 1];
                fc[u] = rec_cell(ir, col);
              }
            }
#pragma unroll
            for (int u = 0; u < IW_BATCH; ++u) {
              const uint32_t len = fc[u].tag == TAG_STRING_LIST ? fc[u].hi() : 0u;
              const uint32_t *toks = list_tokens(st, irec, fc[u].lo());
#pragma unroll
              for (int t = 0; t < IW_TOK; ++t) tk[u][t] = (uint32_t)t < len ? toks[t] : 0u;
            }
          }
#pragma unroll
          for (int u = 0; u < IW_BATCH; ++u) {
            if (f0 + u >= op.dim) break;
            const PrepOut po = pos[op.i1 + f0 + u];
            const unsigned long long *tab = tab_base + (po.tab_off - tab_sub);
            const uint32_t len = fc[u].tag == TAG_STRING_LIST ? fc[u].hi() : 0u;
            double cnt = table_sum_tokens<IW_TOK>(tab, po.tab_cap, tk[u], len, 0.0);
            if (wave_any(len > (uint32_t)IW_TOK))  // the rest of longer lists, in list order
              cnt = table_sum_list(list_tokens(st, irec, fc[u].lo()) + IW_TOK, tab, po.tab_cap, len > (uint32_t)IW_TOK ? len - IW_TOK : 0u, cnt);
            sink.put(dst + f0 + u, cnt);
          }
        }
        break;
      }
      case OP_DIVERSITY: {
        const PrepOut po = pos[op.i1];
        const Cell c = pc;
        double v = NaN;
        if (po.mode == DIV_EMPTY) {
          v = 0.0;
        } else if (po.mode == DIV_DOUBLE) {
          if (c.tag == TAG_DOUBLE) v = c.f64() - po.scalar;
        } else {
          const unsigned long long *tab = tab_base + (po.tab_off - tab_sub);
          const bool one = c.tag == TAG_STRING, list = c.tag == TAG_STRING_LIST;
          const double w1 = 0.0 + (double)table_get(tab, po.tab_cap, c.lo(), one);
          // the list's first TOK_BATCH tokens were fetched ahead; longer lists continue from memory, in list order
          const uint32_t len = list ? c.hi() : 0u;
          uint32_t dtk[TOK_BATCH];
#pragma unroll
          for (int t = 0; t < TOK_BATCH; ++t) dtk[t] = pre.tok[t];
          double wl = table_sum_tokens<TOK_BATCH>(tab, po.tab_cap, dtk, len, 0.0);
          if (wave_any(len > (uint32_t)TOK_BATCH))
            wl = table_sum_list(list_tokens(st, irec, c.lo()) + TOK_BATCH, tab, po.tab_cap, len > (uint32_t)TOK_BATCH ? len - TOK_BATCH : 0u, wl);
          if (one || list) v = (one ? w1 : wl) / po.scalar;
        }
        sink.put(dst + 0, v);
        break;
      }
      case OP_ITEM_AGE: {
        const Cell c = pc;
        double v = NaN;
        if (c.tag == TAG_DOUBLE) {
          const long long updated = java_round(c.f64() * 1000.0);
          long long diff = (long long)((unsigned long long)rq.ts_ms - (unsigned long long)updated);
          if (diff < 0) diff = (long long)(0ull - (unsigned long long)diff);
          if (diff < 0 || diff > 9223372036854LL) { if (sink.active) atomicOr(&b.status[r], ST_ILLEGAL_ARG); }  // FiniteDuration bound
          else v = (double)(diff / 1000);
        }
        sink.put(dst + 0, v);
        break;
      }
      case OP_CONST: {
        const double *cs = b.consts + (size_t)r * prog.n_consts + op.i0;
        for (int k = 0; k < op.dim; ++k) sink.put(dst + k, cs[k]);
        break;
      }
      case OP_FILL_NAN: {
        for (int k = 0; k < op.dim; ++k) sink.put(dst + k, NaN);
        break;
      }
      case OP_BIENCODER: {
        // consts: [0] = query length (or -1: no query), [1..] = query embedding (f32 values widened)
        const double *cs = b.consts + (size_t)r * prog.n_consts + op.i0;
        const int qn = (int)cs[0];
        const Cell c = pc;
        double v = NaN;
        if (qn >= 0 && c.tag == TAG_DOUBLE_LIST) {
          if ((int)c.hi() < qn) {
            if (sink.active) atomicOr(&b.status[r], ST_DIM);
          } else {
            // the products and sums of DistanceFunction.scala:14-26, in element order: f32 query x f64 item, f64 accumulators
            double top = 0.0, a = 0.0, bs = 0.0;
            auto step = [&](int k, double x) {
              const float q = (float)cs[1 + k];
              top = __dadd_rn(top, __dmul_rn((double)q, x));
              a = __dadd_rn(a, (double)__fmul_rn(q, q));  // Float * Float is a Float product
              bs = __dadd_rn(bs, __dmul_rn(x, x));
            };
            if (c.lo() & LIST_F32_BIT) {
              // stored as f32 (exactly the values the reference holds as doubles): 16 bytes = 4 elements per load, half the
              // bytes of the f64 form (the pool keeps ranges 16-byte aligned)
              const float *item = st.f32_pool + (c.lo() & ~LIST_F32_BIT);
              int k = 0;
              for (; k + 4 <= qn; k += 4) {
                const float4 x4 = *(const float4 *)(item + k);
                step(k, (double)x4.x);
                step(k + 1, (double)x4.y);
                step(k + 2, (double)x4.z);
                step(k + 3, (double)x4.w);
              }
              for (; k < qn; ++k) step(k, (double)item[k]);
            } else {
              const double *item = st.f64_pool + c.lo();
              for (int k = 0; k < qn; ++k) step(k, item[k]);
            }
            v = top / (sqrt(a) * sqrt(bs));
          }
        }
        sink.put(dst + 0, v);
        break;
      }
      default: break;
    }
  };

  sink.begin();
  if constexpr (prog_is_static<Prog>(0)) {
    // compile-time program: the loops unroll, every `op` is a constant.  Trip 1 (the record) has been issued by the
    // caller; trip 2 - what the ops need beyond it - is issued here for a whole group of ops before the first of them
    // computes anything.
    Cell pc[Prog::n_ops > 0 ? Prog::n_ops : 1];
    OpPre pre[Prog::n_ops > 0 ? Prog::n_ops : 1];
#ifdef MRK_PHASE_CLOCKS
    unsigned long long t_op = clock64(), op_acc[Prog::n_ops > 0 ? Prog::n_ops : 1] = {}, pre_acc = 0;
#endif
    // Ops run in groups whose fetched-ahead state fits the register file: a group's primary cells (register reads when the
    // record is held in registers, else its first trip to memory), then its second trip, then its arithmetic.
    constexpr bool in_regs = IR::in_regs;
    run_groups<Prog, in_regs, 0>([&](auto lo_c, auto hi_c) __attribute__((always_inline)) {
      constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
      static_for<lo, hi>([&](auto ic) __attribute__((always_inline)) {
        constexpr int oi = decltype(ic)::value;
        constexpr Op op = Prog{}.ops[oi];
        constexpr int own = op_owner<Prog>(oi);
        if (G == 1 || (own & (G - 1)) == og) pc[oi] = primary_cell(op);
      });
      static_for<lo, hi>([&](auto ic) __attribute__((always_inline)) {
        constexpr int oi = decltype(ic)::value;
        constexpr Op op = Prog{}.ops[oi];
        constexpr int own = op_owner<Prog>(oi);
        if (G == 1 || (own & (G - 1)) == og) prefetch_op(op, pc[oi], pre[oi]);
      });
      MRK_PHASE(t_op, pre_acc);
      static_for<lo, hi>([&](auto ic) __attribute__((always_inline)) {
        constexpr int oi = decltype(ic)::value;
        constexpr Op op = Prog{}.ops[oi];
        constexpr int own = op_owner<Prog>(oi);
        if (G == 1 || (own & (G - 1)) == og) run_op(op, pc[oi], pre[oi]);
        MRK_PHASE(t_op, op_acc[oi]);
      });
    });
#ifdef MRK_PHASE_CLOCKS
    if (threadIdx.x == 0)
      for (int i = 0; i < Prog::n_ops && i < 48; ++i) atomicAdd(&mrk_phase_clocks[16 + i], op_acc[i]);
    if (threadIdx.x == 0) atomicAdd(&mrk_phase_clocks[7], pre_acc);   // issuing the groups' primary cells + second trips
#endif
  } else {
    for (int oi = 0; oi < prog.n_ops; ++oi) {
      if (G > 1 && (oi & (G - 1)) != og) continue;  // interpreting kernels: ops dealt round-robin
      const Op op = prog.ops[oi];
      const Cell pc = primary_cell(op);
      OpPre pre;
      prefetch_op(op, pc, pre);
      run_op(op, pc, pre);
    }
  }
  sink.finish();
}

// the record pieces the specialised kernel keeps in registers: the whole fixed part of the candidate's record when the
// program says how long it is and it is not too long (24 pieces = 96 registers); else cell-by-cell loads
template <typename P> __device__ __forceinline__ constexpr auto prog_item_pieces(int) -> decltype(P::item_fixed) { return (P::item_fixed + 15) / 16; }
template <typename P> __device__ __forceinline__ constexpr int prog_item_pieces(...) { return 0; }
constexpr int REC_MAX_PIECES = 24;

template <bool SPLIT = false, typename Prog, typename Sink>
__device__ __forceinline__ void assemble_item(const StoreDev &st, const Prog &prog, const BatchDev &b, int gi, int r,
                                              const ReqDev &rq, const unsigned long long *tab_base, uint32_t tab_sub,
                                              const PrepOut *pos, const Sink &sink, int og = 0, int G = 1) {
  const int islot = sink.active ? b.item_slot[gi] : -1;  // lanes without an item: a missing record
  const uint8_t *irec = record(st, SC_ITEM, islot);
  constexpr int NP = prog_item_pieces<Prog>(0);
  if constexpr (NP > 0 && NP <= REC_MAX_PIECES) {
    const RegRec<NP> ir = load_record_regs<NP>(irec);
    assemble_item_rec<SPLIT>(st, prog, b, gi, r, rq, tab_base, tab_sub, pos, sink, islot, ir, og, G);
  } else {
    const PtrRec ir{irec};
    assemble_item_rec<SPLIT>(st, prog, b, gi, r, rq, tab_base, tab_sub, pos, sink, islot, ir, og, G);
  }
}


// Both phases of one request in one workgroup; hash tables, pre-pass results and the median scratch in LDS.
// Dynamic LDS: [tables: tab_entries x 8 B][median values: vals_cap x 8 B][PrepOut x FUSED_MAX_PREP][PREP_INTS ints]
//              [threshold staging: 2 buffers x thr_cap x 8 B per wavefront]
// `mode` = op_split | slices << 8.
//   op_split (1 | 2 | 4; SPLIT kernels only): the workgroup's lanes are op_split copies of the item lanes (see op_owner).
//   slices (>= 1; SPLIT kernels only): a request is covered by `slices` workgroups; each runs the request's pre-pass (its tables are private,
//     in its own LDS) and assembles one slice of the candidates.  A batch of few large requests - 384 requests x 1 000
//     candidates is 1.5 workgroups per CU, each looping four times over its lanes - fills the chip this way: the
//     pre-pass is paid `slices` times, the dependent chain of a workgroup shrinks by the same factor.
//     (Round 4 measured the alternative - the whole batch's pre-pass once per request in a launch of its own, slices copying the
//     finished tables from the arena: c3 assembly 0.306 -> 0.278 ms but + 0.045 ms of pre-pass launch, 869 -> 814 M items/s same
//     box, profiles/r04_d_ab.txt.  Not kept.)
// lds_skip: bytes at the start of the dynamic LDS that belong to the caller (the one-launch kernel keeps the scorer's slab there).
// rt_src / rt_doubles: compact threshold tables the workgroup copies into the threshold region before its pre-pass (0: none - the
// region is per-wavefront staging buffers); make_sink gets both views of the region.
template <bool SPLIT, typename Prog, typename SinkMaker>
__device__ __forceinline__ void rank_fused_body(const StoreDev &st, const Prog &prog, const BatchDev &b, uint32_t tab_entries,
                                                int vals_cap, uint32_t thr_cap, int mode, const SinkMaker &make_sink, uint32_t lds_skip = 0,
                                                const double *rt_src = nullptr, uint32_t rt_doubles = 0, int blk = -1) {
  const int op_split = SPLIT ? ((mode & 255) > 1 ? (mode & 255) : 1) : 1;
  const int slices = SPLIT ? (((mode >> 8) & 255) > 1 ? ((mode >> 8) & 255) : 1) : 1;   // (the plain kernel keeps its registers for the ops)
  extern __shared__ __align__(16) uint8_t smem_base[];
  uint8_t *smem = smem_base + lds_skip;
  unsigned long long *s_tab = (unsigned long long *)smem;
  double *s_vals = (double *)(smem + (size_t)tab_entries * 8);
  PrepOut *s_po = (PrepOut *)(smem + (size_t)tab_entries * 8 + (size_t)vals_cap * 8);
  int *s_int = (int *)(s_po + FUSED_MAX_PREP);
  const size_t thr_at = ((size_t)((uint8_t *)(s_int + PREP_INTS) - smem) + 15) & ~(size_t)15;  // LDS-DMA writes 16 B per lane
  qs_lds_double *s_thr_all = (qs_lds_double *)(smem + thr_at);
  qs_lds_double *s_thr = s_thr_all + (size_t)(threadIdx.x >> 6) * 2 * thr_cap;
  const int bid = blk < 0 ? (int)blockIdx.x : blk;   // (a serving gang's workgroup i ranks request 0 of ITS slot's batch)
  const int r = bid / slices, sl = bid % slices;
  const ReqDev rq = b.reqs[r];
  if (rq.item_begin >= b.item_hi || rq.item_begin + rq.n_items <= b.item_lo) return;  // not in this shard
  const int item_lanes = (int)blockDim.x / op_split;       // a whole number of wavefronts (the host sizes the workgroup)
  // this workgroup's slice of the candidates: whole rounds of the item lanes
  const int per = ((rq.n_items + slices - 1) / slices + item_lanes - 1) / item_lanes * item_lanes;
  const int slice_lo = sl * per, slice_hi = min(rq.n_items, slice_lo + per);
  if (sl > 0 && slice_lo >= rq.n_items) return;            // a shorter request than the batch's longest: nothing left for this slice
  for (int e = threadIdx.x; e < prog.n_prep; e += blockDim.x) s_po[e] = b.prep_out[(size_t)r * prog.n_prep + e];
  if (rt_doubles) {   // (uniform) the forest's compact tables: on their way while the pre-pass runs; its last barrier publishes them
    const uint4 *src = (const uint4 *)rt_src;
    uint4 *dst = (uint4 *)(smem + thr_at);
    for (uint32_t i = threadIdx.x; i < rt_doubles / 2; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  PrepScratch sc{s_vals, vals_cap, s_int, 0ull, {0, 0, 0, 0, 0, 0}};
#ifdef MRK_PHASE_CLOCKS
  sc.clk = clock64();
#endif
  prepass_request(st, prog, b, r, rq, s_tab, (uint32_t)rq.arena_begin, s_po, sc);
  const int og = (int)threadIdx.x / item_lanes, il = (int)threadIdx.x % item_lanes;
  for (int base = slice_lo; base < slice_hi; base += item_lanes) {
    const int i = base + il;
    const int gi0 = rq.item_begin + i;
    const bool active = i < slice_hi && gi0 >= b.item_lo && gi0 < b.item_hi;
    if (!__any(active)) continue;  // wavefront-uniform
    const int gi = active ? gi0 : rq.item_begin;
    assemble_item<SPLIT>(st, prog, b, gi, r, rq, s_tab, (uint32_t)rq.arena_begin, s_po, make_sink(gi, r, active, s_thr, s_thr_all), og, op_split);
  }
  MRK_PHASE(sc.clk, sc.acc[5]);
#ifdef MRK_PHASE_CLOCKS
  if (threadIdx.x == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&mrk_phase_clocks[i], sc.acc[i]);
    atomicAdd(&mrk_phase_clocks[6], 1ull);
  }
#endif
}


// One lane per candidate across many workgroups (requests too large for one workgroup, tables that do not fit the fused
// kernel's LDS): the tables were built by a previous pre-pass launch and lie in the HBM arena.  Straight into the scorer's
// binned tile.
// lds_entries: dynamic LDS of the launch, in table entries (0: none).  A workgroup whose lanes all belong to ONE request -
// every workgroup of a C4 batch - copies that request's finished tables into LDS first (coalesced, a few KB out of L2) when
// they fit: every table lookup of its candidates (interacted_with: up to IW_TOK tokens x fields, a dependent trip each) is
// then an LDS round trip instead of one to L2 / HBM.  Same-box: c4x (4 M candidates) assembly 0.99 -> see LOG.md.
template <bool F64, typename QS = QsDyn, typename Prog>
__device__ __forceinline__ void assemble_cells_body(const StoreDev &st, const Prog &prog, const BatchDev &b, const QsDev &q, uint16_t *cells,
                                                    uint32_t lds_entries) {
  __shared__ __align__(16) double s_thr[ASM_THREADS / 64][2 * QS_LDS_THR];
  extern __shared__ __align__(16) unsigned long long s_tab_copy[];
  const int wg_lo = b.item_lo + (int)blockIdx.x * ASM_THREADS;
  if (wg_lo >= b.item_hi) return;                  // (uniform)
#ifdef MRK_PHASE_CLOCKS
  unsigned long long clk_t = clock64(), clk_acc[2] = {0, 0};
#endif
  const int wg_hi = min(wg_lo + ASM_THREADS, b.item_hi) - 1;
  const int r_lo = (int)b.item_req[wg_lo], r_hi = (int)b.item_req[wg_hi];
  bool in_lds = false;
  uint32_t tab_sub = 0;
  if (lds_entries > 0u && r_lo == r_hi && prog.n_prep > 0) {
    const PrepOut *po = &b.prep_out[(size_t)r_lo * prog.n_prep];
    const uint32_t a0 = b.reqs[r_lo].arena_begin;
    uint32_t n_ent = 0;
    for (int e = 0; e < prog.n_prep; ++e) n_ent = max(n_ent, po[e].tab_off - a0 + po[e].tab_cap);
    in_lds = n_ent <= lds_entries;                 // (uniform: the whole workgroup takes one branch)
    if (in_lds) {
      const unsigned long long *src = b.arena + (size_t)a0;
      for (uint32_t i = threadIdx.x; i < n_ent; i += ASM_THREADS) s_tab_copy[i] = src[i];
      tab_sub = a0;
      __syncthreads();
    }
  }
  const int gi0 = wg_lo + (int)threadIdx.x;
  const bool active = gi0 < b.item_hi;
  if (!__any(active)) return;                      // whole wavefront past the end
  const int gi = active ? gi0 : b.item_hi - 1;     // lanes without an item ride along on a missing record
  const int r = (int)b.item_req[gi];
  const ReqDev rq = b.reqs[r];
  CellSink<F64, QS> sink{q, cells + (size_t)(gi / QS_TILE_ROWS) * qs_n_views<QS>(q) * QS_TILE_ROWS + (gi % QS_TILE_ROWS), &b.status[r],
                     (qs_lds_double *)s_thr[threadIdx.x >> 6], active};
  // (two instantiations: through a selected pointer the table accesses would be flat instead of ds / global operations)
  MRK_PHASE(clk_t, clk_acc[0]);
  if (in_lds) assemble_item(st, prog, b, gi, r, rq, s_tab_copy, tab_sub, &b.prep_out[(size_t)r * prog.n_prep], sink);
  else assemble_item(st, prog, b, gi, r, rq, b.arena, 0u, &b.prep_out[(size_t)r * prog.n_prep], sink);
  MRK_PHASE(clk_t, clk_acc[1]);
#ifdef MRK_PHASE_CLOCKS
  if (threadIdx.x == 0) {
    atomicAdd(&mrk_phase_clocks[0], clk_acc[0]);   // table copy + request lookup
    atomicAdd(&mrk_phase_clocks[5], clk_acc[1]);   // per-item assembly
    atomicAdd(&mrk_phase_clocks[6], 1ull);
  }
#endif
}

// The item-parallel kernel of a model whose forest signature is a compile-time constant and whose threshold tables fit in LDS
// next to a request's hash tables: PERSISTENT workgroups (the grid is what the chip holds at once; each walks blocks of blockDim.x
// candidates) that load ALL threshold tables once - 48 KB for the 24 Ranklens columns - instead of every wavefront staging every
// column's table for every 64 candidates (750 B of L2 -> LDS traffic per candidate: twice the record itself), and bin through
// CellSinkRT.  A workgroup whose block belongs to one request keeps that request's finished hash tables in LDS as before, and
// keeps them across blocks of the same request (config 4: one copy per workgroup for the whole launch).
// Dynamic LDS: [compact threshold tables: QS::rt_total x 8 B][hash tables: lds_entries x 8 B].
template <bool F64, typename QS, typename Prog>
__device__ __forceinline__ void assemble_cells_rt_body(const StoreDev &st, const Prog &prog, const BatchDev &b, const QsDev &q, uint16_t *cells,
                                                       uint32_t lds_entries) {
  extern __shared__ __align__(16) uint8_t smem_rt[];
  constexpr uint32_t THR = QS::rt_total;   // doubles of the compact tables (an even number)
  qs_lds_double *s_thr = (qs_lds_double *)smem_rt;
  unsigned long long *s_tab_copy = (unsigned long long *)(smem_rt + (size_t)THR * 8);
  const int nthr = (int)blockDim.x;
  {
    const uint4 *src = (const uint4 *)q.thr_rt;
    uint4 *dst = (uint4 *)smem_rt;
    for (uint32_t i = threadIdx.x; i < THR / 2; i += (uint32_t)nthr) dst[i] = src[i];
  }
  __syncthreads();
  const int n_blocks = (b.item_hi - b.item_lo + nthr - 1) / nthr;
  int have_req = -1;   // the request whose tables s_tab_copy holds
#ifdef MRK_PHASE_CLOCKS
  unsigned long long clk_t = clock64(), clk_acc[2] = {0, 0};
#endif
  for (int blk = (int)blockIdx.x; blk < n_blocks; blk += (int)gridDim.x) {
    const int wg_lo = b.item_lo + blk * nthr;
    const int wg_hi = min(wg_lo + nthr, b.item_hi) - 1;
    const int r_lo = (int)b.item_req[wg_lo], r_hi = (int)b.item_req[wg_hi];
    bool in_lds = false;
    uint32_t tab_sub = 0;
    if (lds_entries > 0u && r_lo == r_hi && prog.n_prep > 0) {
      const PrepOut *po = &b.prep_out[(size_t)r_lo * prog.n_prep];
      const uint32_t a0 = b.reqs[r_lo].arena_begin;
      uint32_t n_ent = 0;
      for (int e = 0; e < prog.n_prep; ++e) n_ent = max(n_ent, po[e].tab_off - a0 + po[e].tab_cap);
      in_lds = n_ent <= lds_entries;                 // (uniform: the whole workgroup takes one branch)
      if (in_lds) {
        if (have_req != r_lo) {
          __syncthreads();                           // the previous block's lookups are over
          const unsigned long long *src = b.arena + (size_t)a0;
          for (uint32_t i = threadIdx.x; i < n_ent; i += (uint32_t)nthr) s_tab_copy[i] = src[i];
          __syncthreads();
          have_req = r_lo;
        }
        tab_sub = a0;
      }
    }
    const int gi0 = wg_lo + (int)threadIdx.x;
    const bool active = gi0 < b.item_hi;
    MRK_PHASE(clk_t, clk_acc[0]);
    if (__any(active)) {                             // (else: a whole wavefront past the end)
      const int gi = active ? gi0 : b.item_hi - 1;   // lanes without an item ride along on a missing record
      const int r = (int)b.item_req[gi];
      const ReqDev rq = b.reqs[r];
      CellSinkRT<F64, QS> sink{q, cells + (size_t)(gi / QS_TILE_ROWS) * QS::n_views * QS_TILE_ROWS + (gi % QS_TILE_ROWS), &b.status[r], s_thr, active};
      // (two instantiations: through a selected pointer the table accesses would be flat instead of ds / global operations)
      if (in_lds) assemble_item(st, prog, b, gi, r, rq, s_tab_copy, tab_sub, &b.prep_out[(size_t)r * prog.n_prep], sink);
      else assemble_item(st, prog, b, gi, r, rq, b.arena, 0u, &b.prep_out[(size_t)r * prog.n_prep], sink);
    }
    MRK_PHASE(clk_t, clk_acc[1]);
  }
#ifdef MRK_PHASE_CLOCKS
  if (threadIdx.x == 0) {
    atomicAdd(&mrk_phase_clocks[0], clk_acc[0]);
    atomicAdd(&mrk_phase_clocks[5], clk_acc[1]);
    atomicAdd(&mrk_phase_clocks[6], 1ull);
  }
#endif
}

// rank_fused_body writing ClickthroughQuery's row-major f64 matrix
template <bool SPLIT = false, typename Prog>
__device__ __forceinline__ void rank_fused_matrix_body(const StoreDev &st, const Prog &prog, const BatchDev &b, uint32_t tab_entries, int vals_cap, int mode = 1) {
  rank_fused_body<SPLIT>(st, prog, b, tab_entries, vals_cap, 0u, mode,
                  [&](int gi, int, bool active, qs_lds_double *, qs_lds_double *) { return MatrixSink{b.matrix + (size_t)gi * prog.dim, active}; });
}

// the hot-path instance of rank_fused_body: straight into the scorer's binned tile
template <bool F64, bool SPLIT = false, typename QS = QsDyn, typename Prog>
__device__ __forceinline__ void rank_fused_cells_body(const StoreDev &st, const Prog &prog, const BatchDev &b, uint32_t tab_entries,
                                                      int vals_cap, const QsDev &q, uint16_t *cells, int mode = 1) {
  if constexpr (qs_fused_rt<QS, SPLIT>()) {   // the forest's compact tables fit: resident, searched four columns at a time (CellSinkRT)
    rank_fused_body<SPLIT>(st, prog, b, tab_entries, vals_cap, qs_thr_cap<QS>(q), mode, [&](int gi, int r, bool active, qs_lds_double *, qs_lds_double *s_all) {
      return CellSinkRT<F64, QS, !SPLIT>{q, cells + (size_t)(gi / QS_TILE_ROWS) * QS::n_views * QS_TILE_ROWS + (gi % QS_TILE_ROWS), &b.status[r], s_all, active};
    }, 0u, q.thr_rt, QS::rt_total);
  } else {
    rank_fused_body<SPLIT>(st, prog, b, tab_entries, vals_cap, qs_thr_cap<QS>(q), mode, [&](int gi, int r, bool active, qs_lds_double *s_thr, qs_lds_double *) {
      return CellSink<F64, QS>{q, cells + (size_t)(gi / QS_TILE_ROWS) * qs_n_views<QS>(q) * QS_TILE_ROWS + (gi % QS_TILE_ROWS), &b.status[r], s_thr, active};
    });
  }
}

// ---- ONE launch for a handful of small requests (mrk_rank, the serving queue): pre-pass + assembly + forest + ordering in
// the request's workgroup.  The binned tile never leaves LDS: the assembly writes the request's cells into the slab the
// scorer reads (LDS byte 0 on: `ds_read_addtid_b32` addresses it through M0), the workgroup's wavefronts split the trees
// (qs_score_tile_split: the additions of qs_score_split_kernel, in tree order), the rows' owners order the scores by
// counting (sort_kernel's rule) and write scores / order / status where the host reads them - pinned memory - so the
// whole of Ranker.rerank (ml/Ranker.scala:27-83) is one dispatch and no copy.  Requests of <= QS_TILE_ROWS candidates.
// Dynamic LDS: [slab: slab_bytes = V x 256][the request's status word, 16 B][the regions of rank_fused_body | afterwards:
// leaf values, exit-leaf indices, sort keys of the scoring phase].  The status word lives in LDS too (the batch view the
// assembly sees has its `status` pointer bent there): no global atomic, nothing to wait for before it is copied out.
template <bool F64, typename QS = QsDyn, typename Prog>
__device__ __forceinline__ void rank_one_body(const StoreDev &st, const Prog &prog, const BatchDev &b, uint32_t tab_entries, int vals_cap,
                                              const QsDev &q, const QsForestDev &f, int mode, const OneOut &out, int blk = -1) {
  extern __shared__ __align__(16) uint8_t smem_base[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int r = blk < 0 ? (int)blockIdx.x : blk;
  const ReqDev rq = b.reqs[r];
  const uint32_t slab_bytes = (uint32_t)qs_n_views<QS>(q) * (QS_TILE_ROWS * 2);
  for (uint32_t i = tid; i < slab_bytes / 4 + 4; i += nthr) ((uint32_t *)smem_base)[i] = 0u;  // rows past the request's last candidate; the status word
  int32_t *s_status = (int32_t *)(smem_base + slab_bytes);
  BatchDev bl = b;
  bl.status = s_status - r;   // &bl.status[r] is the LDS word
  __syncthreads();
  if constexpr (qs_fused_rt<QS, true>()) {   // the compact tables resident (no queue: the ops are shared out at run time)
    rank_fused_body<true>(st, prog, bl, tab_entries, vals_cap, qs_thr_cap<QS>(q), mode & 255, [&](int gi, int rr, bool active, qs_lds_double *, qs_lds_double *s_all) {
      return CellSinkRT<F64, QS, false>{q, (uint16_t *)smem_base + (gi - rq.item_begin), &bl.status[rr], s_all, active};
    }, slab_bytes + 16, q.thr_rt, QS::rt_total, r);
  } else {
    rank_fused_body<true>(st, prog, bl, tab_entries, vals_cap, qs_thr_cap<QS>(q), mode & 255, [&](int gi, int rr, bool active, qs_lds_double *s_thr, qs_lds_double *) {
      return CellSink<F64, QS>{q, (uint16_t *)smem_base + (gi - rq.item_begin), &bl.status[rr], s_thr, active, /*vm_stores=*/false};
    }, slab_bytes + 16, nullptr, 0, r);
  }
  // (qs_score_tile_split starts with a barrier: the slab is complete, the assembly's LDS regions are free)
  constexpr int TREE_LEAF_BYTES = QS_LEAVES * (F64 ? 8 : 4);
  const int nw = nthr >> 6;
  uint8_t *s_leaf = smem_base + slab_bytes + 16;
  uint8_t *s_idx = s_leaf + 8 * nw * TREE_LEAF_BYTES;
  unsigned long long *s_key = (unsigned long long *)(s_idx + 8 * nw * QS_TILE_ROWS);
  const double score = qs_score_tile_split<F64>(smem_base, s_leaf, s_idx, f, nw);
  const int n = rq.n_items;
  if (tid < n) {
    out.scores[rq.item_begin + tid] = score;
    s_key[tid] = sort_key(score);
  }
  __syncthreads();
  if (tid < n) {  // the place of candidate `tid`: the pairs (key, index) that precede its own
    const unsigned long long mine = s_key[tid];
    int before = 0;
    for (int j = 0; j < n; ++j) {
      const unsigned long long kj = s_key[j];
      before += (kj < mine || (kj == mine && j < tid)) ? 1 : 0;
    }
    out.order[rq.item_begin + before] = tid;
  }
  if (tid == 0) {
    out.status[r] = *s_status;
    out.status[out.n_req_pad + r] = out.load_status ? out.load_status[r] : 0;
  }
}

// ---- the same idea for FULL batches of small requests (c2: 3 840 requests x 100 candidates): the request's workgroup
// assembles its binned tile (one 128-row tile per request, in global memory: the assembly's LDS is full of tables), then
// pulls the tile back into LDS - over the tables it no longer needs -, scores it with its own wavefronts and orders the
// scores.  One launch instead of three, and - the point - every CU holds workgroups in BOTH kinds of phase at any time:
// the assembly waits on memory (55 % of a wavefront's life, profiles/r03_i), the forest is pure instruction issue; as two
// kernels they met on a CU only by accident of two streams (the assembly kernel's workgroups take the whole LDS).
// Dynamic LDS: max(rank_fused_body's regions, [slab V x 256][16][leaf values, exit-leaf indices of 8 x nw trees][128 sort keys]).
template <bool F64, typename QS = QsDyn, typename Prog>
__device__ __forceinline__ void rank_fused_score_body(const StoreDev &st, const Prog &prog, const BatchDev &b, uint32_t tab_entries, int vals_cap,
                                                      const QsDev &q, const QsForestDev &f, uint16_t *cells) {
  extern __shared__ __align__(16) uint8_t smem_base[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int r = blockIdx.x;
  const ReqDev rq = b.reqs[r];
  const uint32_t slab_bytes = (uint32_t)qs_n_views<QS>(q) * (QS_TILE_ROWS * 2);
  uint16_t *tile = cells + (size_t)r * qs_n_views<QS>(q) * QS_TILE_ROWS;
  // rows past the request's last candidate (their cells only have to be harmless)
  for (uint32_t i = tid; i < (uint32_t)qs_n_views<QS>(q) * QS_TILE_ROWS; i += nthr)
    if ((int)(i % QS_TILE_ROWS) >= rq.n_items) tile[i] = 0;
  if constexpr (qs_fused_rt<QS, false>()) {
    rank_fused_body<false>(st, prog, b, tab_entries, vals_cap, qs_thr_cap<QS>(q), 1, [&](int gi, int rr, bool active, qs_lds_double *, qs_lds_double *s_all) {
      return CellSinkRT<F64, QS>{q, tile + (gi - rq.item_begin), &b.status[rr], s_all, active};
    }, 0u, q.thr_rt, QS::rt_total);
  } else {
    rank_fused_body<false>(st, prog, b, tab_entries, vals_cap, qs_thr_cap<QS>(q), 1, [&](int gi, int rr, bool active, qs_lds_double *s_thr, qs_lds_double *) {
      return CellSink<F64, QS>{q, tile + (gi - rq.item_begin), &b.status[rr], s_thr, active};
    });
  }
  // the tile has reached L2 (this CU's L1 holds none of its lines: nothing has read them), every table is dead
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const uint4 *src = (const uint4 *)tile;
    uint4 *dst = (uint4 *)smem_base;
    for (uint32_t i = tid; i < slab_bytes / 16; i += nthr) dst[i] = src[i];
  }
  constexpr int TREE_LEAF_BYTES = QS_LEAVES * (F64 ? 8 : 4);
  const int nw = nthr >> 6;
  uint8_t *s_leaf = smem_base + slab_bytes + 16;
  uint8_t *s_idx = s_leaf + 8 * nw * TREE_LEAF_BYTES;
  unsigned long long *s_key = (unsigned long long *)(s_idx + 8 * nw * QS_TILE_ROWS);
  const double score = qs_score_tile_split<F64>(smem_base, s_leaf, s_idx, f, nw);   // (starts with a barrier: the slab is complete)
  const int n = rq.n_items;
  if (tid < n) {
    b.scores[rq.item_begin + tid] = score;
    s_key[tid] = sort_key(score);
  }
  __syncthreads();
  if (tid < n) {
    const unsigned long long mine = s_key[tid];
    int before = 0;
    for (int j = 0; j < n; ++j) {
      const unsigned long long kj = s_key[j];
      before += (kj < mine || (kj == mine && j < tid)) ? 1 : 0;
    }
    b.order[rq.item_begin + before] = tid;
  }
}

// ---- the persistent form of the same: ONE workgroup that stays on its CU and serves the requests the host publishes in
// its slot (main/command/Serve.scala:130-150 + api/routes/RankApi.scala:25-41: the process that answers POST /rank - here
// the request path contains no HIP call at all).  Per request: lane 0 polls `seq` in pinned memory; the workgroup copies
// the request's input block from pinned memory into its device scratch (one coalesced pass over PCIe; the assembly reads
// its inputs many times), drops its CU's cached lines of that scratch, runs rank_one_body - which writes scores / order /
// status into the slot's pinned output block - and acknowledges.  A workgroup that has seen no request for `idle_ticks`
// announces `exited` and leaves (a persistent kernel must never outlive its use: hipFree and friends wait for it); the host
// relaunches it with the next request.  If a request slips in between the announcement and the exit it is still served
// (device: store exited, fence, read seq; host: store seq, fence, read exited - one side sees the other).
template <bool F64, typename QS = QsDyn, typename Prog>
__device__ __forceinline__ void rank_serve_body(const StoreDev &st, const Prog &prog, const QsDev &q, const QsForestDev &f, const ServeGangDev &g) {
  extern __shared__ __align__(16) uint8_t smem_base[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const ServeSlotDev s = g.slots[blockIdx.x];
  const uint32_t slab_bytes = (uint32_t)qs_n_views<QS>(q) * (QS_TILE_ROWS * 2);
  volatile uint32_t *s_word = (volatile uint32_t *)(smem_base + slab_bytes + 8);   // [0] seq | STOP, [1] leave after this request
  // the slot's last ANSWERED request: whatever the host has published beyond it is this workgroup's first request (a gang is
  // launched as a whole: most of its slots have nothing pending, the one that asked for the launch has)
  uint32_t last = __hip_atomic_load(&s.ctl->ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  unsigned long long idle_since = wall_clock64();
  const unsigned long long born = idle_since;
  for (;;) {
    // The poller's test must not be provably the test of any `tid == 0` block further down: the compiler then threads the
    // back edge of the lanes that fail both straight to the barrier below - an inner loop the poller's lane never re-enters,
    // in which lanes 1 .. 63 of its wavefront rank the same request for ever (measured: round 6, the first gang build).
    int poller = tid;
    asm volatile("" : "+v"(poller));
    if (poller == 0) {
      uint32_t seq = last, leave = 0, stop = 0;
      for (;;) {
        // A slot under sustained traffic never idles (the host hands out the most recently used slot first), and a
        // resident kernel stalls every hipFree / reallocation / device-wide sync of the process: past `life_ticks` the
        // workgroup takes its leave exactly as if it had been told to (a request already published is served first).
        const unsigned long long now = wall_clock64();
        const bool old = now - born > g.life_ticks;
        seq = __hip_atomic_load(&s.ctl->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (seq != last && !old) break;
        const bool told = old || __hip_atomic_load(&s.ctl->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
        if (!told && now - idle_since > g.idle_ticks) {   // this slot is idle - is the gang?  (its workgroups leave together)
          const unsigned long long seen = __hip_atomic_load(g.clock, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (seen > idle_since && now - seen <= g.idle_ticks) idle_since = seen;
        }
        if (told || now - idle_since > g.idle_ticks) {
          __hip_atomic_store(&s.ctl->exited, g.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __threadfence_system();
          seq = __hip_atomic_load(&s.ctl->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (seq != last) leave = 1;   // a request slipped in: serve it, then leave
          else stop = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      s_word[0] = stop ? 0xffffffffu : seq;
      s_word[1] = leave;
    }
    __syncthreads();
    const uint32_t seq = s_word[0], leave = s_word[1];
    if (seq == 0xffffffffu) return;
    const unsigned long long t_seen = wall_clock64();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // the header and the block were written before `seq`
    const ServeCtl *ctl = s.ctl;
    const uint32_t in_bytes = ctl->in_bytes;
    {
      const uint4 *src = (const uint4 *)s.in_host;
      uint4 *dst = (uint4 *)s.in_dev;
      for (uint32_t i = tid; i < (in_bytes + 15) / 16; i += nthr) dst[i] = src[i];
    }
    BatchDev b = {};
    b.reqs = (const ReqDev *)(s.in_dev + ctl->o_reqs);
    b.n_req = 1;
    b.total_items = (int32_t)ctl->total_items;
    b.item_lo = 0;
    b.item_hi = b.total_items;
    b.item_slot = (const int32_t *)(s.in_dev + ctl->o_slot);
    b.item_req = (const uint32_t *)(s.in_dev + ctl->o_ireq);
    b.consts = (const double *)(s.in_dev + ctl->o_consts);
    b.irf = (const int32_t *)(s.in_dev + ctl->o_irf);
    b.prep_out = (PrepOut *)(s.in_dev + ctl->o_prep);
    const uint32_t tab_entries = ctl->tab_entries, vals_cap = ctl->vals_cap, mode = ctl->mode;
    // the copy must have reached L2 and this CU must not serve the previous request's lines of the scratch from its
    // vector or scalar cache
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t_in = wall_clock64();
    const unsigned long long c_in = clock64();   // shader cycles next to the 100 MHz wall clock: the clock the request actually ran at
    rank_one_body<F64, QS>(st, prog, b, tab_entries, (int)vals_cap, q, f, (int)mode, s.out, 0);
    const unsigned long long c_ranked = clock64();
    const unsigned long long t_ranked = wall_clock64();
    __threadfence_system();   // every lane's results are in host memory before the acknowledgement
    __syncthreads();
    if (tid == 0) {
      // where the request's time went on the device, in 100 MHz ticks (mrk_serve_stats): input copy + cache drops, ranking,
      // result writes reaching host memory
      unsigned long long *clk = (unsigned long long *)(s.out.status + 16);
      clk[0] = t_in - t_seen;
      clk[1] = t_ranked - t_in;
      clk[2] = wall_clock64() - t_ranked;
      clk[3] = c_ranked - c_in;
      __hip_atomic_store(g.clock, wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the gang's last activity
      __threadfence_system();
      __hip_atomic_store(&s.ctl->ack, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (leave) return;
    last = seq;
    idle_since = wall_clock64();
  }
}

}  // namespace

}  // namespace mrk
