// Device-side helpers of the bit-vector scorer shared by score_qs.hip (standalone binning of an f64
// matrix) and rank.hip (feature assembly writing binned cells directly).  Format: forest.hpp.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.hpp"

namespace mrk {

constexpr int QS_TILE_ROWS = 128;  // rows of one scorer wavefront (two per lane)

struct QsDev {             // what a kernel needs to bin a value of any matrix column
  const QsFeature *feats;  // n_feats
  const QsView *views;
  const double *thr;
  const double *thr_rt;    // the compact tables (QsSig::rt_off): what a resident-table sink copies into LDS
  int32_t n_feats;
  int32_t n_views;
  uint32_t thr_cap;        // doubles per LDS staging buffer: the longest staged table rounded up to QS_STAGE_CHUNK
  uint32_t rt_doubles;     // doubles of the compact tables (0: no signature): what the launchers size a resident region by
};

typedef __attribute__((address_space(3))) double qs_lds_double;  // forces ds_read for tables staged in LDS
constexpr uint32_t QS_LDS_THR = 256;  // thresholds of one column a wavefront stages in LDS (LightGBM: max_bin - 1 = 254)
// descriptors are read through the constant address space: uniform loads from it are always scalar loads
typedef const __attribute__((address_space(4))) QsFeature *QsFeatureK;

// The library's own preprocessing of a dense-row value.  ok = false iff XGBoost would reject it.
template <bool F64>
__device__ __forceinline__ double qs_prep(double x, bool &ok) {
  ok = true;
  if constexpr (F64) {
    // LightGBM RowFunctionFromDenseMatric keeps a cell only if |x| > kZeroThreshold (1e-35f) or NaN
    const double kZero = (double)1e-35f;
    return (fabs(x) > kZero || x != x) ? x : 0.0;
  } else {
    // ltrlib narrows Double -> Float before DMatrix; XGBoost rejects +-inf ("Input data contains `inf`")
    const float f = (float)x;
    ok = !__builtin_isinf(f);
    return (double)f;
  }
}

// bin = number of thresholds strictly below x (LightGBM: x <= t goes left) / not above x (XGBoost: x < t).
// T: sorted, `len` entries, in global memory (const double *) or LDS (qs_lds_double *).
template <bool F64, typename P>
__device__ __forceinline__ uint32_t qs_bin_search(P T, uint32_t len, double x) {
  // branch-free lower bound: `n` (hence the trip count) is the same in every lane, every read is inside the table
  uint32_t pos = 0;
  if (len) {
#pragma unroll 1
    for (uint32_t n = len; n > 1;) {
      const uint32_t half = n >> 1;
      const double t = T[pos + half - 1];
      const bool below = F64 ? (t < x) : (t <= x);
      pos = below ? pos + half : pos;
      n -= half;
    }
    const double t = T[pos];
    pos += (F64 ? (t < x) : (t <= x)) ? 1u : 0u;
  }
  return pos;
}

// The same lower bound over a table STAGED IN LDS: the staged copy is a whole number of 128-entry chunks, +inf past the table's
// end (forest.cpp pads the tables), at most 256 entries - so the search is 7 or 8 steps with COMPILE-TIME strides: a step is one
// ds_read_b64 at an immediate offset from the running address, one compare, one conditional add; no loop, no scalar
// arithmetic (the run-time loop cost ~6 VALU + 4 SALU per step, 24 columns x 8 steps per candidate).
// `len` may be the table's length rounded up to whole chunks (what a kernel that holds the forest's view signature as
// constants knows at compile time: the padding answers like the table's end); `real_len()` - the exact length, a scalar
// load in such a kernel - is asked for by XGBoost's clamp only.
template <bool F64, typename RealLen>
__device__ __forceinline__ uint32_t qs_bin_search_staged(qs_lds_double *T, uint32_t len, double x, RealLen real_len) {
  if (len == 0u) return 0u;
  auto below = [&](double t) { return F64 ? (t < x) : (t <= x); };
  uint32_t pos = 0;
  if (len > 128u) pos = below(T[127]) ? 128u : 0u;   // (uniform)
  qs_lds_double *p = T + pos;
#pragma unroll
  for (int h = 64; h >= 1; h >>= 1) {
    const bool b = below(p[h - 1]);
    p += b ? h : 0;
  }
  pos = (uint32_t)(p - T);
  pos += below(p[0]) ? 1u : 0u;
  // XGBoost's test is t <= x: x = +inf walks through the +inf padding (such a request fails with ST_XGB_INF anyway, but the
  // cell it leaves behind stays inside the column's bins, as qs_bin_search's does)
  if constexpr (!F64) pos = min(pos, (uint32_t)real_len());
  return pos;
}

// (Measured and rejected, round 3: a three-level 8-ary search over the staged table - 17 independent LDS reads in 3 trips
// instead of 8 dependent ones - made the c2 assembly kernel 40 % SLOWER (0.284 -> 0.406 ms, gpurun_out r03_j): after the
// first level the lanes' ranges start at multiples of 256 B, i.e. in the same LDS bank, and the LDS pipe - shared by the 16
// wavefronts of a CU - is the resource this kernel is short of, not the round trips of one wavefront.)
// every view of the column: (view index, cell) -> emit
// one view's cell of a binned value
template <bool F64>
__device__ __forceinline__ uint32_t qs_view_cell(uint32_t kind, double x, uint32_t pos, uint32_t zero_bin, bool isn, bool isz) {
  uint32_t cell;
  if (kind == QV_CAT) {
    // the category id; the node's bitset is consulted by the scorer
    if (isn) cell = QS_CAT_NAN;
    else if constexpr (F64) {
      // LightGBM Tree::CategoricalDecision: int(fval) < 0 goes right, like NaN
      const int iv = (int)x;  // v_cvt_i32_f64 saturates
      cell = iv < 0 ? (uint32_t)QS_CAT_NAN : (iv >= (int)QS_CAT_BEYOND ? (uint32_t)QS_CAT_BEYOND : (uint32_t)iv);
    } else {
      // XGBoost common::Decision: negative or >= 2^24 is an invalid category (goes left)
      if (x < 0.0 || x >= 16777216.0) cell = QS_CAT_INVALID;
      else {
        const int iv = (int)x;
        cell = iv >= (int)QS_CAT_BEYOND ? (uint32_t)QS_CAT_BEYOND : (uint32_t)iv;
      }
    }
  } else if (kind == QV_NAN_ZERO) {
    cell = isn ? zero_bin : pos;  // MissingType::None: NaN is compared as 0.0
  } else {
    const bool miss = (kind >= QV_MISS_RIGHT) ? (isn || isz) : isn;
    const uint32_t mval = (kind & 1) ? 0u : (uint32_t)QS_RIGHT;  // *_LEFT kinds are odd
    cell = miss ? mval : pos;
  }
  return cell;
}

template <bool F64, typename Emit>
__device__ __forceinline__ void qs_emit_views(double x, uint32_t pos, const QsFeature ft, const QsView *__restrict__ views, Emit emit) {
  const bool isn = x != x;
  const bool isz = x == 0.0;
#pragma unroll 1
  for (uint32_t v = ft.view_begin; v < ft.view_end; ++v) {  // 1 - 2 views per column: unrolling only grows the code
    const uint32_t kind = (ft.view_kinds >> (4u * (v - ft.view_begin))) & 15u;  // no load: the descriptor carries the kinds
    emit(v, qs_view_cell<F64>(kind, x, pos, ft.zero_bin, isn, isz));
  }
}

// the same for a column whose views are compile-time constants (QsSig): the loop unrolls, the kinds fold, `zero_bin()` -
// a scalar load - is asked for by QV_NAN_ZERO views only
constexpr int QS_SIG_MAX_VIEWS = 6;
template <bool F64, typename ZeroBin, typename Emit>
__device__ __forceinline__ void qs_emit_views_sig(double x, uint32_t pos, const QsSig s, ZeroBin zero_bin, Emit emit) {
  const bool isn = x != x;
  const bool isz = x == 0.0;
#pragma unroll
  for (int i = 0; i < QS_SIG_MAX_VIEWS; ++i) {
    if ((uint32_t)s.view_begin + i >= s.view_end) break;
    const uint32_t kind = (s.view_kinds >> (4u * i)) & 15u;
    emit((uint32_t)s.view_begin + i, qs_view_cell<F64>(kind, x, pos, kind == QV_NAN_ZERO ? (uint32_t)zero_bin() : 0u, isn, isz));
  }
}

// Bins one matrix value `x` of a column described by `ft` into every view of that column and hands
// (view index, cell) to `emit`.  F64: LightGBM semantics, else XGBoost.  Returns false iff the value
// is one XGBoost rejects (+-inf after the Double -> Float narrowing ltrlib performs).
template <bool F64, typename Emit>
__device__ __forceinline__ bool qs_bin_column(double x, const QsFeature ft, const QsView *__restrict__ views,
                                              const double *__restrict__ thr, Emit emit) {
  bool ok;
  x = qs_prep<F64>(x, ok);
  qs_emit_views<F64>(x, qs_bin_search<F64>(thr + ft.thr_off, ft.thr_len, x), ft, views, emit);
  return ok;
}

// ---------------------------------------------------------------------------------- scoring (shared by score_qs.hip and
// the one-launch rank kernel of rank_device.hpp)
typedef short short2v __attribute__((ext_vector_type(2)));

struct QsForestDev {   // the bit-vector image of a forest (forest.hpp "qs"): what a scoring kernel reads
  const uint32_t *nodes;       // per tree QS_TREE_WORDS dwords; one all-zero tree after the last
  const uint8_t *leaves;       // per tree QS_LEAVES values (f64 | f32), position order
  const QsCatNode *cat_nodes;
  const uint32_t *cat_bits;
  int32_t n_trees;
  int32_t n_views;
  double base;                 // XGBoost base margin
};

// categorical nodes of one tree (rare path): the cell is the category id, tested against the node's bitset
template <bool F64>
__device__ __forceinline__ uint32_t qs_cat_pair(const QsCatNode &cn, uint32_t cc, const uint32_t *__restrict__ cat_bits) {
  const bool dl = (cn.view_dl >> 16) != 0;
  uint32_t removed = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t cat = (cc >> (16 * h)) & 0xffffu;
    const uint32_t w = cat >> 5;
    bool in = false;
    if (cat < QS_CAT_BEYOND && w < cn.bits_words) in = (cat_bits[cn.bits_begin + w] >> (cat & 31)) & 1u;
    bool right;
    if constexpr (F64) right = !in;  // LightGBM: member -> left; NaN / negative / unknown -> right
    else right = cat == QS_CAT_NAN ? !dl : in;  // XGBoost: member -> right; invalid / unknown -> left
    if (right) removed |= cn.mm & (0xffffu << (16 * h));
  }
  return removed;
}

struct QsNodeRegs {
  uint32_t kk[QS_SLOTS], mv[QS_SLOTS];  // kk[QS_SLOTS-1] = categorical word
};

__device__ __forceinline__ void qs_load_nodes(QsNodeRegs &r, const uint32_t *__restrict__ nd) {
#pragma unroll
  for (int s = 0; s < QS_SLOTS; ++s) {
    r.kk[s] = nd[s];
    r.mv[s] = nd[QS_SLOTS + s];
  }
}

// the same through the constant address space: uniform loads from it are always scalar loads, whatever else the kernel
// writes (the forest image is read-only for every kernel)
typedef const __attribute__((address_space(4))) uint32_t *QsNodesK;
__device__ __forceinline__ void qs_load_nodes_k(QsNodeRegs &r, QsNodesK nd) {
#pragma unroll
  for (int s = 0; s < QS_SLOTS; ++s) {
    r.kk[s] = nd[s];
    r.mv[s] = nd[QS_SLOTS + s];
  }
}

// One 128-row tile scored by the `nw` wavefronts of the calling workgroup (nw * 64 lanes, all of them call): the same
// arithmetic as qs_score_split_kernel - per chunk of 8 * nw trees every wavefront writes the exit-leaf indices of its
// trees to LDS, then row `tid` (tid < 128) adds the chunk's leaf values in tree order.  The tile's slab (V x 256 B)
// must start at LDS byte 0 (`ds_read_addtid_b32` addresses it through M0); s_leaf: 8 * nw trees x QS_LEAVES values,
// s_idx: 8 * nw x 128 B.  Returns the score of row threadIdx.x (meaningful for threadIdx.x < 128).
template <bool F64>
__device__ __forceinline__ double qs_score_tile_split(const uint8_t *smem, uint8_t *s_leaf, uint8_t *s_idx, const QsForestDev &f, int nw) {
  constexpr int LS = F64 ? 8 : 4;
  constexpr int TREE_LEAF_BYTES = QS_LEAVES * LS;
  const int CH = 8 * nw;
  const int tid = threadIdx.x, lane = tid & 63, nthr = nw * 64;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double acc64 = 0.0;
  float acc32 = (float)f.base;
  for (int c0 = 0; c0 < f.n_trees; c0 += CH) {
    const int nt = min(CH, f.n_trees - c0);
    __syncthreads();  // slab complete / previous chunk consumed
    {
      const uint4 *src = (const uint4 *)(f.leaves + (size_t)c0 * TREE_LEAF_BYTES);
      uint4 *dst = (uint4 *)s_leaf;
      for (int i = tid; i < nt * (TREE_LEAF_BYTES / 16); i += nthr) dst[i] = src[i];
    }
    for (int tt = wave; tt < nt; tt += nw) {
      const QsNodesK nd = (QsNodesK)(unsigned long long)(f.nodes + (size_t)(c0 + tt) * QS_TREE_WORDS);
      QsNodeRegs r;
      qs_load_nodes_k(r, nd);
      uint32_t c[QS_SLOTS - 1], mm[QS_SLOTS - 1];
#pragma unroll
      for (int s = 0; s < QS_SLOTS - 1; ++s)
        asm volatile("s_lshr_b32 m0, %2, 16\n\ts_pack_ll_b32_b16 %1, %2, %2\n\tds_read_addtid_b32 %0"
                     : "=v"(c[s]), "=s"(mm[s]) : "s"(r.mv[s]) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the reads above are invisible to the compiler
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < QS_SLOTS - 1; ++s)
        c[s] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, r.kk[s]) - __builtin_bit_cast(short2v, c[s]));
#pragma unroll
      for (int s = 0; s < QS_SLOTS - 1; ++s)
        c[s] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, c[s]) >> 15);
      uint32_t acc_a = 0, acc_b = 0;
#pragma unroll
      for (int s = 0; s < QS_SLOTS - 1; s += 2) {
        asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc_a) : "v"(c[s]), "s"(mm[s]));
        if (s + 1 < QS_SLOTS - 1) asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc_b) : "v"(c[s + 1]), "s"(mm[s + 1]));
      }
      uint32_t accn = acc_a | acc_b;
      const uint32_t catw = __builtin_amdgcn_readfirstlane(r.kk[QS_SLOTS - 1]);
      if (catw >> 24) {
        const QsCatNode *cn = f.cat_nodes + (catw & 0xffffffu);
        for (uint32_t j = 0; j < (catw >> 24); ++j) {
          const QsCatNode cnode = cn[j];
          accn |= qs_cat_pair<F64>(cnode, *(const uint32_t *)(smem + ((cnode.view_dl & 0xffffu) << 8) + lane * 4), f.cat_bits);
        }
      }
      const uint32_t inv = ~accn;
      const uint32_t pair = (uint32_t)__builtin_ctz(inv) | ((uint32_t)__builtin_ctz(inv >> 16) << 8);
      *(uint16_t *)(s_idx + tt * QS_TILE_ROWS + lane * 2) = (uint16_t)pair;  // rows 2 * lane, 2 * lane + 1
    }
    __syncthreads();
    if (tid < QS_TILE_ROWS) {  // row `tid`: the chunk's leaves, in tree order
      for (int tt = 0; tt < nt; ++tt) {
        const uint32_t li = s_idx[tt * QS_TILE_ROWS + tid];
        if constexpr (F64) acc64 += *(const double *)(s_leaf + tt * TREE_LEAF_BYTES + li * 8);
        else acc32 += *(const float *)(s_leaf + tt * TREE_LEAF_BYTES + li * 4);
      }
    }
  }
  return F64 ? acc64 : (double)acc32;
}

}  // namespace mrk
