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
  int32_t n_feats;
  int32_t n_views;
  uint32_t thr_cap;        // doubles per LDS staging buffer: the longest staged table rounded up to QS_STAGE_CHUNK
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

// every view of the column: (view index, cell) -> emit
template <bool F64, typename Emit>
__device__ __forceinline__ void qs_emit_views(double x, uint32_t pos, const QsFeature ft, const QsView *__restrict__ views, Emit emit) {
  const bool isn = x != x;
  const bool isz = x == 0.0;
#pragma unroll 1
  for (uint32_t v = ft.view_begin; v < ft.view_end; ++v) {  // 1 - 2 views per column: unrolling only grows the code
    const uint32_t kind = (ft.view_kinds >> (4u * (v - ft.view_begin))) & 15u;  // no load: the descriptor carries the kinds
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
      cell = isn ? ft.zero_bin : pos;  // MissingType::None: NaN is compared as 0.0
    } else {
      const bool miss = (kind >= QV_MISS_RIGHT) ? (isn || isz) : isn;
      const uint32_t mval = (kind & 1) ? 0u : (uint32_t)QS_RIGHT;  // *_LEFT kinds are odd
      cell = miss ? mval : pos;
    }
    emit(v, cell);
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

}  // namespace mrk
