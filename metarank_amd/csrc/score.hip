// Forest scoring kernel for gfx950 (CDNA4): the device side of Booster.predictMat
// (reference call site: ml/rank/LambdaMARTRanker.scala:348; arithmetic: lib_lightgbm /
// libxgboost).
//
// Mapping: one lane = one matrix row (item).  A workgroup of TILE lanes stages its TILE x D
// feature tile in LDS *feature-major* (rows[f * TILE + lane]); with TILE a multiple of 32 the
// bank of a lane's read depends only on the lane, so the data-dependent feature gather of the
// tree walk is LDS-conflict-free whatever features the lanes ask for.  The forest streams
// through LDS in chunks of whole trees (a tree's node array is <= 256 B for a 16-leaf LightGBM
// tree, i.e. one LDS bank row: distinct nodes never collide, equal nodes broadcast).  Every lane
// walks U trees at once (independent dependency chains hide the ~2 x 64-cycle LDS latency per
// node visit), then adds the U leaves in tree order, so the per-row sum is the same sequence of
// f64 (LightGBM) / f32 (XGBoost) additions as the reference libraries perform: bit-identical.
//
// No MFMA: this is compare/index work.  Bound: LDS issue + VALU, not HBM.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <string>

#include "runtime.hpp"

namespace mrk {

namespace {

#ifndef MRK_SCORE_U
#define MRK_SCORE_U 4
#endif
constexpr int U = MRK_SCORE_U;  // trees walked concurrently per lane

struct alignas(16) Node16 {
  uint32_t w0, w1;  // thr (f64) | thr (f32) + feat/flags
  uint32_t w2, w3;
};

__device__ __forceinline__ bool in_bitset(const uint32_t *__restrict__ bits, uint32_t begin,
                                          uint32_t words, int c) {
  uint32_t w = (uint32_t)c >> 5;
  if (w >= words) return false;
  return (bits[begin + w] >> (c & 31)) & 1u;
}

// LightGBM Tree::NumericalDecision / CategoricalDecision on a value that already went through the
// dense-row zero flush (|v| <= 1e-35 -> 0.0).  Returns true for "go left".
__device__ __forceinline__ bool decide64(uint32_t w0, uint32_t w1, uint32_t w2, double v,
                                         const uint32_t *__restrict__ cat_bits) {
  const uint32_t flags = (w2 >> 16) & 0xffu;
  const bool nan_left = (w2 >> 24) & 1u;
  const bool isn = v != v;
  if (__builtin_expect(flags & NF_CATEGORICAL, 0)) {
    if (isn) return false;
    int iv = (int)v;  // v_cvt_i32_f64 saturates; out-of-range -> not in the bitset / negative
    if (iv < 0) return false;
    return in_bitset(cat_bits, w0, w1, iv);
  }
  const double thr = __hiloint2double((int)w1, (int)w0);
  bool left = v <= thr;
  if ((flags & NF_MISS_ZERO) && v == 0.0) left = (flags & NF_DEFAULT_LEFT) != 0;
  return isn ? nan_left : left;
}

// XGBoost RegTree::GetNext with a float feature value; NaN is "missing".
__device__ __forceinline__ bool decide32(uint32_t w0, uint32_t w1, uint32_t w3, float v,
                                         const uint32_t *__restrict__ cat_bits) {
  const uint32_t flags = (w1 >> 16) & 0xffu;
  const bool def_left = (flags & NF_DEFAULT_LEFT) != 0;
  if (v != v) return def_left;
  if (__builtin_expect(flags & NF_CATEGORICAL, 0)) {
    // common::Decision: invalid category (negative or >= 2^24) or beyond the bitset -> left;
    // member of the set -> right.
    if (v < 0.f || v >= 16777216.f) return true;
    int c = (int)v;
    return !in_bitset(cat_bits, w0, w3, c);
  }
  return v < __uint_as_float(w0);
}

template <bool F64>
struct RowT;
template <>
struct RowT<true> { using type = double; };
template <>
struct RowT<false> { using type = float; };

// dense-row preprocessing applied once per cell when the tile is staged
template <bool F64>
__device__ __forceinline__ typename RowT<F64>::type prep(double x, int *flag, int bit = 1) {
  if constexpr (F64) {
    // LightGBM RowFunctionFromDenseMatric keeps a cell only if |x| > kZeroThreshold (1e-35f) or NaN;
    // everything else reads back as 0.0 from the prediction buffer.
    const double kZero = (double)1e-35f;
    return (fabs(x) > kZero || x != x) ? x : 0.0;
  } else {
    // ltrlib narrows Double -> Float before DMatrix (round-to-nearest-even, overflow -> inf);
    // XGBoost rejects +-inf when `missing` is NaN ("Input data contains `inf`").
    float f = (float)x;
    if (__builtin_isinf(f)) atomicOr(flag, bit);
    return f;
  }
}

template <bool F64, int TILE, bool ROWS_LDS>
__global__ void __launch_bounds__(TILE)
score_kernel(const uint8_t *__restrict__ image, const TreeRef *__restrict__ trees,
             const ChunkRef *__restrict__ chunks, int n_chunks,
             const uint32_t *__restrict__ cat_bits, const double *__restrict__ X, int rows, int cols,
             double base, double *__restrict__ out, int *__restrict__ flag,
             const uint32_t *__restrict__ row_req, uint32_t chunk_cap, uint32_t ref_cap) {
  using row_t = typename RowT<F64>::type;
  extern __shared__ __align__(16) uint8_t smem[];
  uint8_t *s_chunk = smem;                                   // chunk_cap bytes
  TreeRef *s_refs = (TreeRef *)(smem + chunk_cap);           // ref_cap bytes
  row_t *s_rows = (row_t *)(smem + chunk_cap + ref_cap);     // cols * TILE

  const int tid = threadIdx.x;
  const long long row0 = (long long)blockIdx.x * TILE;
  const long long row = row0 + tid;
  const int tile_rows = (int)min((long long)TILE, (long long)rows - row0);

  if constexpr (ROWS_LDS) {
    // coalesced read of the contiguous tile_rows x cols block, transposed into feature-major LDS
    const long long total = (long long)tile_rows * cols;
    const double *src = X + row0 * cols;
    for (long long e = tid; e < total; e += TILE) {
      int r = (int)(e / cols);
      int c = (int)(e - (long long)r * cols);
      // row_req != null: per-request status words (bit 32 = inf met), else one flag word (bit 1)
      s_rows[c * TILE + r] = row_req ? prep<F64>(src[e], flag + row_req[row0 + r], 32) : prep<F64>(src[e], flag);
    }
    // tail lanes keep walking (uniform control flow); give them zeros
    for (int e = tile_rows + tid; e < TILE; e += TILE)
      for (int c = 0; c < cols; ++c) s_rows[c * TILE + e] = (row_t)0;
  } else if constexpr (!F64) {
    // rows read from global memory on demand (matrices too wide for the LDS tile): XGBoost's DMatrix rejects a row for
    // an inf in ANY column, visited by the walk or not - the same inputs fail on every scorer path
    if (row < rows)
      for (int c = 0; c < cols; ++c) (void)(row_req ? prep<false>(X[row * cols + c], flag + row_req[row], 32) : prep<false>(X[row * cols + c], flag));
  }

  double acc64 = 0.0;
  float acc32 = (float)base;  // XGBoost: predictions start at the base margin, f32

  for (int ci = 0; ci < n_chunks; ++ci) {
    const ChunkRef ch = chunks[ci];
    __syncthreads();  // previous chunk fully consumed (and tile staged, first iteration)
    {
      const uint4 *src = (const uint4 *)(image + ch.byte_off);
      uint4 *dst = (uint4 *)s_chunk;
      const int n16 = (int)(ch.byte_len >> 4);
      for (int i = tid; i < n16; i += TILE) dst[i] = src[i];
      const uint32_t *rsrc = (const uint32_t *)(trees + ch.first_tree);
      uint32_t *rdst = (uint32_t *)s_refs;
      const int nw = (int)ch.n_trees * 3;
      for (int i = tid; i < nw; i += TILE) rdst[i] = rsrc[i];
    }
    __syncthreads();

    const int nt = (int)ch.n_trees;
    for (int t0 = 0; t0 < nt; t0 += U) {
      int node[U];
      uint32_t nbase[U], lbase[U];
      int maxd = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = min(t0 + u, nt - 1);
        const TreeRef tr = s_refs[t];
        nbase[u] = tr.node_off;
        lbase[u] = tr.leaf_off;
        const bool live = (t0 + u) < nt && tr.n_nodes != 0;
        node[u] = live ? 0 : -1;  // -1 == ~0: single-leaf tree (or padding slot, never added)
        maxd = max(maxd, live ? (int)tr.depth : 0);
      }
      for (int d = 0; d < maxd; ++d) {
        // Phase-structured so that the U dependency chains overlap: all node reads are issued
        // back to back, then all feature gathers, then the (branch-free) numerical decisions.
        // Categorical nodes are rare and fixed up afterwards under one wave-level branch.
        Node16 nd[U];
        bool walking[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          walking[u] = node[u] >= 0;  // finished lanes re-read node 0; keep their gather in range
          nd[u] = *(const Node16 *)(s_chunk + nbase[u] + (uint32_t)max(node[u], 0) * 16u);
        }
        bool left[U];
        uint32_t any_flags = 0;
        if constexpr (F64) {
          double v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t feat = walking[u] ? (nd[u].w2 & 0xffffu) : 0u;
            if constexpr (ROWS_LDS) v[u] = s_rows[feat * TILE + tid];
            else v[u] = (walking[u] && row < rows) ? (row_req ? prep<true>(X[row * cols + feat], flag + row_req[row], 32)
                                                              : prep<true>(X[row * cols + feat], flag)) : 0.0;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t flags = (nd[u].w2 >> 16) & 0xffu;
            any_flags |= walking[u] ? flags : 0u;
            const double thr = __hiloint2double((int)nd[u].w1, (int)nd[u].w0);
            bool l = v[u] <= thr;
            if ((flags & NF_MISS_ZERO) && v[u] == 0.0) l = (flags & NF_DEFAULT_LEFT) != 0;
            left[u] = (v[u] != v[u]) ? (((nd[u].w2 >> 24) & 1u) != 0) : l;
          }
          if (__builtin_expect((any_flags & NF_CATEGORICAL) != 0, 0)) {
#pragma unroll
            for (int u = 0; u < U; ++u)
              if (walking[u] && (((nd[u].w2 >> 16) & NF_CATEGORICAL) != 0))
                left[u] = decide64(nd[u].w0, nd[u].w1, nd[u].w2, v[u], cat_bits);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int next = left[u] ? (int)(short)(nd[u].w3 & 0xffffu) : (int)(short)(nd[u].w3 >> 16);
            node[u] = walking[u] ? next : node[u];
          }
        } else {
          float v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t feat = walking[u] ? (nd[u].w1 & 0xffffu) : 0u;
            if constexpr (ROWS_LDS) v[u] = s_rows[feat * TILE + tid];
            else v[u] = (walking[u] && row < rows) ? (row_req ? prep<false>(X[row * cols + feat], flag + row_req[row], 32)
                                                              : prep<false>(X[row * cols + feat], flag)) : 0.f;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t flags = (nd[u].w1 >> 16) & 0xffu;
            any_flags |= walking[u] ? flags : 0u;
            const bool l = v[u] < __uint_as_float(nd[u].w0);
            left[u] = (v[u] != v[u]) ? ((flags & NF_DEFAULT_LEFT) != 0) : l;
          }
          if (__builtin_expect((any_flags & NF_CATEGORICAL) != 0, 0)) {
#pragma unroll
            for (int u = 0; u < U; ++u)
              if (walking[u] && (((nd[u].w1 >> 16) & NF_CATEGORICAL) != 0))
                left[u] = decide32(nd[u].w0, nd[u].w1, nd[u].w3, v[u], cat_bits);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int next = left[u] ? (int)(short)(nd[u].w2 & 0xffffu) : (int)(short)(nd[u].w2 >> 16);
            node[u] = walking[u] ? next : node[u];
          }
        }
      }
      // leaves are added strictly in tree order
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (t0 + u < nt) {
          const int leaf = ~node[u];
          if constexpr (F64) acc64 += *(const double *)(s_chunk + lbase[u] + (uint32_t)leaf * 8u);
          else acc32 += *(const float *)(s_chunk + lbase[u] + (uint32_t)leaf * 4u);
        }
      }
    }
  }
  if (row < rows) out[row] = F64 ? acc64 : (double)acc32;
}

template <bool F64, int TILE, bool ROWS_LDS>
void launch_t(mrk_ctx *ctx, mrk_model *m, const double *d_x, int rows, int cols, double *d_out,
              int *d_flag, const uint32_t *d_row_req, uint32_t chunk_cap, uint32_t ref_cap, size_t smem) {
  auto kern = score_kernel<F64, TILE, ROWS_LDS>;
  lds_optin(ctx, (const void *)kern);
  const int grid = (rows + TILE - 1) / TILE;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(TILE), smem, ctx->launch, m->d_image.as<uint8_t>(),
                     m->d_trees.as<TreeRef>(), m->d_chunks.as<ChunkRef>(), (int)m->packed.chunks.size(),
                     m->d_cat.as<uint32_t>(), d_x, rows, cols, m->forest.base_score, d_out, d_flag, d_row_req,
                     chunk_cap, ref_cap);
  MRK_HIP(hipGetLastError());
}

template <bool F64>
void launch_b(mrk_ctx *ctx, mrk_model *m, const double *d_x, int rows, int cols, double *d_out, int *d_flag,
              const uint32_t *d_row_req) {
  const uint32_t chunk_cap = (m->packed.max_chunk_bytes + 15u) & ~15u;
  const uint32_t ref_cap = ((uint32_t)m->packed.max_chunk_trees * (uint32_t)sizeof(TreeRef) + 15u) & ~15u;
  const size_t esz = F64 ? 8 : 4;
  const size_t budget = 150 * 1024;  // of the 160 KiB LDS per CU
  auto fits = [&](int tile) { return chunk_cap + ref_cap + (size_t)cols * tile * esz <= budget; };
  auto smem = [&](int tile) { return (size_t)chunk_cap + ref_cap + (size_t)cols * tile * esz; };
  // Prefer the tile that still leaves room for two workgroups per CU (more waves to hide LDS
  // latency); fall back to smaller tiles, then to reading rows from global memory.
  const size_t half = 78 * 1024;
  // (512-row tiles: the forest chunk is shared by twice the rows, so two such workgroups hold 16 wavefronts per CU where
  // three 256-row ones hold 12 - MRK_WALK_TILE=256 keeps the smaller tile for A/B runs)
  if (rows > 256 && smem(512) <= half && switches().walk_tile != 256) launch_t<F64, 512, true>(ctx, m, d_x, rows, cols, d_out, d_flag, d_row_req, chunk_cap, ref_cap, smem(512));
  else if (rows > 128 && smem(256) <= half) launch_t<F64, 256, true>(ctx, m, d_x, rows, cols, d_out, d_flag, d_row_req, chunk_cap, ref_cap, smem(256));
  else if (rows > 64 && smem(128) <= half) launch_t<F64, 128, true>(ctx, m, d_x, rows, cols, d_out, d_flag, d_row_req, chunk_cap, ref_cap, smem(128));
  else if (smem(64) <= half) launch_t<F64, 64, true>(ctx, m, d_x, rows, cols, d_out, d_flag, d_row_req, chunk_cap, ref_cap, smem(64));
  else if (fits(256) && rows > 128) launch_t<F64, 256, true>(ctx, m, d_x, rows, cols, d_out, d_flag, d_row_req, chunk_cap, ref_cap, smem(256));
  else if (fits(128) && rows > 64) launch_t<F64, 128, true>(ctx, m, d_x, rows, cols, d_out, d_flag, d_row_req, chunk_cap, ref_cap, smem(128));
  else if (fits(64)) launch_t<F64, 64, true>(ctx, m, d_x, rows, cols, d_out, d_flag, d_row_req, chunk_cap, ref_cap, smem(64));
  else launch_t<F64, 256, false>(ctx, m, d_x, rows, cols, d_out, d_flag, d_row_req, chunk_cap, ref_cap, (size_t)chunk_cap + ref_cap);
}

}  // namespace

uint32_t score_chunk_budget() { return 24 * 1024; }

void launch_score_batch(mrk_ctx *ctx, mrk_model *m, const double *d_x, int rows, int cols, double *d_out,
                        int *d_status, const uint32_t *d_row_req) {
  if (rows <= 0) return;
  // MRK_SCORER=walk forces the tree-walk kernel (A/B measurements, parity of both kernels); read per call so
  // that a test can flip it
  const bool walk_only = switches().scorer_walk;
  if (!walk_only && launch_score_qs(ctx, m, d_x, rows, cols, d_out, d_status, d_row_req)) return;
  ScopedKernelTimer timer(ctx, "score");
  if (m->forest.backend == Backend::LightGBM) launch_b<true>(ctx, m, d_x, rows, cols, d_out, d_status, d_row_req);
  else launch_b<false>(ctx, m, d_x, rows, cols, d_out, d_status, d_row_req);
}

void launch_score(mrk_ctx *ctx, mrk_model *m, const double *d_x, int rows, int cols, double *d_out,
                  int *d_flag) {
  launch_score_batch(ctx, m, d_x, rows, cols, d_out, d_flag, nullptr);
}

}  // namespace mrk
