// Bit-vector forest scorer for gfx950 (CDNA4): Booster.predictMat for forests of small trees
// (<= 16 leaves, LightGBM's numLeaves default in Metarank, config/BoosterConfig.scala:19-28).
// Reference call site: ml/rank/LambdaMARTRanker.scala:348; arithmetic: lib_lightgbm / libxgboost
// (SURVEY.md §8c).  Format and the exactness argument: forest.hpp ("qs" image).
//
// Two kernels:
//   qs_bin_kernel    f64 matrix -> u16 cells, one column ("view") per (column, missing rule) pair,
//                    laid out [wave tile][view][row] so that the scorer's tile is one contiguous slab.
//   qs_score_kernel  one wavefront owns R*64 rows; its slab sits in LDS.  For every tree ALL node
//                    tests are evaluated: node constants come from scalar loads (they are uniform
//                    across the wave), the cells of R rows per lane come from one conflict-free LDS
//                    read, and two rows share each VALU op (v_pk_sub_i16 / v_pk_ashrrev_i16 /
//                    v_and_or_b32 on 16-bit halves).  No branches, no dependent loads: the tree walk's
//                    latency chain is gone.  Leaves are added in tree order in f64 (LightGBM) / f32
//                    (XGBoost), the same additions the libraries make: scores are bit-identical.
//
// No MFMA: compare/index work.  Bound: VALU issue (about 1.75 ops per row-node), then LDS reads.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "launch_shape.hpp"
#include "qs_device.hpp"
#include "runtime.hpp"

namespace mrk {

QsDev qs_device_view(const mrk_model *m);

namespace {

constexpr int QS_WAVES = 4;  // wavefronts per workgroup (they share the staged leaf values)

// ---------------------------------------------------------------------------------- binning

template <bool F64>
__global__ void __launch_bounds__(256)
qs_bin_kernel(const double *__restrict__ X, int rows, int cols, QsDev q, uint16_t *__restrict__ cells, int tile_rows,
              long long padded_rows, int *__restrict__ flag, const uint32_t *__restrict__ row_req) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  if (row >= padded_rows) return;
  const long long tile = row / tile_rows;
  const int r = (int)(row - tile * tile_rows);
  uint16_t *dst = cells + (size_t)tile * q.n_views * tile_rows + r;
  const bool valid = row < rows;
  const double *xr = X + (valid ? row : 0) * cols;
  const int nf = q.n_feats < cols ? q.n_feats : cols;
  if constexpr (!F64)  // columns beyond the last one the forest knows are part of the DMatrix row too
    for (int c = nf; c < cols; ++c) {
      bool ok;
      (void)qs_prep<F64>(xr[c], ok);
      if (!ok && valid) {
        if (row_req) atomicOr(flag + row_req[row], 32);
        else atomicOr(flag, 1);
      }
    }
  for (int c = 0; c < nf; ++c) {
    const QsFeature ft = q.feats[c];  // uniform: scalar loads
    bool ok = true;
    if (ft.view_begin == ft.view_end) {
      if constexpr (!F64) (void)qs_prep<F64>(xr[c], ok);  // XGBoost rejects an inf in any column, split on or not
    } else {
      ok = qs_bin_column<F64>(xr[c], ft, q.views, q.thr, [&](uint32_t v, uint32_t cell) {
        dst[(size_t)v * tile_rows] = valid ? (uint16_t)cell : (uint16_t)0;
      });
    }
    if (!ok && valid) {
      if (row_req) atomicOr(flag + row_req[row], 32);
      else atomicOr(flag, 1);
    }
  }
}

// ---------------------------------------------------------------------------------- scoring

// Generic kernel: QS_WAVES wavefronts per workgroup share LDS-staged leaf chunks; R = 2, 4 or 8 rows per lane.
template <bool F64, int R>
__global__ void __launch_bounds__(QS_WAVES * 64)
qs_score_kernel(const uint32_t *__restrict__ nodes, const uint8_t *__restrict__ leaves,
                const QsCatNode *__restrict__ cat_nodes, const uint32_t *__restrict__ cat_bits,
                const uint16_t *__restrict__ cells, int n_trees, int V, int rows, long long n_tiles, double base,
                double *__restrict__ out, int chunk_trees) {
  constexpr int LS = F64 ? 8 : 4;               // bytes per leaf
  constexpr int TREE_LEAF_BYTES = QS_LEAVES * LS;
  constexpr int VIEW_BYTES = R * 64 * 2;        // one view of one wave tile
  constexpr int VIEW_SHIFT = R == 8 ? 2 : (R == 4 ? 1 : 0);  // the node word carries view * 256
  static_assert(R == 2 || R == 4 || R == 8, "rows per lane");
  extern __shared__ __align__(16) uint8_t smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t leaf_cap = (uint32_t)chunk_trees * TREE_LEAF_BYTES;
  const long long tile = (long long)blockIdx.x * QS_WAVES + wave;
  const uint32_t tile_off = leaf_cap + (uint32_t)wave * (uint32_t)V * VIEW_BYTES;

  // stage this wave's slab: V views x R*64 cells, contiguous in HBM
  if (tile < n_tiles) {
    const uint4 *src = (const uint4 *)(cells + (size_t)tile * V * (R * 64));
    uint4 *dst = (uint4 *)(smem + tile_off);
    const int n16 = V * (VIEW_BYTES / 16);
    for (int i = lane; i < n16; i += 64) dst[i] = src[i];
  }

  const uint32_t lane_off = tile_off + (uint32_t)lane * (R * 2);
  double acc64[R];
  float acc32[R];
#pragma unroll
  for (int j = 0; j < R; ++j) { acc64[j] = 0.0; acc32[j] = (float)base; }

  for (int t0 = 0; t0 < n_trees; t0 += chunk_trees) {
    const int nt = min(chunk_trees, n_trees - t0);
    __syncthreads();  // previous leaf chunk consumed
    {
      const uint4 *src = (const uint4 *)(leaves + (size_t)t0 * TREE_LEAF_BYTES);
      uint4 *dst = (uint4 *)smem;
      const int n16 = nt * (TREE_LEAF_BYTES / 16);
      for (int i = tid; i < n16; i += QS_WAVES * 64) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t *nd = nodes + (size_t)t0 * QS_TREE_WORDS;
    for (int t = 0; t < nt; ++t, nd += QS_TREE_WORDS) {
      uint32_t accn[R / 2];
#pragma unroll
      for (int p = 0; p < R / 2; ++p) accn[p] = 0;
#pragma unroll
      for (int s = 0; s < QS_SLOTS - 1; ++s) {
        const uint32_t kk = nd[s];             // k | k << 16        (scalar)
        const uint32_t mv = nd[QS_SLOTS + s];  // m | view << 24     (scalar)
        const uint32_t mm = (mv & 0xffffu) * 0x10001u;
        const uint32_t addr = ((mv >> 16) << VIEW_SHIFT) + lane_off;  // v_lshl_add_u32
        uint32_t c[R / 2];
        if constexpr (R == 2) {
          c[0] = *(const uint32_t *)(smem + addr);
        } else if constexpr (R == 4) {
          const uint2 q = *(const uint2 *)(smem + addr);
          c[0] = q.x; c[1] = q.y;
        } else {
          const uint4 q = *(const uint4 *)(smem + addr);
          c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w;
        }
#pragma unroll
        for (int p = 0; p < R / 2; ++p) {
          const short2v d = __builtin_bit_cast(short2v, kk) - __builtin_bit_cast(short2v, c[p]);  // < 0 <=> cell > k
          const short2v sg = d >> 15;                                                            // 0xFFFF where the test is false
          accn[p] |= __builtin_bit_cast(uint32_t, sg) & mm;
        }
      }
      const uint32_t catw = nd[QS_SLOTS - 1];
      if (catw >> 24) {
        const QsCatNode *cn = cat_nodes + (catw & 0xffffffu);
        for (uint32_t j = 0; j < (catw >> 24); ++j) {
          const QsCatNode c = cn[j];  // scalar
          const uint32_t addr = ((c.view_dl & 0xffffu) << (8 + VIEW_SHIFT)) + lane_off;
#pragma unroll
          for (int p = 0; p < R / 2; ++p) accn[p] |= qs_cat_pair<F64>(c, *(const uint32_t *)(smem + addr + 4 * p), cat_bits);
        }
      }
      // exit leaf = lowest position not removed; position nl-1 is in no left subtree, so a zero bit exists
      const uint32_t lbase = (uint32_t)t * TREE_LEAF_BYTES;
#pragma unroll
      for (int p = 0; p < R / 2; ++p) {
        const uint32_t inv = ~accn[p];
        const uint32_t l0 = (uint32_t)__builtin_ctz(inv);
        const uint32_t l1 = (uint32_t)__builtin_ctz(inv >> 16);
        if constexpr (F64) {
          acc64[2 * p] += *(const double *)(smem + lbase + l0 * 8u);
          acc64[2 * p + 1] += *(const double *)(smem + lbase + l1 * 8u);
        } else {
          acc32[2 * p] += *(const float *)(smem + lbase + l0 * 4u);
          acc32[2 * p + 1] += *(const float *)(smem + lbase + l1 * 4u);
        }
      }
    }
  }
  if (tile < n_tiles) {
    const long long row0 = tile * (R * 64) + (long long)lane * R;
#pragma unroll
    for (int j = 0; j < R; ++j)
      if (row0 + j < rows) out[row0 + j] = F64 ? acc64[j] : (double)acc32[j];
  }
}

// One-wavefront workgroups, two rows per lane (the default).  Scalar (SMEM) loads and LDS reads share one
// counter (lgkmcnt) and SMEM returns out of order, so any wait for an LDS result while a scalar load is in
// flight drains the scalar load too.  The loop keeps the two apart:
//   A  issue the 15 cell reads of tree t: `ds_read_addtid_b32` (LDS address = M0 + 4 * lane), M0 set by the
//      scalar unit, so a read costs no VALU op and no address VGPR
//   B  wait for them (the one explicit s_waitcnt: the reads are inline asm, invisible to the compiler's
//      wait-count pass - cdna_hip_programming.md §5.7)
//   C  issue the scalar loads of tree t+1's node constants          <- nothing else outstanding
//   D  pure VALU: 15 x {v_pk_sub_i16, v_pk_ashrrev_i16, v_and_or_b32}, the exit leaf, the leaf adds of
//      tree t-2; then the global loads of tree t's two leaf values (L1/L2 resident, on vmcnt)
// so the scalar-load latency of the next tree hides behind D and behind the other waves of the SIMD.
// No LDS leaf chunks and no barriers: a workgroup's LDS footprint is its V x 256 B slab, so occupancy is
// 160 KB / (V x 256 B) waves per CU at wave granularity.
template <bool F64>
__global__ void __launch_bounds__(64)
qs_score_wave_kernel(const uint32_t *__restrict__ nodes, const uint8_t *__restrict__ leaves,
                     const QsCatNode *__restrict__ cat_nodes, const uint32_t *__restrict__ cat_bits,
                     const uint16_t *__restrict__ cells, int n_trees, int V, int rows, double base,
                     double *__restrict__ out) {
  constexpr int R = 2;
  constexpr int LS = F64 ? 8 : 4;
  constexpr int TREE_LEAF_BYTES = QS_LEAVES * LS;
  extern __shared__ __align__(16) uint8_t smem[];
  const int lane = threadIdx.x;
  const long long tile = blockIdx.x;
  {
    const uint4 *src = (const uint4 *)(cells + (size_t)tile * V * (R * 64));
    uint4 *dst = (uint4 *)smem;
    const int n16 = V * 16;
    for (int i = lane; i < n16; i += 64) dst[i] = src[i];
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // slab written (one wave: its LDS ops complete in order)
  double acc64[R] = {0.0, 0.0};
  float acc32[R] = {(float)base, (float)base};

  // one pipeline step for tree t: `cur` holds its node constants, `nxt` receives tree t+1's; `lv*` holds
  // the leaf values of tree t-2 (added here) and then receives those of tree t
  auto step = [&](QsNodeRegs &cur, QsNodeRegs &nxt, const uint32_t *__restrict__ nd_next, const uint8_t *__restrict__ lv_tree,
                  double (&lv64)[R], float (&lv32)[R]) {
    // A: cell reads, M0 = view * 256.  A DS add-TID instruction needs one wait state after an M0 write;
    // the slot is filled with the s_pack_ll_b32_b16 that replicates the node's mask for phase D.
    uint32_t c[QS_SLOTS - 1], mm[QS_SLOTS - 1];
#pragma unroll
    for (int s = 0; s < QS_SLOTS - 1; ++s)
      asm volatile("s_lshr_b32 m0, %2, 16\n\ts_pack_ll_b32_b16 %1, %2, %2\n\tds_read_addtid_b32 %0"
                   : "=v"(c[s]), "=s"(mm[s]) : "s"(cur.mv[s]) : "memory");
    // B
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    // C
    qs_load_nodes(nxt, nd_next);
    __builtin_amdgcn_sched_barrier(0);
    // D
    // (all subtracts, then all shifts, then the mask accumulate on two chains: no VALU op waits on its predecessor)
#pragma unroll
    for (int s = 0; s < QS_SLOTS - 1; ++s)
      c[s] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, cur.kk[s]) - __builtin_bit_cast(short2v, c[s]));  // < 0 <=> cell > k
#pragma unroll
    for (int s = 0; s < QS_SLOTS - 1; ++s)
      c[s] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, c[s]) >> 15);  // 0xFFFF where the test is false
    uint32_t acc_a = 0, acc_b = 0;
#pragma unroll
    for (int s = 0; s < QS_SLOTS - 1; s += 2) {
      asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc_a) : "v"(c[s]), "s"(mm[s]));
      if (s + 1 < QS_SLOTS - 1) asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc_b) : "v"(c[s + 1]), "s"(mm[s + 1]));
    }
    uint32_t accn = acc_a | acc_b;
#pragma unroll
    for (int j = 0; j < R; ++j) {  // tree t-2's leaves (-0.0 for the first two trees: x + -0.0 == x, bit for bit)
      if constexpr (F64) acc64[j] += lv64[j];
      else acc32[j] += lv32[j];
    }
    const uint32_t catw = cur.kk[QS_SLOTS - 1];
    if (catw >> 24) {
      const QsCatNode *cn = cat_nodes + (catw & 0xffffffu);
      for (uint32_t j = 0; j < (catw >> 24); ++j) {
        const QsCatNode cnode = cn[j];
        accn |= qs_cat_pair<F64>(cnode, *(const uint32_t *)(smem + ((cnode.view_dl & 0xffffu) << 8) + lane * 4), cat_bits);
      }
    }
    // exit leaf = lowest position not removed; position nl-1 is in no left subtree, so a zero bit exists
    const uint32_t inv = ~accn;
    const uint32_t o0 = (uint32_t)__builtin_ctz(inv) * LS;
    const uint32_t o1 = (uint32_t)__builtin_ctz(inv >> 16) * LS;
    if constexpr (F64) {
      lv64[0] = *(const double *)(lv_tree + o0);
      lv64[1] = *(const double *)(lv_tree + o1);
    } else {
      lv32[0] = *(const float *)(lv_tree + o0);
      lv32[1] = *(const float *)(lv_tree + o1);
    }
  };

  QsNodeRegs ra, rb;
  qs_load_nodes(ra, nodes);
  double la64[R] = {-0.0, -0.0}, lb64[R] = {-0.0, -0.0};
  float la32[R] = {-0.f, -0.f}, lb32[R] = {-0.f, -0.f};
  const uint32_t *nd = nodes;      // the array ends with one all-zero tree: the last prefetch stays in bounds
  const uint8_t *lt = leaves;
  for (int t = 0; t < n_trees; t += 2) {
    step(ra, rb, nd + QS_TREE_WORDS, lt, la64, la32);
    if (t + 1 < n_trees) step(rb, ra, nd + 2 * QS_TREE_WORDS, lt + TREE_LEAF_BYTES, lb64, lb32);
    nd += 2 * QS_TREE_WORDS;
    lt += 2 * TREE_LEAF_BYTES;
  }
  // drain: the last two trees' leaves, in tree order
  const bool odd = n_trees & 1;
  if (odd) {
#pragma unroll
    for (int j = 0; j < R; ++j) { if constexpr (F64) acc64[j] += lb64[j]; else acc32[j] += lb32[j]; }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) { if constexpr (F64) acc64[j] += la64[j]; else acc32[j] += la32[j]; }
  if (!odd) {
#pragma unroll
    for (int j = 0; j < R; ++j) { if constexpr (F64) acc64[j] += lb64[j]; else acc32[j] += lb32[j]; }
  }
  const long long row0 = tile * (R * 64) + (long long)lane * R;
#pragma unroll
  for (int j = 0; j < R; ++j)
    if (row0 + j < rows) out[row0 + j] = F64 ? acc64[j] : (double)acc32[j];
}

// Few tiles (a single request, a 100 000-candidate request on 256 CUs): NW (2 / 4 / 8 / 16) wavefronts share ONE tile and split
// its trees, so the forest is walked NW times faster.  The scores stay bit-identical because only the exit
// LEAF INDICES are computed in parallel: per chunk of 8 * NW trees every wavefront writes the indices of its
// trees (tree w, w + NW, ...) to LDS, then the rows' owners add the chunk's leaf values in tree order - the same
// additions as the one-wavefront kernel, in the same order.
template <bool F64, int NW>
__global__ void __launch_bounds__(NW * 64)
qs_score_split_kernel(const uint32_t *__restrict__ nodes, const uint8_t *__restrict__ leaves,
                      const QsCatNode *__restrict__ cat_nodes, const uint32_t *__restrict__ cat_bits,
                      const uint16_t *__restrict__ cells, int n_trees, int V, int rows, double base,
                      double *__restrict__ out) {
  constexpr int LS = F64 ? 8 : 4;
  constexpr int TREE_LEAF_BYTES = QS_LEAVES * LS;
  constexpr int CH = 8 * NW;  // trees per chunk
  extern __shared__ __align__(16) uint8_t smem[];
  // [slab: V x 256 B][leaf values of the chunk: CH x 16 leaves][exit leaf index of (tree, row): CH x 128 B]
  uint8_t *s_leaf = smem + (size_t)V * 256;
  uint8_t *s_idx = s_leaf + CH * TREE_LEAF_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wavefront-uniform: node constants by scalar loads
  const long long tile = blockIdx.x;
  {
    const uint4 *src = (const uint4 *)(cells + (size_t)tile * V * QS_TILE_ROWS);
    uint4 *dst = (uint4 *)smem;
    for (int i = tid; i < V * 16; i += NW * 64) dst[i] = src[i];
  }
  double acc64 = 0.0;
  float acc32 = (float)base;
  for (int c0 = 0; c0 < n_trees; c0 += CH) {
    const int nt = min(CH, n_trees - c0);
    __syncthreads();  // slab staged / previous chunk consumed
    {
      const uint4 *src = (const uint4 *)(leaves + (size_t)c0 * TREE_LEAF_BYTES);
      uint4 *dst = (uint4 *)s_leaf;
      for (int i = tid; i < nt * (TREE_LEAF_BYTES / 16); i += NW * 64) dst[i] = src[i];
    }
    for (int tt = wave; tt < nt; tt += NW) {  // this wavefront's trees of the chunk
      const uint32_t *nd = nodes + (size_t)(c0 + tt) * QS_TREE_WORDS;
      QsNodeRegs r;
      qs_load_nodes(r, nd);
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
      const uint32_t catw = r.kk[QS_SLOTS - 1];
      if (catw >> 24) {
        const QsCatNode *cn = cat_nodes + (catw & 0xffffffu);
        for (uint32_t j = 0; j < (catw >> 24); ++j) {
          const QsCatNode cnode = cn[j];
          accn |= qs_cat_pair<F64>(cnode, *(const uint32_t *)(smem + ((cnode.view_dl & 0xffffu) << 8) + lane * 4), cat_bits);
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
  const long long row = tile * QS_TILE_ROWS + tid;
  if (tid < QS_TILE_ROWS && row < rows) out[row] = F64 ? acc64 : (double)acc32;
}

// (Measured and removed, round 6 - commit 4ad2903 has the code: a PIPELINED form of the split kernel - 7 evaluating wavefronts + 1
// adding wavefront per tile, the exit leaves of chunk k double-buffered against the adds of chunk k - 1, one barrier per chunk,
// nobody idle.  Same box: 0.226 vs 0.223 ms at 384 000 rows, 1.966 vs 1.908 ms at 4 M rows, bench 1 041 vs 1 142 M items/s
// (profiles/r06_b_score_pipe_ab.txt).  The idle adders were never the loss: at 4 M rows the split kernel runs a tree step in 300
// cycles per SIMD where the bare step - no scalar loads, no leaves - sustains 265 at 8 wavefronts per SIMD and its 58 VALU
// instructions alone take 244 (tools/native/issue_bench.hip, profiles/r06_a_issue_bench.txt); what a 3 000-tile launch loses
// on top is the DRAIN: the last workgroups run on an emptying chip, where one wavefront per SIMD needs 723 cycles per tree.)
template <bool F64>
void launch_wave(mrk_ctx *ctx, mrk_model *m, const uint16_t *d_cells, int rows, double *d_out) {
  const PackedForestQS &q = m->qs;
  const int V = (int)q.views.size();
  const long long n_tiles = ((long long)rows + QS_TILE_ROWS - 1) / QS_TILE_ROWS;
  // wavefronts per tile: by residency and fill (launch_shape.hpp has the rule and the measurements behind it)
  const int split_env = switches().qs_split;
  const int nw = split_env >= 0 ? split_env : scorer_waves_per_tile(n_tiles, V, F64, ctx->n_cus, QS_LEAVES, QS_TILE_ROWS);
  if (nw == 2 || nw == 4 || nw == 8 || nw == 16) {
    const size_t lds = (size_t)V * 256 + (size_t)8 * nw * (QS_LEAVES * (F64 ? 8 : 4) + QS_TILE_ROWS);
    if (lds <= 160 * 1024) {
      ScopedKernelTimer timer(ctx, "score");
#define MRK_SPLIT(NW_)                                                                                                      \
      {                                                                                                                       \
        auto sk = qs_score_split_kernel<F64, NW_>;                                                                            \
        lds_optin(ctx, (const void *)sk);                                                                                     \
        hipLaunchKernelGGL(sk, dim3((unsigned)n_tiles), dim3(NW_ * 64), lds, ctx->launch, m->d_qs_nodes.as<uint32_t>(),       \
                           m->d_qs_leaves.as<uint8_t>(), m->d_qs_catnodes.as<QsCatNode>(), m->d_qs_cat.as<uint32_t>(), d_cells, \
                           q.n_trees, V, rows, m->forest.base_score, d_out);                                                  \
      }
      if (nw == 16) MRK_SPLIT(16) else if (nw == 8) MRK_SPLIT(8) else if (nw == 4) MRK_SPLIT(4) else MRK_SPLIT(2)
#undef MRK_SPLIT
      MRK_HIP(hipGetLastError());
      return;
    }
  }
  auto wk = qs_score_wave_kernel<F64>;
  lds_optin(ctx, (const void *)wk);
  // LDS request = the slab, nothing more: measured on 3000 tiles, any padding that lowers the number of
  // resident wavefronts per CU below ceil(tiles / CUs) costs a second round (0.24 -> 0.34 ms at 13 KB);
  // the allocation granularity makes a 12.25 KB request the largest that still fits 12 per CU.
  const size_t lds = (size_t)V * 256;
  ScopedKernelTimer timer(ctx, "score");
  hipLaunchKernelGGL(wk, dim3((unsigned)n_tiles), dim3(64), lds, ctx->launch, m->d_qs_nodes.as<uint32_t>(),
                     m->d_qs_leaves.as<uint8_t>(), m->d_qs_catnodes.as<QsCatNode>(), m->d_qs_cat.as<uint32_t>(), d_cells,
                     q.n_trees, V, rows, m->forest.base_score, d_out);
  MRK_HIP(hipGetLastError());
}

template <bool F64, int R>
void launch_generic(mrk_ctx *ctx, mrk_model *m, const uint16_t *d_cells, int rows, double *d_out, int chunk_trees, size_t smem) {
  const PackedForestQS &q = m->qs;
  const int V = (int)q.views.size();
  const long long n_tiles = ((long long)rows + R * 64 - 1) / (R * 64);
  auto kern = qs_score_kernel<F64, R>;
  lds_optin(ctx, (const void *)kern);
  ScopedKernelTimer timer(ctx, "score");
  const int grid = (int)((n_tiles + QS_WAVES - 1) / QS_WAVES);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(QS_WAVES * 64), smem, ctx->launch, m->d_qs_nodes.as<uint32_t>(),
                     m->d_qs_leaves.as<uint8_t>(), m->d_qs_catnodes.as<QsCatNode>(), m->d_qs_cat.as<uint32_t>(), d_cells,
                     q.n_trees, V, rows, n_tiles, m->forest.base_score, d_out, chunk_trees);
  MRK_HIP(hipGetLastError());
}

template <bool F64>
void launch_bin(mrk_ctx *ctx, mrk_model *m, const double *d_x, int rows, int cols, int tile_rows, int *d_flag,
                const uint32_t *d_row_req) {
  const long long n_tiles = ((long long)rows + tile_rows - 1) / tile_rows;
  const long long padded = n_tiles * tile_rows;
  ctx->d_cells.reserve((size_t)padded * m->qs.views.size() * 2);
  ScopedKernelTimer timer(ctx, "bin");
  const int grid = (int)((padded + 255) / 256);
  hipLaunchKernelGGL(qs_bin_kernel<F64>, dim3(grid), dim3(256), 0, ctx->launch, d_x, rows, cols, qs_device_view(m),
                     ctx->d_cells.as<uint16_t>(), tile_rows, padded, d_flag, d_row_req);
  MRK_HIP(hipGetLastError());
}

template <bool F64>
bool launch_qs_b(mrk_ctx *ctx, mrk_model *m, const double *d_x, int rows, int cols, double *d_out, int *d_flag,
                 const uint32_t *d_row_req) {
  // MRK_QS_KERNEL=0 selects the multi-wave generic kernel with MRK_QS_R rows per lane (A/B measurements)
  const int variant = switches().qs_kernel;
  if (variant != 0) {
    launch_bin<F64>(ctx, m, d_x, rows, cols, QS_TILE_ROWS, d_flag, d_row_req);
    launch_wave<F64>(ctx, m, ctx->d_cells.as<uint16_t>(), rows, d_out);
    return true;
  }
  const size_t V = m->qs.views.size();
  const int leaf_bytes = QS_LEAVES * (F64 ? 8 : 4);
  const int chunk_trees = std::min(m->qs.n_trees, 8 * 1024 / leaf_bytes);
  auto smem = [&](int r) { return (size_t)chunk_trees * leaf_bytes + (size_t)QS_WAVES * V * r * 128; };
  const int forced = switches().qs_r;
  const int r = (forced == 4 || forced == 8) ? forced : 2;
  if (smem(r) > 160 * 1024) return false;
  launch_bin<F64>(ctx, m, d_x, rows, cols, r * 64, d_flag, d_row_req);
  if (r == 8) launch_generic<F64, 8>(ctx, m, ctx->d_cells.as<uint16_t>(), rows, d_out, chunk_trees, smem(8));
  else if (r == 4) launch_generic<F64, 4>(ctx, m, ctx->d_cells.as<uint16_t>(), rows, d_out, chunk_trees, smem(4));
  else launch_generic<F64, 2>(ctx, m, ctx->d_cells.as<uint16_t>(), rows, d_out, chunk_trees, smem(2));
  return true;
}

}  // namespace

QsDev qs_device_view(const mrk_model *m) {
  QsDev q;
  q.feats = m->d_qs_feats.as<QsFeature>();
  q.views = m->d_qs_views.as<QsView>();
  q.thr = m->d_qs_thr.as<double>();
  q.thr_rt = q.thr + m->qs.thr.size() + 2 * QS_STAGE_CHUNK;   // (capi.cpp upload_model: tables, slack, compact tables)
  q.n_feats = (int32_t)m->qs.feats.size();
  q.n_views = (int32_t)m->qs.views.size();
  static_assert(QS_LDS_THR == 256u, "qs_stage_cap (forest.cpp) stages tables of up to QS_LDS_THR entries");
  q.thr_cap = switches().thr_stage ? qs_stage_cap(m->qs) : 0u;
  q.rt_doubles = switches().thr_stage && switches().jit_sig && m->qs_sig.ok ? m->qs_sig.rt_total : 0u;
  return q;
}

QsForestDev qs_forest_view(const mrk_model *m) {
  QsForestDev f;
  f.nodes = m->d_qs_nodes.as<uint32_t>();
  f.leaves = m->d_qs_leaves.as<uint8_t>();
  f.cat_nodes = m->d_qs_catnodes.as<QsCatNode>();
  f.cat_bits = m->d_qs_cat.as<uint32_t>();
  f.n_trees = m->qs.n_trees;
  f.n_views = (int32_t)m->qs.views.size();
  f.base = m->forest.base_score;
  return f;
}

// Returns false when the model has no bit-vector image (trees with more than 16 leaves, ...): the
// caller falls back to the tree-walk kernel (score.hip).  Both produce identical bits.
bool launch_score_qs(mrk_ctx *ctx, mrk_model *m, const double *d_x, int rows, int cols, double *d_out, int *d_flag,
                     const uint32_t *d_row_req) {
  if (!m->qs.ok) return false;
  if (m->forest.backend == Backend::LightGBM) return launch_qs_b<true>(ctx, m, d_x, rows, cols, d_out, d_flag, d_row_req);
  return launch_qs_b<false>(ctx, m, d_x, rows, cols, d_out, d_flag, d_row_req);
}

// Scores rows whose binned cells ([tile of 128 rows][view][row] u16) were written by the assembly kernels.
void launch_score_qs_cells(mrk_ctx *ctx, mrk_model *m, const uint16_t *d_cells, int rows, double *d_out) {
  if (m->forest.backend == Backend::LightGBM) launch_wave<true>(ctx, m, d_cells, rows, d_out);
  else launch_wave<false>(ctx, m, d_cells, rows, d_out);
}

}  // namespace mrk
