// Feature assembly + ordering kernels for gfx950 (CDNA4): the device side of
//   Ranker.makeQuery  (FeatureValueLoader.fromStateBackend + ItemValue.fromState + ClickthroughQuery)
//   Ranker.rerank's   sortBy(-_.score)
// Reference: ml/Ranker.scala:27-83,97-106; feature/*.scala (cited per op in rank.hpp).
//
// Two phases per request, then the scorer (score_qs.hip / score.hip), then the ordering:
//   pre-pass   the cross-item reductions (session-profile token histograms for interacted_with, token
//              histogram / median over the first `top` present items for diversity) -> small
//              open-addressing tables
//   assemble   one lane per candidate item: gathers the item's record (one contiguous record per
//              item, see store.hpp) and evaluates every op of the model program; each value goes to a
//              *sink*: the row-major f64 matrix (ClickthroughQuery's layout: explain / parity / models
//              without a bit-vector image) or, for the hot path, straight into the scorer's binned
//              u16 tile (qs_device.hpp) - the f64 matrix is then never written or read
// Small requests run both phases in ONE workgroup with the tables in LDS (rank_fused_*_kernel); large
// requests (more candidates than one workgroup should loop over, or tables that do not fit LDS) use
// prepass_kernel -> tables in HBM -> assemble_kernel across many workgroups.
// The device code of both phases lives in rank_device.hpp: the kernels here interpret any model program from
// device memory; jit.cpp compiles the same code once more per model with the program as constants (the hot path).
//   sort_kernel      one workgroup per request: stable descending order, java.lang.Double.compare
// Compiled with -ffp-contract=off: the JVM never fuses a*b+c, and parity is bit-exact.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <utility>

#include "rank_device.hpp"
#include "sort_device.hpp"
#include "runtime.hpp"

namespace mrk {

namespace {

// the pre-pass alone, interpreting the program (rank_device.hpp prepass_body)
__global__ void __launch_bounds__(PREP_THREADS)
prepass_kernel(StoreDev st, ProgramDev prog, BatchDev b, uint32_t lds_entries) { prepass_body(st, prog, b, lds_entries); }

__global__ void __launch_bounds__(ASM_THREADS)
assemble_kernel(StoreDev st, ProgramDev prog, BatchDev b) {
  const int gi0 = b.item_lo + blockIdx.x * ASM_THREADS + threadIdx.x;
  const bool active = gi0 < b.item_hi;
  if (!__any(active)) return;                      // whole wavefront past the end
  const int gi = active ? gi0 : b.item_hi - 1;     // lanes without an item ride along on a missing record
  const int r = (int)b.item_req[gi];
  const ReqDev rq = b.reqs[r];
  MatrixSink sink{b.matrix + (size_t)gi * prog.dim, active};
  assemble_item(st, prog, b, gi, r, rq, b.arena, 0u, &b.prep_out[(size_t)r * prog.n_prep], sink);
}

template <bool F64>
__global__ void __launch_bounds__(ASM_THREADS)
assemble_cells_kernel(StoreDev st, ProgramDev prog, BatchDev b, QsDev q, uint16_t *cells, uint32_t lds_entries) {
  assemble_cells_body<F64>(st, prog, b, q, cells, lds_entries);
}

// (the interpreting kernels always carry the op split: they are the fall-back, not the hot path)
__global__ void __launch_bounds__(512)
rank_fused_matrix_kernel(StoreDev st, ProgramDev prog, BatchDev b, uint32_t tab_entries, int vals_cap, int mode) {
  rank_fused_matrix_body<true>(st, prog, b, tab_entries, vals_cap, mode);
}

template <bool F64>
__global__ void __launch_bounds__(512)
rank_fused_cells_kernel(StoreDev st, ProgramDev prog, BatchDev b, uint32_t tab_entries, int vals_cap, QsDev q, uint16_t *cells, int mode) {
  rank_fused_cells_body<F64, true>(st, prog, b, tab_entries, vals_cap, q, cells, mode);
}

// pre-pass + assembly + forest + ordering of a small request in ONE launch (rank_device.hpp rank_one_body)
template <bool F64>
__global__ void __launch_bounds__(512)
rank_one_kernel(StoreDev st, ProgramDev prog, BatchDev b, uint32_t tab_entries, int vals_cap, QsDev q, QsForestDev f, int mode, OneOut out) {
  rank_one_body<F64>(st, prog, b, tab_entries, vals_cap, q, f, mode, out);
}

// full batches of small requests: assembly, forest and ordering in the request's workgroup (rank_device.hpp rank_fused_score_body)
template <bool F64>
__global__ void __launch_bounds__(256)
rank_fused_score_kernel(StoreDev st, ProgramDev prog, BatchDev b, uint32_t tab_entries, int vals_cap, QsDev q, QsForestDev f, uint16_t *cells) {
  rank_fused_score_body<F64>(st, prog, b, tab_entries, vals_cap, q, f, cells);
}

// the persistent form: one workgroup serves the requests published in its slot (rank_device.hpp rank_serve_body)
template <bool F64>
__global__ void __launch_bounds__(512)
rank_serve_kernel(StoreDev st, ProgramDev prog, QsDev q, QsForestDev f, ServeGangDev gang) {
  rank_serve_body<F64>(st, prog, q, f, gang);
}

__global__ void override_kernel(BatchDev b, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.n_overrides) return;
  const Override o = b.overrides[i];
  if ((int)o.item < b.item_lo || (int)o.item >= b.item_hi) return;
  b.matrix[(size_t)o.item * dim + o.col] = o.value;
}

// the same for the binned tile: re-bin the overriding value into every view of its column
template <bool F64>
__global__ void override_cells_kernel(BatchDev b, QsDev q, uint16_t *cells) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.n_overrides) return;
  const Override o = b.overrides[i];
  if ((int)o.item < b.item_lo || (int)o.item >= b.item_hi) return;
  // lanes of this kernel hold different columns: the per-lane search in global memory, not the staged one
  if ((int)o.col >= q.n_feats) return;
  const QsFeature ft = q.feats[o.col];
  uint16_t *d = cells + (size_t)(o.item / QS_TILE_ROWS) * q.n_views * QS_TILE_ROWS + (o.item % QS_TILE_ROWS);
  const bool ok = qs_bin_column<F64>(o.value, ft, q.views, q.thr, [d](uint32_t view, uint32_t cell) { d[view * QS_TILE_ROWS] = (uint16_t)cell; });
  if (!ok) atomicOr(&b.status[b.item_req[o.item]], 32);
}

// ---------------------------------------------------------------- ordering
// sortBy(-_.score): ascending java.lang.Double.compare on the negated score, stable (keys: sort_device.hpp).
constexpr int SORT_THREADS = 256;
constexpr int SORT_COUNT_MAX = 256;  // requests up to this size are ordered by counting (sort_kernel)

// Dynamic LDS: the keys (8 B) - and for requests of more than SORT_COUNT_MAX candidates the indices (4 B) - of the batch's
// LARGEST request rounded up to a power of two (`cap` entries), not of the largest request the kernel can order: with the
// 48 KB a 4 096-candidate request needs, the workgroups of a batch of 100-candidate requests could not be placed next to
// another batch's assembly workgroups (which fill a CU's LDS) and the kernel took as long as that assembly did
// (gpurun_out r03_r: 24 us alone, 250 - 290 us in the serving loop - on the critical path of every batch's download).
__global__ void __launch_bounds__(SORT_THREADS)
sort_kernel(BatchDev b, int cap) {
  extern __shared__ __align__(16) unsigned long long sort_smem[];
  unsigned long long *s_key = sort_smem;
  int *s_idx = (int *)(sort_smem + cap);
  const int r = blockIdx.x;
  const int tid = threadIdx.x;
  const ReqDev rq = b.reqs[r];
  const int n = rq.n_items;
  if (n <= 0 || n > SORT_MAX_ITEMS) return;  // larger requests: the multi-workgroup sort below
  if (n <= SORT_COUNT_MAX) {
    // A request of the usual size (100 candidates): every lane counts the elements that precede its own - the keys are
    // read from LDS at one address per instruction (a broadcast), there is one barrier instead of 28 bitonic steps, and
    // (key, index) pairs are distinct, so the count IS the position in the stable order.
    for (int i = tid; i < n; i += SORT_THREADS) s_key[i] = sort_key(b.scores[rq.item_begin + i]);
    __syncthreads();
    for (int i = tid; i < n; i += SORT_THREADS) {
      const unsigned long long mine = s_key[i];
      int before = 0;
      for (int j = 0; j < n; ++j) {
        const unsigned long long kj = s_key[j];
        before += (kj < mine || (kj == mine && j < i)) ? 1 : 0;
      }
      b.order[rq.item_begin + before] = i;
    }
    return;
  }
  int p2 = 1;
  while (p2 < n) p2 <<= 1;
  for (int i = tid; i < p2; i += SORT_THREADS) {
    s_key[i] = i < n ? sort_key(b.scores[rq.item_begin + i]) : ~0ull;
    s_idx[i] = i < n ? i : 0x7fffffff;
  }
  __syncthreads();
  // bitonic sort on (key, index): the index makes every pair distinct => equals the stable order.  One compare-exchange
  // per lane and step: pair p of a step with stride j is (i, i | j) with i = p's bits spread around bit j.
  for (int k = 2; k <= p2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int p = tid; p < p2 / 2; p += SORT_THREADS) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), ixj = i | j;
        const unsigned long long ka = s_key[i], kb = s_key[ixj];
        const int ia = s_idx[i], ib = s_idx[ixj];
        const bool gt = ka > kb || (ka == kb && ia > ib);
        const bool up = (i & k) == 0;
        if (gt == up) { s_key[i] = kb; s_key[ixj] = ka; s_idx[i] = ib; s_idx[ixj] = ia; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < n; i += SORT_THREADS) b.order[rq.item_begin + i] = s_idx[i];
}

// ---------------------------------------------------------------- Normalize.scale over one matrix column
// ml/onnx/Normalize.scala:13-45, applied by the bi-/cross-encoder field_match features to the request's raw values
// (FieldMatchBiencoderFeature.scala:107, FieldMatchCrossEncoderFeature.scala:111).  One workgroup per request.
//   linear    (v - min) / (max - min) with min / max over the non-NaN values (scala Ordering.Double = Double.compare:
//             -0.0 < +0.0); nothing changes when every value is NaN; max == min gives NaN (0 / 0) like the reference
//   position  sortedIndex / size after a stable ascending sort of ALL values (NaN last); NaN values stay NaN
__global__ void __launch_bounds__(SORT_THREADS)
normalize_kernel(BatchDev b, int dim, int col, int mode) {
  __shared__ unsigned long long s_key[SORT_MAX_ITEMS];
  __shared__ int s_idx[SORT_MAX_ITEMS];
  const int r = blockIdx.x;
  const int tid = threadIdx.x;
  const ReqDev rq = b.reqs[r];
  const int n = rq.n_items;
  if (n <= 0) return;
  double *colp = b.matrix + (size_t)rq.item_begin * dim + col;
  if (mode == NORM_MINMAX) {
    if (tid == 0) { s_key[0] = ~0ull; s_key[1] = 0ull; }
    __syncthreads();
    unsigned long long lo = ~0ull, hi = 0ull;
    for (int i = tid; i < n; i += SORT_THREADS) {
      const double v = colp[(size_t)i * dim];
      if (v != v) continue;
      const unsigned long long k = asc_key(v);
      lo = k < lo ? k : lo;
      hi = k > hi ? k : hi;
    }
    if (lo != ~0ull) { atomicMin(&s_key[0], lo); atomicMax(&s_key[1], hi); }  // (no double maps to key ~0: that is a NaN pattern)
    __syncthreads();
    if (s_key[0] == ~0ull) return;  // scores.minOption == None: values unchanged
    const double mn = asc_value(s_key[0]), mx = asc_value(s_key[1]);
    const double span = __dsub_rn(mx, mn);
    for (int i = tid; i < n; i += SORT_THREADS) {
      double *p = colp + (size_t)i * dim;
      *p = __ddiv_rn(__dsub_rn(*p, mn), span);
    }
    return;
  }
  if (n > SORT_MAX_ITEMS) return;  // ordered by the multi-workgroup sort, applied by norm_position_apply_kernel (launch_normalize_big)
  int p2 = 1;
  while (p2 < n) p2 <<= 1;
  for (int i = tid; i < p2; i += SORT_THREADS) {
    s_key[i] = i < n ? asc_key(colp[(size_t)i * dim]) : ~0ull;
    s_idx[i] = i < n ? i : 0x7fffffff;
  }
  __syncthreads();
  for (int k = 2; k <= p2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < p2; i += SORT_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long ka = s_key[i], kb = s_key[ixj];
          const int ia = s_idx[i], ib = s_idx[ixj];
          const bool gt = ka > kb || (ka == kb && ia > ib);
          const bool up = (i & k) == 0;
          if (gt == up) { s_key[i] = kb; s_key[ixj] = ka; s_idx[i] = ib; s_idx[ixj] = ia; }
        }
      }
      __syncthreads();
    }
  }
  const unsigned long long nan_key = asc_key(d_nan());
  const double size = (double)n;
  for (int s = tid; s < n; s += SORT_THREADS)
    if (s_key[s] != nan_key) colp[(size_t)s_idx[s] * dim] = __ddiv_rn((double)s, size);
}

}  // namespace

void launch_big_sort(mrk_ctx *ctx, hipStream_t stream, const SortSrc &src, int n, int *out_order, void *scratch);  // bigsort.hip

// jit_fn: mrk_jit_prepass of this program (jit.cpp), or nullptr = the kernel that interprets the program
void launch_prepass(mrk_ctx *ctx, const StoreDev &st, const ProgramDev &prog, const BatchDev &b, uint32_t max_req_entries, void *jit_fn) {
  if (b.n_req <= 0 || prog.n_prep <= 0) return;
  // tables in LDS when the largest request's fit next to the kernel's 33 KB of static scratch (MRK_PREPASS_LDS=0: HBM arena)
  constexpr uint32_t LDS_TABLE_BUDGET = 96 * 1024;
  uint32_t lds_entries = switches().prepass_lds && (uint64_t)max_req_entries * 8 <= LDS_TABLE_BUDGET ? max_req_entries : 0;
  lds_optin(ctx, (const void *)prepass_kernel, LDS_TABLE_BUDGET);  // static + dynamic <= 160 KB
  ScopedKernelTimer timer(ctx, "prepass");
  // (a module function keeps the default 64 KB limit on static + dynamic LDS: 33 KB are static here)
  if (jit_fn && (size_t)lds_entries * 8 <= 30 * 1024) {
    StoreDev a_st = st;
    BatchDev a_b = b;
    void *args[] = {&a_st, &a_b, &lds_entries};
    MRK_HIP(hipModuleLaunchKernel((hipFunction_t)jit_fn, (unsigned)b.n_req, 1, 1, PREP_THREADS, 1, 1, (unsigned)((size_t)lds_entries * 8), ctx->launch, args, nullptr));
  } else {
    hipLaunchKernelGGL(prepass_kernel, dim3(b.n_req), dim3(PREP_THREADS), (size_t)lds_entries * 8, ctx->launch, st, prog, b, lds_entries);
  }
  MRK_HIP(hipGetLastError());
}

void launch_assemble(mrk_ctx *ctx, const StoreDev &st, const ProgramDev &prog, const BatchDev &b) {
  if (b.item_hi <= b.item_lo) return;
  {
    ScopedKernelTimer timer(ctx, "assemble");
    const int grid = (b.item_hi - b.item_lo + ASM_THREADS - 1) / ASM_THREADS;
    hipLaunchKernelGGL(assemble_kernel, dim3(grid), dim3(ASM_THREADS), 0, ctx->launch, st, prog, b);
    MRK_HIP(hipGetLastError());
  }
  if (b.n_overrides > 0) {
    ScopedKernelTimer timer(ctx, "override");
    hipLaunchKernelGGL(override_kernel, dim3((b.n_overrides + 255) / 256), dim3(256), 0, ctx->launch, b, prog.dim);
    MRK_HIP(hipGetLastError());
  }
}

static void launch_override_cells(mrk_ctx *ctx, const BatchDev &b, const QsDev &q, uint16_t *cells, bool f64) {
  if (b.n_overrides <= 0) return;
  ScopedKernelTimer timer(ctx, "override");
  const dim3 grid((b.n_overrides + 255) / 256);
  if (f64) hipLaunchKernelGGL(override_cells_kernel<true>, grid, dim3(256), 0, ctx->launch, b, q, cells);
  else hipLaunchKernelGGL(override_cells_kernel<false>, grid, dim3(256), 0, ctx->launch, b, q, cells);
  MRK_HIP(hipGetLastError());
}

// assembly straight into the scorer's binned tile (tables from a previous launch_prepass); jit_fn: the specialised
// mrk_jit_assemble_cells of this program, or nullptr = the kernel that interprets the program
// max_req_entries: the largest request's table entries (the pre-pass launch's figure): when they fit ITEMS_LDS_TABLE_BYTES,
// single-request workgroups copy their request's tables into LDS (rank_device.hpp assemble_cells_body); MRK_ITEMS_LDS=0: never
void launch_assemble_cells(mrk_ctx *ctx, const StoreDev &st, const ProgramDev &prog, const BatchDev &b, const QsDev &q,
                           uint16_t *cells, bool f64, void *jit_fn, uint32_t max_req_entries, void *jit_rt_fn, uint32_t thr_total) {
  if (b.item_hi <= b.item_lo) return;
  if (jit_rt_fn) {
    // the resident-table form (rank_device.hpp assemble_cells_rt_body): persistent workgroups, all threshold tables in LDS
    ScopedKernelTimer timer(ctx, "assemble");
    const int n_items = b.item_hi - b.item_lo;
    constexpr uint32_t RT_LDS_TABLE_BYTES = 24 * 1024;   // (as the staging form: a request's hash tables up to 24 KB next to the thresholds)
    uint32_t lds_entries = switches().items_lds && prog.n_prep > 0 && (uint64_t)max_req_entries * 8 <= RT_LDS_TABLE_BYTES ? max_req_entries : 0u;
    const size_t lds = (size_t)thr_total * 8 + (size_t)lds_entries * 8;
    // lanes per workgroup: 512 (8 wavefronts share one copy of the tables) when that still gives every CU two workgroups' worth
    // of blocks, else 256 (a 100 000-candidate request is 391 blocks of 256)
    int threads = (long long)n_items >= 2ll * 512 * ctx->n_cus ? 512 : 256;
    if (switches().items_rt_threads == 256 || switches().items_rt_threads == 512) threads = switches().items_rt_threads;
    const int n_blocks = (n_items + threads - 1) / threads;
    // resident workgroups per CU: by LDS (160 KB) and by wavefronts (16 at 128 VGPRs)
    const int per_cu = std::max(1, std::min((int)((160 * 1024) / std::max<size_t>(lds, 1)), 16 / (threads / 64)));
    const unsigned grid = (unsigned)std::min(n_blocks, per_cu * ctx->n_cus);
    StoreDev a_st = st;
    BatchDev a_b = b;
    QsDev a_q = q;
    void *args[] = {&a_st, &a_b, &a_q, &cells, &lds_entries};
    MRK_HIP(hipModuleLaunchKernel((hipFunction_t)jit_rt_fn, grid, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, ctx->launch, args, nullptr));
    launch_override_cells(ctx, b, q, cells, f64);
    return;
  }
  {
    ScopedKernelTimer timer(ctx, "assemble");
    const dim3 grid((b.item_hi - b.item_lo + ASM_THREADS - 1) / ASM_THREADS);
    // 16 KB of static threshold staging + <= 24 KB of tables: four workgroups per CU, what 128 VGPRs allow anyway
    constexpr uint32_t ITEMS_LDS_TABLE_BYTES = 24 * 1024;
    uint32_t lds_entries = switches().items_lds && prog.n_prep > 0 && (uint64_t)max_req_entries * 8 <= ITEMS_LDS_TABLE_BYTES ? max_req_entries : 0u;
    const size_t lds = (size_t)lds_entries * 8;
    if (jit_fn) {  // the kernel specialised for this model's program (jit.cpp)
      StoreDev a_st = st;
      BatchDev a_b = b;
      QsDev a_q = q;
      void *args[] = {&a_st, &a_b, &a_q, &cells, &lds_entries};
      MRK_HIP(hipModuleLaunchKernel((hipFunction_t)jit_fn, grid.x, 1, 1, ASM_THREADS, 1, 1, (unsigned)lds, ctx->launch, args, nullptr));
    } else if (f64) hipLaunchKernelGGL(assemble_cells_kernel<true>, grid, dim3(ASM_THREADS), lds, ctx->launch, st, prog, b, q, cells, lds_entries);
    else hipLaunchKernelGGL(assemble_cells_kernel<false>, grid, dim3(ASM_THREADS), lds, ctx->launch, st, prog, b, q, cells, lds_entries);
    MRK_HIP(hipGetLastError());
  }
  launch_override_cells(ctx, b, q, cells, f64);
}

// thr_cap: doubles per threshold staging buffer (QsDev::thr_cap; QS_LDS_THR = the largest any model needs)
// rt_bytes: the forest's compact threshold tables when the kernel that runs may keep them resident (rank_device.hpp FUSED_RT_MAX_BYTES;
// the region then holds whichever the kernel's sink uses - staging buffers or the resident copy - so it is sized for both)
size_t fused_lds_bytes(uint32_t tab_entries, int vals_cap, int threads, uint32_t thr_cap, size_t rt_bytes, bool split) {
  const size_t staging = (size_t)((threads + 63) / 64) * 2 * thr_cap * 8;  // two buffers per wavefront
  return (size_t)tab_entries * 8 + (size_t)vals_cap * 8 + FUSED_MAX_PREP * sizeof(PrepOut) + PREP_INTS * sizeof(int) +
         16 + std::max(staging, rt_bytes <= (split ? FUSED_RT_MAX_BYTES_SPLIT : FUSED_RT_MAX_BYTES) ? rt_bytes : 0);  // (16-B aligned)
}
size_t fused_rt_max_bytes() { return FUSED_RT_MAX_BYTES_SPLIT; }
int fused_max_prep() { return FUSED_MAX_PREP; }

// pre-pass + assembly of every request in one launch (one workgroup per request, tables in LDS).
// cells == nullptr: write the f64 matrix; else write the binned tile.
// jit_fn: the hipFunction_t of the kernel specialised for this program (jit.cpp), or nullptr = the generic kernel.
void launch_rank_fused(mrk_ctx *ctx, const StoreDev &st, const ProgramDev &prog, const BatchDev &b, uint32_t tab_entries,
                       int vals_cap, int threads, int op_split, int slices, const QsDev *q, uint16_t *cells, bool f64, void *jit_fn, size_t rt_bytes) {
  if (b.n_req <= 0) return;
  int mode = (op_split > 1 ? op_split : 1) | ((slices > 1 ? slices : 1) << 8);  // rank_device.hpp rank_fused_body
  const unsigned grid = (unsigned)b.n_req * (unsigned)(slices > 1 ? slices : 1);
  size_t lds = fused_lds_bytes(tab_entries, vals_cap, threads, q ? q->thr_cap : 0u, q && cells ? rt_bytes : 0, op_split > 1 || slices > 1);
  if (switches().fused_lds_min > 0) lds = std::max(lds, (size_t)switches().fused_lds_min);  // experiments: cap the kernel's residency (co-residency with the scorer)
  {
    ScopedKernelTimer timer(ctx, "assemble");
    if (!jit_fn) lds_optin(ctx, !cells ? (const void *)rank_fused_matrix_kernel : f64 ? (const void *)rank_fused_cells_kernel<true> : (const void *)rank_fused_cells_kernel<false>);
    if (!cells && jit_fn) {  // mrk_jit_rank_matrix
      StoreDev a_st = st;
      BatchDev a_b = b;
      int a_vals = vals_cap;
      void *args[] = {&a_st, &a_b, &tab_entries, &a_vals, &mode};   // mrk_jit_rank_matrix (no op-split form: the caller asked for none)
      MRK_HIP(hipModuleLaunchKernel((hipFunction_t)jit_fn, grid, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, ctx->launch, args, nullptr));
    } else if (cells && jit_fn) {
      StoreDev a_st = st;
      BatchDev a_b = b;
      QsDev a_q = *q;
      int a_vals = vals_cap;
      void *args[] = {&a_st, &a_b, &tab_entries, &a_vals, &a_q, &cells, &mode};
      MRK_HIP(hipModuleLaunchKernel((hipFunction_t)jit_fn, grid, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, ctx->launch, args, nullptr));
    } else if (!cells) hipLaunchKernelGGL(rank_fused_matrix_kernel, dim3(grid), dim3(threads), lds, ctx->launch, st, prog, b, tab_entries, vals_cap, mode);
    else if (f64) hipLaunchKernelGGL(rank_fused_cells_kernel<true>, dim3(grid), dim3(threads), lds, ctx->launch, st, prog, b, tab_entries, vals_cap, *q, cells, mode);
    else hipLaunchKernelGGL(rank_fused_cells_kernel<false>, dim3(grid), dim3(threads), lds, ctx->launch, st, prog, b, tab_entries, vals_cap, *q, cells, mode);
    MRK_HIP(hipGetLastError());
  }
  if (!cells) {
    if (b.n_overrides > 0) {
      ScopedKernelTimer timer(ctx, "override");
      hipLaunchKernelGGL(override_kernel, dim3((b.n_overrides + 255) / 256), dim3(256), 0, ctx->launch, b, prog.dim);
      MRK_HIP(hipGetLastError());
    }
  } else {
    launch_override_cells(ctx, b, *q, cells, f64);
  }
}

// norm: position of a request of more than SORT_MAX_ITEMS candidates: `order` = the candidates in stable ascending order
// of the column (bigsort.hip); the one at sorted place s gets s / size, NaN values stay NaN (Normalize.scala:25-40)
__global__ void __launch_bounds__(SORT_THREADS)
norm_position_apply_kernel(const int *__restrict__ order, int n, double *colp, int dim) {
  const int s = blockIdx.x * SORT_THREADS + threadIdx.x;
  if (s >= n) return;
  double *p = colp + (size_t)order[s] * dim;
  const double v = *p;
  if (v == v) *p = __ddiv_rn((double)s, (double)n);
}

void launch_normalize_big(mrk_ctx *ctx, const BatchDev &b, int dim, int col, int item_begin, int n, int *order, void *scratch) {
  ScopedKernelTimer timer(ctx, "normalize");
  double *colp = b.matrix + (size_t)item_begin * dim + col;
  const SortSrc src{colp, nullptr, dim, 0};
  launch_big_sort(ctx, ctx->launch, src, n, order, scratch);
  hipLaunchKernelGGL(norm_position_apply_kernel, dim3((n + SORT_THREADS - 1) / SORT_THREADS), dim3(SORT_THREADS), 0, ctx->launch, order, n, colp, dim);
  MRK_HIP(hipGetLastError());
}

// LDS of the one-launch kernel: [slab][status word][max(assembly regions, scoring regions)]
size_t rank_one_lds_bytes(uint32_t tab_entries, int vals_cap, int threads, uint32_t thr_cap, int n_views, bool f64, size_t rt_bytes) {
  const size_t nw = (size_t)threads / 64;
  const size_t scoring = 8 * nw * (QS_LEAVES * (f64 ? 8 : 4) + QS_TILE_ROWS) + QS_TILE_ROWS * 8;
  return (size_t)n_views * QS_TILE_ROWS * 2 + 16 + std::max(fused_lds_bytes(tab_entries, vals_cap, threads, thr_cap, rt_bytes, true), scoring);
}

// One workgroup per request (requests of <= QS_TILE_ROWS candidates, `threads` = item lanes x op split, <= 512);
// jit_fn: the specialised mrk_jit_rank_one of this program, or nullptr = the interpreting kernel.
void launch_rank_one(mrk_ctx *ctx, const StoreDev &st, const ProgramDev &prog, const BatchDev &b, uint32_t tab_entries, int vals_cap,
                     int threads, int op_split, const QsDev &q, const QsForestDev &f, const OneOut &out, bool f64, void *jit_fn) {
  if (b.n_req <= 0) return;
  int mode = op_split > 1 ? op_split : 1;
  const size_t lds = rank_one_lds_bytes(tab_entries, vals_cap, threads, q.thr_cap, f.n_views, f64, (size_t)q.rt_doubles * 8);
  ScopedKernelTimer timer(ctx, "rank_one");
  if (jit_fn) {
    StoreDev a_st = st;
    BatchDev a_b = b;
    QsDev a_q = q;
    QsForestDev a_f = f;
    OneOut a_out = out;
    int a_vals = vals_cap;
    void *args[] = {&a_st, &a_b, &tab_entries, &a_vals, &a_q, &a_f, &mode, &a_out};
    MRK_HIP(hipModuleLaunchKernel((hipFunction_t)jit_fn, (unsigned)b.n_req, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, ctx->launch, args, nullptr));
  } else {
    lds_optin(ctx, f64 ? (const void *)rank_one_kernel<true> : (const void *)rank_one_kernel<false>);
    if (f64) hipLaunchKernelGGL(rank_one_kernel<true>, dim3(b.n_req), dim3(threads), lds, ctx->launch, st, prog, b, tab_entries, vals_cap, q, f, mode, out);
    else hipLaunchKernelGGL(rank_one_kernel<false>, dim3(b.n_req), dim3(threads), lds, ctx->launch, st, prog, b, tab_entries, vals_cap, q, f, mode, out);
    MRK_HIP(hipGetLastError());
  }
}

size_t rank_fused_score_lds_bytes(uint32_t tab_entries, int vals_cap, int threads, uint32_t thr_cap, int n_views, bool f64, size_t rt_bytes) {
  const size_t nw = (size_t)threads / 64;
  const size_t scoring = (size_t)n_views * QS_TILE_ROWS * 2 + 16 + 8 * nw * (QS_LEAVES * (f64 ? 8 : 4) + QS_TILE_ROWS) + QS_TILE_ROWS * 8;
  return std::max(fused_lds_bytes(tab_entries, vals_cap, threads, thr_cap, rt_bytes, false), scoring);
}

// One workgroup per request of a full batch (requests of <= QS_TILE_ROWS candidates, 128 or 64 lanes); `cells`: one tile per request.
void launch_rank_fused_score(mrk_ctx *ctx, const StoreDev &st, const ProgramDev &prog, const BatchDev &b, uint32_t tab_entries, int vals_cap,
                             int threads, const QsDev &q, const QsForestDev &f, uint16_t *cells, bool f64, void *jit_fn) {
  if (b.n_req <= 0) return;
  const size_t lds = rank_fused_score_lds_bytes(tab_entries, vals_cap, threads, q.thr_cap, f.n_views, f64, (size_t)q.rt_doubles * 8);
  ScopedKernelTimer timer(ctx, "rank_fused");
  if (jit_fn) {
    StoreDev a_st = st;
    BatchDev a_b = b;
    QsDev a_q = q;
    QsForestDev a_f = f;
    int a_vals = vals_cap;
    void *args[] = {&a_st, &a_b, &tab_entries, &a_vals, &a_q, &a_f, &cells};
    MRK_HIP(hipModuleLaunchKernel((hipFunction_t)jit_fn, (unsigned)b.n_req, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, ctx->launch, args, nullptr));
    return;
  }
  lds_optin(ctx, f64 ? (const void *)rank_fused_score_kernel<true> : (const void *)rank_fused_score_kernel<false>);
  if (f64) hipLaunchKernelGGL(rank_fused_score_kernel<true>, dim3(b.n_req), dim3(threads), lds, ctx->launch, st, prog, b, tab_entries, vals_cap, q, f, cells);
  else hipLaunchKernelGGL(rank_fused_score_kernel<false>, dim3(b.n_req), dim3(threads), lds, ctx->launch, st, prog, b, tab_entries, vals_cap, q, f, cells);
  MRK_HIP(hipGetLastError());
}

// a gang of `n_slots` persistent workgroups of `threads` lanes with `lds` bytes of dynamic LDS each, one kernel on `stream`
// (capi_rank.cpp mrk_serve_*)
void launch_rank_serve(mrk_ctx *ctx, hipStream_t stream, const StoreDev &st, const ProgramDev &prog, const QsDev &q, const QsForestDev &f, const ServeGangDev &gang,
                       int n_slots, int threads, size_t lds, bool f64, void *jit_fn) {
  if (jit_fn) {
    StoreDev a_st = st;
    QsDev a_q = q;
    QsForestDev a_f = f;
    ServeGangDev a_s = gang;
    void *args[] = {&a_st, &a_q, &a_f, &a_s};
    MRK_HIP(hipModuleLaunchKernel((hipFunction_t)jit_fn, (unsigned)n_slots, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, stream, args, nullptr));
    return;
  }
  lds_optin(ctx, f64 ? (const void *)rank_serve_kernel<true> : (const void *)rank_serve_kernel<false>);
  if (f64) hipLaunchKernelGGL(rank_serve_kernel<true>, dim3(n_slots), dim3(threads), lds, stream, st, prog, q, f, gang);
  else hipLaunchKernelGGL(rank_serve_kernel<false>, dim3(n_slots), dim3(threads), lds, stream, st, prog, q, f, gang);
  MRK_HIP(hipGetLastError());
}

// Normalize.scale over matrix column `col` of every request of the batch (after assembly and overrides)
void launch_normalize(mrk_ctx *ctx, const BatchDev &b, int dim, int col, int mode) {
  if (b.n_req <= 0 || mode == NORM_NOOP) return;
  ScopedKernelTimer timer(ctx, "normalize");
  hipLaunchKernelGGL(normalize_kernel, dim3(b.n_req), dim3(SORT_THREADS), 0, ctx->launch, b, dim, col, mode);
  MRK_HIP(hipGetLastError());
}

// item-sharded runs: status[r] |= OR over the ranks of all[rank][r]
__global__ void status_or_kernel(const int32_t *__restrict__ all, int world, int n_req, int32_t *__restrict__ status) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_req) return;
  int32_t v = status[r];
  for (int w = 0; w < world; ++w) v |= all[(size_t)w * n_req + r];
  status[r] = v;
}

void launch_status_or(hipStream_t stream, const int32_t *all, int world, int n_req, int32_t *status) {
  if (n_req <= 0) return;
  hipLaunchKernelGGL(status_or_kernel, dim3((n_req + 255) / 256), dim3(256), 0, stream, all, world, n_req, status);
  MRK_HIP(hipGetLastError());
}

// max_items: the batch's largest request (requests of more than SORT_MAX_ITEMS are ordered by bigsort.hip)
void launch_sort(mrk_ctx *ctx, const BatchDev &b, int max_items) {
  if (b.n_req <= 0) return;
  int cap = 64;
  while (cap < std::min(max_items, SORT_MAX_ITEMS)) cap <<= 1;
  const size_t lds = cap <= SORT_COUNT_MAX ? (size_t)cap * 8 : (size_t)cap * 12;
  ScopedKernelTimer timer(ctx, "sort");
  hipLaunchKernelGGL(sort_kernel, dim3(b.n_req), dim3(SORT_THREADS), lds, ctx->launch, b, cap);
  MRK_HIP(hipGetLastError());
}

}  // namespace mrk
