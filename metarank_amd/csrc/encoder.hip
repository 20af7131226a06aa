// BERT-family encoder forward pass for gfx950 (the only MFMA work on the /rank path).
//
// Replaces `session.run` of OnnxBiEncoder.embed / OnnxCrossEncoder.encode (ml/onnx/sbert/OnnxBiEncoder.scala:29,
// OnnxCrossEncoder.scala:38) and the post-processing around it (avgpool :36-60; logits(j)(0) :40-45).
//
// Numerics: weights and the operands of every matrix product are fp16, products accumulate in f32 on the matrix
// cores (v_mfma_f32_32x32x16_f16); the residual stream, LayerNorm statistics, softmax and GELU are f32; the mean
// pool accumulates in f64 in token order exactly as avgpool does.
//
// Layouts: activations are token-major [n*seq, width]; Linear weights stay in the checkpoint's [out, in] order, which
// is already the "B transposed" form an MFMA B-fragment wants (8 consecutive k per lane = one 16-byte load).
// MFMA 32x32x16 fragments (guide §3): A lane l = row l&31, k = (l>>5)*8 + j; B lane l = col l&31, same k;
// C/D lane l = col l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdlib>

#include "encoder.hpp"

namespace mrk {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int LN_MAX_PER_LANE = 16;  // hidden <= 1024

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LayerNorm of one row held as v[e] = row[e*64 + lane]; two-pass statistics in f32 (as the f32 graph computes them)
__device__ __forceinline__ void layer_norm_row(float (&v)[LN_MAX_PER_LANE], int H, int lane, const float *g, const float *b, float eps,
                                               float *x_out, _Float16 *xh_out) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < LN_MAX_PER_LANE; ++e) if (e * 64 + lane < H) s += v[e];
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < LN_MAX_PER_LANE; ++e) if (e * 64 + lane < H) { const float d = v[e] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
  for (int e = 0; e < LN_MAX_PER_LANE; ++e) {
    const int c = e * 64 + lane;
    if (c < H) {
      const float o = (v[e] - mean) * rstd * g[c] + b[c];
      x_out[c] = o;
      xh_out[c] = (_Float16)o;
    }
  }
}

// word + position + token-type embeddings, then LayerNorm: one wavefront per token
// (pos_ids != nullptr: the tokens are PACKED - sequences back to back without padding - and pos_ids[t] is the token's
// position inside its sequence; else the batch is padded to `seq` tokens per row)
template <typename T>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t *ids, const int32_t *types, const int32_t *pos_ids, int M, int seq, int H, int vocab, int type_vocab,
                                                       const T *word, const T *pos, const T *type, const float *g, const float *b,
                                                       float eps, float *x, _Float16 *xh) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= M) return;
  int id = ids[t], ty = types[t];
  id = id < 0 ? 0 : id >= vocab ? vocab - 1 : id;
  ty = ty < 0 ? 0 : ty >= type_vocab ? type_vocab - 1 : ty;
  const int p = pos_ids ? pos_ids[t] : t % seq;
  float v[LN_MAX_PER_LANE];
#pragma unroll
  for (int e = 0; e < LN_MAX_PER_LANE; ++e) {
    const int c = e * 64 + lane;
    v[e] = c < H ? ((float)word[(size_t)id * H + c] + (float)pos[(size_t)p * H + c]) + (float)type[(size_t)ty * H + c] : 0.f;
  }
  layer_norm_row(v, H, lane, g, b, eps, x + (size_t)t * H, xh + (size_t)t * H);
}

__global__ __launch_bounds__(256) void ln_kernel(const float *y, int M, int H, const float *g, const float *b, float eps, float *x, _Float16 *xh) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= M) return;
  float v[LN_MAX_PER_LANE];
#pragma unroll
  for (int e = 0; e < LN_MAX_PER_LANE; ++e) {
    const int c = e * 64 + lane;
    v[e] = c < H ? y[(size_t)t * H + c] : 0.f;
  }
  layer_norm_row(v, H, lane, g, b, eps, x + (size_t)t * H, xh + (size_t)t * H);
}

// erf-GELU, 0.5 x (1 + erf(x / sqrt 2)), with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the
// fp16 the result is stored in): ~20 VALU operations instead of the ~100 of the branchy libm erff, which made the
// FFN epilogue -- 1536 activations per token -- cost more than its matrix product.  Explicit fma/rcp/exp, so every
// kernel that inlines it computes the same bits.
__device__ __forceinline__ float gelu_erf(float v) {
  const float z = fabsf(v) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__fmaf_rn(0.3275911f, z, 1.0f));
  float p = __fmaf_rn(1.061405429f, t, -1.453152027f);
  p = __fmaf_rn(p, t, 1.421413741f);
  p = __fmaf_rn(p, t, -0.284496736f);
  p = __fmaf_rn(p, t, 0.254829592f);
  const float e = __expf(-(z * z));
  const float erf_abs = __fmaf_rn(-(p * t), e, 1.0f);
  return 0.5f * v * (1.0f + copysignf(erf_abs, v));
}

// ---- C[M,N] = A[M,K] * W[N,K]^T + bias, fused epilogue -------------------------------------------------
// With K = 384 (6 k tiles) a workgroup's life is prologue latency + epilogue, not MFMA issue: what pays is MANY resident
// workgroups (3 wavefronts per SIMD here) and a short epilogue.  Measured on the c5 batch (33.7 k tokens,
// profiles/r02_v_encoder_tiles.txt): LDS-DMA double-buffered variants of this kernel with 128x128, 128x192, 256x192 and
// 256x256 tiles - fewer, longer-lived workgroups - were all 7-9 % SLOWER than this one and were removed again; batching
// the residual loads of the epilogue and dropping the per-element bounds checks from full tiles was worth 20 %.
enum { EPI_F16 = 0, EPI_GELU_F16 = 1, EPI_RES_F32 = 2 };
constexpr int BK = 64, LDS_PAD = 8;


// f32 -> f16 of a FINISHED f32 value.  Without the (empty) asm the compiler may fold the last f32 multiply / add into
// `v_fma_mixlo_f16`, which rounds ONCE - and does so in some instantiations of an epilogue and not in others, so a value
// would depend on which kernel produced it (caught by the packed-vs-padded test: one f16 ulp in one `mid` element).
__device__ __forceinline__ _Float16 to_half(float v) {
  asm("" : "+v"(v));
  return (_Float16)v;
}

// a global-memory pointer the compiler keeps in scalar registers (then `p[lane_offset]` is one `global_load … v_off, s[base]`)
template <typename T>
__device__ __forceinline__ __attribute__((address_space(1))) T *scalar_ptr(T *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (__attribute__((address_space(1))) T *)(((unsigned long long)hi << 32) | lo);
}

// One 32 x 32 accumulator block out: bias, then GELU / residual, then f16 / f32.  (m_blk, n_blk) = the block's corner,
// uniform across the wavefront; the lane owns column `col` and rows rbase + (r & 3) + 8 (r >> 2).  RAGGED = the tile may
// run past row M.  Addresses are a scalar base per row + ONE per-lane 32-bit offset (16 address pairs in registers cost
// the kernel a wavefront per SIMD).  The residual values of a block are fetched together BEFORE its first store (a load
// per element between the stores is a trip to memory each: measured, the K = 384 product with the residual epilogue
// took 57 us for 4 us of MFMA work).
template <int EPI, bool RAGGED>
__device__ __forceinline__ void store_block(const floatx16 &acc, int m_blk, int n_blk, int rbase, int col, int M, int N, float bv,
                                            const float *__restrict__ res, void *__restrict__ out) {
  const size_t corner = (size_t)m_blk * N + n_blk;            // uniform
  const uint32_t lane_off = (uint32_t)rbase * (uint32_t)N + (uint32_t)col;
  const int m_lane = m_blk + rbase;
  if constexpr (EPI == EPI_RES_F32) {
    const float *rrow = res + corner;
    float *orow = (float *)out + corner;
    float rv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = (r & 3) + 8 * (r >> 2);
      const bool live = !RAGGED || m_lane + dr < M;
      rv[r] = live ? scalar_ptr(rrow + (size_t)dr * N)[lane_off] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = (r & 3) + 8 * (r >> 2);
      if (RAGGED && m_lane + dr >= M) continue;
      scalar_ptr(orow + (size_t)dr * N)[lane_off] = (acc[r] + bv) + rv[r];
    }
    asm volatile("" ::: "memory");  // one block's 16 loads in flight at a time
  } else {
    _Float16 *orow = (_Float16 *)out + corner;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = (r & 3) + 8 * (r >> 2);
      if (RAGGED && m_lane + dr >= M) continue;
      float v = acc[r] + bv;
      if (EPI == EPI_GELU_F16) v = gelu_erf(v);
      scalar_ptr(orow + (size_t)dr * N)[lane_off] = to_half(v);
    }
  }
}

template <int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const _Float16 *__restrict__ A, const _Float16 *__restrict__ W, const float *__restrict__ bias,
                                                   const float *__restrict__ res, void *__restrict__ out, int M, int N, int K) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  __shared__ _Float16 As[BM][BK + LDS_PAD];
  __shared__ _Float16 Bs[BN][BK + LDS_PAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  The 1-D grid is folded so that the
  // workgroups of one XCD walk the N tiles of the same rows of A back to back: A is then fetched from HBM by one L2
  // instead of by up to eight, and the (small) weight matrix sits in every L2.
  const int n_tiles = N / BN, m_tiles = (M + BM - 1) / BM, per_xcd = (m_tiles + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int m_tile = xcd * per_xcd + slot / n_tiles;
  if (m_tile >= m_tiles) return;
  const int m0 = m_tile * BM, n0 = (slot % n_tiles) * BN;
  const int lr = tid >> 3, lc = (tid & 7) * 8;  // staging: row within a 32-row slab, 8-half chunk of the 64-wide k block

  half8 ra[2 * WM], rb[2 * WN];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2 * WM; ++i) {
      int m = m0 + i * 32 + lr;
      m = m < M ? m : M - 1;
      ra[i] = *(const half8 *)(A + (size_t)m * K + k0 + lc);
    }
#pragma unroll
    for (int j = 0; j < 2 * WN; ++j) rb[j] = *(const half8 *)(W + (size_t)(n0 + j * 32 + lr) * K + k0 + lc);
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 2 * WM; ++i) *(half8 *)&As[i * 32 + lr][lc] = ra[i];
#pragma unroll
    for (int j = 0; j < 2 * WN; ++j) *(half8 *)&Bs[j * 32 + lr][lc] = rb[j];
  };

  floatx16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = K / BK;
  load_tiles(0);
  store_tiles();
  __syncthreads();
  const int fr = lane & 31, fk = (lane >> 5) * 8;
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) load_tiles((kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      half8 fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) fa[i] = *(const half8 *)&As[(wr * WM + i) * 32 + fr][kk + fk];
#pragma unroll
      for (int j = 0; j < WN; ++j) fb[j] = *(const half8 *)&Bs[(wc * WN + j) * 32 + fr][kk + fk];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < KT) {
      store_tiles();
      __syncthreads();
    }
  }

  const int col = lane & 31, rbase = 4 * (lane >> 5);
  const bool ragged = m0 + BM > M;  // uniform: only the last row tile pays for the bounds checks
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n_blk = n0 + (wc * WN + j) * 32, m_blk = m0 + (wr * WM + i) * 32;
      const float bv = bias[n_blk + col];
      if (ragged) store_block<EPI, true>(acc[i][j], m_blk, n_blk, rbase, col, M, N, bv, res, out);
      else store_block<EPI, false>(acc[i][j], m_blk, n_blk, rbase, col, M, N, bv, res, out);
    }
}

// Same product for M <= 64 rows (a request's query): nothing to amortise a tile pipeline over, so the critical path is
// what counts.  One workgroup per 32 output columns; its four wavefronts each read a quarter of the K range straight
// from global memory into registers (no LDS staging, every load in flight at once) and then continue ONE accumulator
// chain in turn, handing it over through LDS -- the MFMAs run in the same k order as in gemm_kernel, so a row's
// result does not depend on which of the two kernels produced it.
template <int MT, int STEPS, int EPI>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const _Float16 *__restrict__ A, const _Float16 *__restrict__ W, const float *__restrict__ bias,
                                                         const float *__restrict__ res, void *__restrict__ out, int M, int N, int K) {
  __shared__ float hand[MT][16][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fr = lane & 31, fk = (lane >> 5) * 8;
  const int n0 = blockIdx.x * 32, kbase = wave * STEPS * 16 + fk;
  const _Float16 *wrow = W + (size_t)(n0 + fr) * K + kbase;
  half8 b[STEPS], a[MT][STEPS];
#pragma unroll
  for (int u = 0; u < STEPS; ++u) b[u] = *(const half8 *)(wrow + 16 * u);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = i * 32 + fr;
    const _Float16 *arow = A + (size_t)(m < M ? m : M - 1) * K + kbase;
#pragma unroll
    for (int u = 0; u < STEPS; ++u) a[i][u] = *(const half8 *)(arow + 16 * u);
  }
  floatx16 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      if (w > 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = hand[i][r][lane];
      }
#pragma unroll
      for (int u = 0; u < STEPS; ++u)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][u], b[u], acc[i], 0, 0, 0);
      if (w < 3) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) hand[i][r][lane] = acc[i][r];
      }
    }
    if (w < 3) __syncthreads();
  }
  if (wave != 3) return;
  const int n = n0 + fr, rbase = 4 * (lane >> 5);
  const float bv = bias[n];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
      if (m >= M) continue;
      float v = acc[i][r] + bv;
      if (EPI == EPI_GELU_F16) v = gelu_erf(v);
      if (EPI == EPI_RES_F32) ((float *)out)[(size_t)m * N + n] = v + res[(size_t)m * N + n];
      else ((_Float16 *)out)[(size_t)m * N + n] = to_half(v);
    }
}

template <int MT, int EPI>
bool launch_skinny(const _Float16 *A, const _Float16 *W, const float *bias, const float *res, void *out, int M, int N, int K, hipStream_t s) {
#define MRK_SKINNY(STEPS)                                                                                                    \
  case STEPS:                                                                                                                \
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, STEPS, EPI>), dim3(N / 32), dim3(256), 0, s, A, W, bias, res, out, M, N, K);     \
    return true;
  if (K % 64) return false;
  switch (K / 64) {
    MRK_SKINNY(1) MRK_SKINNY(2) MRK_SKINNY(4) MRK_SKINNY(6) MRK_SKINNY(8) MRK_SKINNY(12) MRK_SKINNY(16) MRK_SKINNY(24)
    default: return false;
  }
#undef MRK_SKINNY
}

// ---- attention: one wavefront per (32 queries, head, sequence); scores are computed transposed (keys x queries)
// so that every lane owns one query column: the row-wise softmax is then a per-lane reduction plus one exchange
// with lane^32, and the probabilities already sit in B-fragment order for O^T = V^T P^T.
// Padded batches (cu == nullptr): sequence b owns rows [b * seq_pad, (b + 1) * seq_pad), `mask` says which keys are live.
// Packed batches (cu != nullptr): sequence b owns rows [cu[b], cu[b + 1]), every one of them live.  The arithmetic of a
// query row is the same in both: dead keys contribute exact zeros, whole dead key blocks leave the running state untouched.
// A workgroup is ATT_HEADS independent wavefronts, one head each (no LDS, no barrier): a batch of 3 840 nine-token
// queries is 46 080 wavefronts, and dispatching them four to a workgroup is cheaper than one by one.
constexpr int ATT_HEADS = 4;
template <int DH>
__global__ __launch_bounds__(64 * ATT_HEADS) void attention_kernel(const _Float16 *__restrict__ qkv, const int32_t *__restrict__ mask, const int32_t *__restrict__ cu,
                                                                  int seq_pad, int H, int heads, float scale, _Float16 *__restrict__ ctx) {
  constexpr int KC = DH / 16, DT = DH / 32;
  const int lane = threadIdx.x & 63, q = lane & 31, g = lane >> 5;
  const int q0 = blockIdx.x * 32, head = blockIdx.y * ATT_HEADS + ((int)threadIdx.x >> 6), b = blockIdx.z;
  if (head >= heads) return;
  const size_t row = (size_t)3 * H;
  const size_t first = cu ? (size_t)cu[b] : (size_t)b * seq_pad;
  const int seq = cu ? cu[b + 1] - cu[b] : seq_pad;
  if (q0 >= seq) return;  // (packed: the grid covers the longest sequence)
  const _Float16 *base = qkv + first * row + head * DH;
  const int32_t *mrow = mask ? mask + first : nullptr;

  const int qi = q0 + q < seq ? q0 + q : seq - 1;
  half8 qf[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) qf[c] = *(const half8 *)(base + (size_t)qi * row + c * 16 + g * 8);

  floatx16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = 0; k0 < seq; k0 += 32) {
    const int kr = k0 + q < seq ? k0 + q : seq - 1;  // A operand: this lane's key row
    floatx16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const half8 kf = *(const half8 *)(base + (size_t)kr * row + H + c * 16 + g * 8);
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[c], st, 0, 0, 0);
    }
    float p[16];
    float bm = -FLT_MAX;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * g;
      const bool live = key < seq && (mrow == nullptr || mrow[key < seq ? key : seq - 1] != 0);
      p[r] = live ? st[r] * scale : -FLT_MAX;
      bm = fmaxf(bm, p[r]);
    }
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    const float m_new = fmaxf(m_run, bm);
    const float alpha = __expf(m_run - m_new);
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = p[r] == -FLT_MAX && m_new != -FLT_MAX ? 0.f : __expf(p[r] - m_new);
      ls += p[r];
    }
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    // O^T += V^T P^T, 16 keys per MFMA; slot j of chunk c stands for key 16c + 8(j>>2) + 4g + (j&3) on BOTH operands
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      half8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (_Float16)p[8 * c + j];
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        half8 vf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          int key = k0 + 16 * c + 8 * (j >> 2) + 4 * g + (j & 3);
          key = key < seq ? key : seq - 1;
          vf[j] = base[(size_t)key * row + 2 * H + d * 32 + q];
        }
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[d], 0, 0, 0);
      }
    }
  }
  if (q0 + q >= seq) return;
  const float inv = 1.0f / l_run;
  _Float16 *dst = ctx + (first + q0 + q) * H + head * DH;
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      half4 h;
#pragma unroll
      for (int k = 0; k < 4; ++k) h[k] = (_Float16)(o[d][r4 * 4 + k] * inv);
      *(half4 *)(dst + d * 32 + 8 * r4 + 4 * g) = h;
    }
}

// OnnxBiEncoder.avgpool (OnnxBiEncoder.scala:36-60): f64 sum of the first sum(mask) tokens in token order
__global__ void meanpool_kernel(const float *x, const int32_t *mask, const int32_t *cu, int seq, int H, float *out) {
  const int b = blockIdx.x;
  const size_t first = cu ? (size_t)cu[b] : (size_t)b * seq;
  int cnt = 0;
  if (cu) cnt = cu[b + 1] - cu[b];
  else for (int j = 0; j < seq; ++j) cnt += mask[(size_t)b * seq + j];
  const int rows = cu ? cnt : seq;
  for (int d = threadIdx.x; d < H; d += blockDim.x) {
    double acc = 0.0;
    for (int j = 0; j < cnt && j < rows; ++j) acc += (double)x[(first + j) * H + d];
    out[(size_t)b * H + d] = (float)(acc / (double)cnt);
  }
}

// BertPooler (dense + tanh on the [CLS] row) and the 1-logit classifier: one workgroup per sequence
template <typename T>
__global__ __launch_bounds__(256) void classify_kernel(const float *x, const int32_t *cu, int seq, int H, const T *pw, const float *pb, const float *cw,
                                                       const float *cb, float *out) {
  extern __shared__ float sm[];  // H inputs | 4 partials
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *cls = x + (cu ? (size_t)cu[b] : (size_t)b * seq) * H;
  for (int c = tid; c < H; c += 256) sm[c] = cls[c];
  __syncthreads();
  float part = 0.f;
  for (int o = tid; o < H; o += 256) {
    float acc = 0.f;
    const T *w = pw + (size_t)o * H;
    for (int c = 0; c < H; ++c) acc += (float)w[c] * sm[c];
    part += tanhf(acc + pb[o]) * cw[o];
  }
  part = wave_sum(part);
  if ((tid & 63) == 0) sm[H + (tid >> 6)] = part;
  __syncthreads();
  if (tid == 0) out[b] = ((sm[H] + sm[H + 1]) + (sm[H + 2] + sm[H + 3])) + cb[0];
}

template <int EPI>
void launch_gemm(const uint16_t *A, const uint16_t *W, const float *bias, const float *res, void *out, int M, int N, int K, hipStream_t s) {
  // 128x128 tiles once the grid still covers the chip with them, 64x64 tiles otherwise
  const bool big = N % 128 == 0 && (size_t)((M + 127) / 128) * (N / 128) >= 256;
  const int which = EPI == EPI_F16 ? 1 : EPI == EPI_GELU_F16 ? 2 : K == N ? 4 : 8;
  const bool skinny = (switches().encoder_skinny & which) != 0;  // MRK_ENCODER_SKINNY: A/B switch (tests pin both kernels to the same bits)
  if (skinny && M <= 32 && launch_skinny<1, EPI>((const _Float16 *)A, (const _Float16 *)W, bias, res, out, M, N, K, s)) return;
  if (skinny && M > 32 && M <= 64 && launch_skinny<2, EPI>((const _Float16 *)A, (const _Float16 *)W, bias, res, out, M, N, K, s)) return;
  auto grid_of = [&](int bm, int bn) { return dim3((unsigned)(8 * (((M + bm - 1) / bm + 7) / 8) * (N / bn))); };
  if (big) hipLaunchKernelGGL((gemm_kernel<2, 2, EPI>), grid_of(128, 128), dim3(256), 0, s, (const _Float16 *)A, (const _Float16 *)W, bias, res, out, M, N, K);
  else hipLaunchKernelGGL((gemm_kernel<1, 1, EPI>), grid_of(64, 64), dim3(256), 0, s, (const _Float16 *)A, (const _Float16 *)W, bias, res, out, M, N, K);
}

// ---- precision f32 (mrk_encoder_load_ex(MRK_ENCODER_F32)): the arithmetic of the reference's fp32 ONNX session - f32
// operands, f32 accumulation, libm erf / exp - on the matrix cores.  gfx950's f32-input MFMA (v_mfma_f32_16x16x4_f32) is
// bit for bit a chain of f32 fma's, one rounding per product, at the f32 vector rate (157 TF dense): exact f32, no TF32-like
// shortcut exists on this chip.  EVERY f32 product here - 128 x 128 tiles, 64 x 64 tiles, the <= 32-row kernel of a single
// request's query, and the plain-VALU twin kept as the test instrument - accumulates an output element as ONE chain that
// starts at 0 and walks k in the same canonical order, so a row's bits depend on neither the kernel nor the batch it
// travelled in: the order a lane's 16-byte operand loads give without any shuffling,
//     for t in 0 .. K/16:  for c in 0..4:  for q in 0..4:  k = 16 t + 4 q + c        (f32_chain_k below)
// (lane (r, q) of a 16x16x4 operand holds k-slot q of row r; it loads row r's floats 16t + 4q .. + 3 at once and feeds
// component c to the c-th of four MFMAs).  The products are computed TRANSPOSED - the weight rows are the MFMA's A operand,
// the token rows its B operand - so that a lane ends up with 4 consecutive output columns of one token: one 16-byte store,
// one 16-byte bias / residual load.  (a x b is commutative: the transposition does not change a bit.)
enum { EPI32_NONE = 0, EPI32_GELU = 1, EPI32_RES = 2 };
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int F32_BK = 32, F32_LD = F32_BK + 4;   // row stride 36 floats: a 16-lane group's 16-byte fragment reads cover all 64 banks once

__host__ __device__ constexpr int f32_chain_k(int i) { return 16 * (i >> 4) + 4 * (i & 3) + ((i >> 2) & 3); }  // i-th k of the chain

// bias, then GELU / residual, of the 4 consecutive columns n4 .. n4 + 3 of row m
template <int EPI>
__device__ __forceinline__ void store_f32_quad(const floatx4 &acc, int m, int n4, int M, int N, const float *__restrict__ bias, const float *__restrict__ res,
                                               float *__restrict__ out) {
  if (m >= M) return;
  const floatx4 bv = *(const floatx4 *)(bias + n4);
  floatx4 rv = {0.f, 0.f, 0.f, 0.f};
  if (EPI == EPI32_RES) rv = *(const floatx4 *)(res + (size_t)m * N + n4);
  floatx4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = acc[e] + bv[e];
    if (EPI == EPI32_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    if (EPI == EPI32_RES) v += rv[e];
    o[e] = v;
  }
  *(floatx4 *)(out + (size_t)m * N + n4) = o;
}

// C[M, N] = A[M, K] W[N, K]^T + bias (+ GELU | + res).  Four wavefronts in 2 x 2, each TM x TN blocks of 16 x 16
// (TM = TN = 2: 64 x 64 tiles for grids that 128 x 128 tiles would not cover the chip with; the 128 x 128 tile is
// gemm_f32_mfma32_kernel below).
// Operands staged through LDS in their natural layout, one 32-wide k block at a time, the next block's global loads in
// flight during the MFMAs (at 1/16 of the f16 rate a k block is ~4 000 cycles of matrix work per wavefront: prologue and
// barriers are noise here, unlike in gemm_kernel).  N % (32 TN) == 0, K % 32 == 0.
template <int TM, int TN, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
                                                            const float *__restrict__ res, float *__restrict__ out, int M, int N, int K) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  __shared__ __align__(16) float As[BM][F32_LD];   // 16-byte fragment reads: an LDS access off its natural alignment is replayed at 64 cycles
  __shared__ __align__(16) float Bs[BN][F32_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // the XCD fold of gemm_kernel: one XCD's workgroups walk the N tiles of the same rows of A back to back
  const int n_tiles = N / BN, m_tiles = (M + BM - 1) / BM, per_xcd = (m_tiles + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int m_tile = xcd * per_xcd + slot / n_tiles;
  if (m_tile >= m_tiles) return;
  const int m0 = m_tile * BM, n0 = (slot % n_tiles) * BN;
  const int lr = tid >> 3, lc = (tid & 7) * 4;   // staging: row within a 32-row slab, 4-float chunk of the k block

  floatx4 ra[TM], rb[TN];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int m = m0 + i * 32 + lr;
      m = m < M ? m : M - 1;
      ra[i] = *(const floatx4 *)(A + (size_t)m * K + k0 + lc);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) rb[j] = *(const floatx4 *)(W + (size_t)(n0 + j * 32 + lr) * K + k0 + lc);
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i) *(floatx4 *)&As[i * 32 + lr][lc] = ra[i];
#pragma unroll
    for (int j = 0; j < TN; ++j) *(floatx4 *)&Bs[j * 32 + lr][lc] = rb[j];
  };

  floatx4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int KT = K / F32_BK;
  load_tiles(0);
  store_tiles();
  __syncthreads();
  const int fr = lane & 15, fq = (lane >> 4) * 4;
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) load_tiles((kt + 1) * F32_BK);
#pragma unroll
    for (int g = 0; g < F32_BK; g += 16) {
      floatx4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *(const floatx4 *)&As[(wr * TM + i) * 16 + fr][g + fq];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *(const floatx4 *)&Bs[(wc * TN + j) * 16 + fr][g + fq];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][c], fa[i][c], acc[j][i], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < KT) {
      store_tiles();
      __syncthreads();
    }
  }
  // D[i = weight row][j = token]: the lane owns token fr of its block and output columns fq .. fq + 3
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
      store_f32_quad<EPI>(acc[j][i], m0 + (wr * TM + i) * 16 + fr, n0 + (wc * TN + j) * 16 + fq, M, N, bias, res, out);
}

// The 128 x 128 tile - what packed batches run on - on the 32 x 32 x 2 form of the instruction (64 cycles each, half as many
// instructions; each wavefront 2 x 2 blocks of 32 x 32; 152 registers = 3 workgroups per CU).  Same-box A/B against the same tile
// on 16 x 16 x 4 (184 registers, 2 per CU): the c5 batch's forward pass 8.46 -> 8.13 ms (profiles/r05_d_*), identical bits.  Lane (r = l & 31, g = l >> 5) holds k-slot g of row r; the canonical chain order "k = 16 t + 4 q + c, q inner"
// pairs q = 2 h, 2 h + 1 into step (c, h), so the lane's operand for that step is component c of the 16-byte piece at column
// 16 t + 8 h + 4 g of its row - the natural LDS layout again.  Same bits as gemm_f32_mfma_kernel and the <= 32-row kernel (tests).
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_mfma32_kernel(const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
                                                              const float *__restrict__ res, float *__restrict__ out, int M, int N, int K) {
  constexpr int BM = 128, BN = 128, TS = 4;
  __shared__ __align__(16) float As[BM][F32_LD];   // 16-byte fragment reads: an LDS access off its natural alignment is replayed at 64 cycles
  __shared__ __align__(16) float Bs[BN][F32_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int n_tiles = N / BN, m_tiles = (M + BM - 1) / BM, per_xcd = (m_tiles + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int m_tile = xcd * per_xcd + slot / n_tiles;
  if (m_tile >= m_tiles) return;
  const int m0 = m_tile * BM, n0 = (slot % n_tiles) * BN;
  const int lr = tid >> 3, lc = (tid & 7) * 4;
  floatx4 ra[TS], rb[TS];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < TS; ++i) {
      int m = m0 + i * 32 + lr;
      m = m < M ? m : M - 1;
      ra[i] = *(const floatx4 *)(A + (size_t)m * K + k0 + lc);
    }
#pragma unroll
    for (int j = 0; j < TS; ++j) rb[j] = *(const floatx4 *)(W + (size_t)(n0 + j * 32 + lr) * K + k0 + lc);
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < TS; ++i) *(floatx4 *)&As[i * 32 + lr][lc] = ra[i];
#pragma unroll
    for (int j = 0; j < TS; ++j) *(floatx4 *)&Bs[j * 32 + lr][lc] = rb[j];
  };
  floatx16 acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
  const int KT = K / F32_BK;
  load_tiles(0);
  store_tiles();
  __syncthreads();
  const int fr = lane & 31, fg = (lane >> 5) * 4;
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) load_tiles((kt + 1) * F32_BK);
#pragma unroll
    for (int g = 0; g < F32_BK; g += 16) {
      floatx4 fa[2][2], fb[2][2];   // [block][h]
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          fa[i][h] = *(const floatx4 *)&As[(wr * 2 + i) * 32 + fr][g + 8 * h + fg];
          fb[i][h] = *(const floatx4 *)&Bs[(wc * 2 + i) * 32 + fr][g + 8 * h + fg];
        }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][h][c], fa[i][h][c], acc[j][i], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < KT) {
      store_tiles();
      __syncthreads();
    }
  }
  // D[i = weight row][j = token]: the lane owns token fr; registers 4 v .. 4 v + 3 are weight rows 8 v + fg .. + 3 of the block
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const floatx4 quad = {acc[j][i][4 * v], acc[j][i][4 * v + 1], acc[j][i][4 * v + 2], acc[j][i][4 * v + 3]};
        store_f32_quad<EPI>(quad, m0 + (wr * 2 + i) * 32 + fr, n0 + (wc * 2 + j) * 32 + 8 * v + fg, M, N, bias, res, out);
      }
}

// The same product for M <= 16 MT rows - ONE request's query (MRK_ENCODER_AUTO / _F32 through mrk_rank), where the critical
// path is what counts.  One workgroup per 16 output columns; its four wavefronts each read a quarter of the K range straight
// from global memory into registers (every load in flight at once) and continue ONE accumulator chain in turn, handing it
// over through LDS - the chain of gemm_f32_mfma_kernel, hence its bits.  GROUPS = 16-wide k groups per wavefront = K / 64.
template <int MT, int GROUPS, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_skinny_kernel(const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
                                                              const float *__restrict__ res, float *__restrict__ out, int M, int N, int K) {
  __shared__ floatx4 hand[MT][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fr = lane & 15, fq = (lane >> 4) * 4;
  const int n0 = blockIdx.x * 16, kbase = wave * GROUPS * 16 + fq;
  const float *wrow = W + (size_t)(n0 + fr) * K + kbase;
  floatx4 b[GROUPS], a[MT][GROUPS];
#pragma unroll
  for (int t = 0; t < GROUPS; ++t) b[t] = *(const floatx4 *)(wrow + 16 * t);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = i * 16 + fr;
    const float *arow = A + (size_t)(m < M ? m : M - 1) * K + kbase;
#pragma unroll
    for (int t = 0; t < GROUPS; ++t) a[i][t] = *(const floatx4 *)(arow + 16 * t);
  }
  floatx4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      if (w > 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = hand[i][lane];
      }
#pragma unroll
      for (int t = 0; t < GROUPS; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[t][c], a[i][t][c], acc[i], 0, 0, 0);
      if (w < 3) {
#pragma unroll
        for (int i = 0; i < MT; ++i) hand[i][lane] = acc[i];
      }
    }
    if (w < 3) __syncthreads();
  }
  if (wave != 3) return;
#pragma unroll
  for (int i = 0; i < MT; ++i) store_f32_quad<EPI>(acc[i], i * 16 + fr, n0 + fq, M, N, bias, res, out);
}

// The test instrument: the same chain on the vector unit (v_fma_f32, one output element per chain, k in f32_chain_k order).
// MRK_ENCODER_F32_MFMA=0 routes every f32 product here; tests/test_encoder_gpu.py requires the bits of the MFMA kernels.
// 64 x 64 tiles, 4 x 4 outputs per lane.  N % 64 == 0, K % 16 == 0.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
                                                       const float *__restrict__ res, float *__restrict__ out, int M, int N, int K) {
  constexpr int T = 64, KB = 16;
  __shared__ float As[KB][T + 4], Ws[KB][T + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += KB) {
    for (int e = tid; e < T * KB; e += 256) {
      const int r = e / KB, k = e % KB;
      const int m = m0 + r < M ? m0 + r : M - 1;
      As[k][r] = A[(size_t)m * K + k0 + k];
      Ws[k][r] = W[(size_t)(n0 + r) * K + k0 + k];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      const int k = f32_chain_k(kk);
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Ws[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __fmaf_rn(b[j], a[i], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nn = n0 + tx * 4 + j;
      float v = acc[i][j] + bias[nn];
      if (EPI == EPI32_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
      if (EPI == EPI32_RES) v += res[(size_t)m * N + nn];
      out[(size_t)m * N + nn] = v;
    }
  }
}

template <int MT, int EPI>
bool launch_f32_skinny(const float *A, const float *W, const float *bias, const float *res, float *out, int M, int N, int K, hipStream_t s) {
#define MRK_SKINNY32(G)                                                                                                          \
  case G:                                                                                                                        \
    hipLaunchKernelGGL((gemm_f32_skinny_kernel<MT, G, EPI>), dim3(N / 16), dim3(256), 0, s, A, W, bias, res, out, M, N, K);        \
    return true;
  if (K % 64 || N % 16) return false;
  switch (K / 64) {
    MRK_SKINNY32(1) MRK_SKINNY32(2) MRK_SKINNY32(4) MRK_SKINNY32(6) MRK_SKINNY32(8) MRK_SKINNY32(12) MRK_SKINNY32(16) MRK_SKINNY32(24)
    default: return false;
  }
#undef MRK_SKINNY32
}

template <int EPI>
void launch_gemm_f32(const float *A, const float *W, const float *bias, const float *res, float *out, int M, int N, int K, hipStream_t s) {
  if (switches().encoder_f32_mfma && K % F32_BK == 0 && N % 64 == 0) {
    if (M <= 16 && launch_f32_skinny<1, EPI>(A, W, bias, res, out, M, N, K, s)) return;
    if (M > 16 && M <= 32 && launch_f32_skinny<2, EPI>(A, W, bias, res, out, M, N, K, s)) return;
    // 128 x 128 tiles once the grid still covers the chip with them (two workgroups per CU), 64 x 64 tiles otherwise
    const bool big = N % 128 == 0 && (size_t)((M + 127) / 128) * (N / 128) >= 512;
    auto grid_of = [&](int bm, int bn) { return dim3((unsigned)(8 * (((M + bm - 1) / bm + 7) / 8) * (N / bn))); };
    // (Measured, profiles/r05_e / r05_f: with neither memory traffic nor barriers the 128 x 128 kernel's MFMA stream alone takes 0.240 of
    //  the QKV product's 0.270 ms - 121 TFLOP/s is what the f32 matrix pipe sustains on the whole chip under load, the clock
    //  following the power budget; one, two or three workgroups per CU (LDS padding): 0.307 / 0.272 / 0.270 ms.  The kernel is at
    //  ~89 % of that ceiling; what is left is the k loop's loads, LDS round trip and barriers.)
    if (big) hipLaunchKernelGGL((gemm_f32_mfma32_kernel<EPI>), grid_of(128, 128), dim3(256), 0, s, A, W, bias, res, out, M, N, K);
    else hipLaunchKernelGGL((gemm_f32_mfma_kernel<2, 2, EPI>), grid_of(64, 64), dim3(256), 0, s, A, W, bias, res, out, M, N, K);
    return;
  }
  hipLaunchKernelGGL((gemm_f32_kernel<EPI>), dim3(N / 64, (M + 63) / 64), dim3(256), 0, s, A, W, bias, res, out, M, N, K);
}

// f32 attention on the same instruction: one wavefront per (16 queries, head, sequence), scores transposed (keys x
// queries) as in attention_kernel, so a lane owns one query column: softmax = a per-lane reduction over its 4 keys plus
// two exchanges (lane ^ 16, lane ^ 32), and the probabilities are already the B operand of O^T = V^T P^T (k-slot q of the
// MFMA = key 4 q + c = the lane's own p[c]).  Running maximum / sum over key blocks of 16, libm expf.  A query row's
// arithmetic depends on its own sequence only: dead keys contribute exact zeros, dead key blocks leave the state untouched.
template <int DH>
__global__ __launch_bounds__(64 * ATT_HEADS) void attention_f32_mfma_kernel(const float *__restrict__ qkv, const int32_t *__restrict__ mask,
                                                                           const int32_t *__restrict__ cu, int seq_pad, int H, int heads, float scale,
                                                                           float *__restrict__ ctx) {
  constexpr int KG = DH / 16;
  const int lane = threadIdx.x & 63, r = lane & 15, q4 = (lane >> 4) * 4;
  const int q0 = blockIdx.x * 16, head = blockIdx.y * ATT_HEADS + ((int)threadIdx.x >> 6), b = blockIdx.z;
  if (head >= heads) return;
  const size_t row = (size_t)3 * H;
  const size_t first = cu ? (size_t)cu[b] : (size_t)b * seq_pad;
  const int seq = cu ? cu[b + 1] - cu[b] : seq_pad;
  if (q0 >= seq) return;
  const float *base = qkv + first * row + head * DH;
  const int32_t *mrow = mask ? mask + first : nullptr;
  const int qi = q0 + r < seq ? q0 + r : seq - 1;
  floatx4 qf[KG];
#pragma unroll
  for (int t = 0; t < KG; ++t) qf[t] = *(const floatx4 *)(base + (size_t)qi * row + 16 * t + q4);
  floatx4 o[KG];
#pragma unroll
  for (int d = 0; d < KG; ++d) o[d] = floatx4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  for (int k0 = 0; k0 < seq; k0 += 16) {
    const int kr = k0 + r < seq ? k0 + r : seq - 1;
    floatx4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KG; ++t) {
      const floatx4 kf = *(const floatx4 *)(base + (size_t)kr * row + H + 16 * t + q4);
#pragma unroll
      for (int c = 0; c < 4; ++c) st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c], qf[t][c], st, 0, 0, 0);
    }
    float p[4];
    float bm = -FLT_MAX;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = k0 + q4 + e;
      const bool live = key < seq && (mrow == nullptr || mrow[key < seq ? key : seq - 1] != 0);
      p[e] = live ? st[e] * scale : -FLT_MAX;
      bm = fmaxf(bm, p[e]);
    }
    bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    // a block of dead keys (liveness does not depend on the query: the branch is uniform) leaves the state untouched - also
    // while no live key has been seen yet (m_run = -inf), where exp(p - m_new) would have been exp(0) for every dead key; and a
    // non-finite V at a dead key never meets a zero probability.  A sequence without any live key ends as 0 / 0 = NaN, like
    // attention_f32_kernel's and the fp32 oracle's.
    if (bm == -FLT_MAX) continue;
    const float m_new = fmaxf(m_run, bm);
    const float alpha = expf(m_run - m_new);
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p[e] = p[e] == -FLT_MAX ? 0.f : expf(p[e] - m_new);
      ls += p[e];
    }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < KG; ++d)
#pragma unroll
      for (int e = 0; e < 4; ++e) o[d][e] *= alpha;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int key = k0 + q4 + c;
      key = key < seq ? key : seq - 1;
#pragma unroll
      for (int d = 0; d < KG; ++d) o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(base[(size_t)key * row + 2 * H + d * 16 + r], p[c], o[d], 0, 0, 0);
    }
  }
  if (q0 + r >= seq) return;
  const float inv = 1.0f / l_run;   // (l_run == 0: no live key at all -> 0 * inf = NaN, as in attention_f32_kernel)
  float *dst = ctx + (first + q0 + r) * H + head * DH;
#pragma unroll
  for (int d = 0; d < KG; ++d) {
    floatx4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = o[d][e] * inv;
    *(floatx4 *)(dst + d * 16 + q4) = v;
  }
}

// head sizes the MFMA kernel does not cover: one wavefront per (query row, head, sequence): scores over the sequence's live
// keys, softmax, weighted sum of V.  Dynamic LDS: 4 wavefronts x seq floats (the probabilities).
__global__ __launch_bounds__(256) void attention_f32_kernel(const float *__restrict__ qkv, const int32_t *__restrict__ mask, const int32_t *__restrict__ cu,
                                                            int seq_pad, int H, int DH, float scale, float *__restrict__ ctx) {
  extern __shared__ float att_p[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, head = blockIdx.y, qi = blockIdx.x * 4 + wave;
  const size_t row = (size_t)3 * H;
  const size_t first = cu ? (size_t)cu[b] : (size_t)b * seq_pad;
  const int seq = cu ? cu[b + 1] - cu[b] : seq_pad;
  if (qi >= seq) return;
  const float *base = qkv + first * row + head * DH;
  const int32_t *mrow = mask ? mask + first : nullptr;
  float *p = att_p + (size_t)wave * seq_pad;
  const float *q = base + (size_t)qi * row;
  float mx = -FLT_MAX;
  for (int j = lane; j < seq; j += 64) {
    const bool live = mrow == nullptr || mrow[j] != 0;
    float s = 0.f;
    const float *k = base + (size_t)j * row + H;
    for (int d = 0; d < DH; ++d) s = __fmaf_rn(q[d], k[d], s);
    s = live ? s * scale : -FLT_MAX;
    p[j] = s;
    mx = fmaxf(mx, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int j = lane; j < seq; j += 64) {
    const float e = p[j] == -FLT_MAX ? 0.f : expf(p[j] - mx);
    p[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_wave_barrier();
  if (lane < DH) {
    float o = 0.f;
    const float *v = base + 2 * H + lane;
    for (int j = 0; j < seq; ++j) o = __fmaf_rn(p[j], v[(size_t)j * row], o);
    ctx[(first + qi) * H + head * DH + lane] = o / sum;
  }
}

}  // namespace

void encoder_reserve(const EncoderDev &enc, EncoderScratch &sc, int n, int seq) {
  const size_t M = (size_t)n * seq, H = enc.shape.hidden, I = enc.shape.inter;
  const size_t w = enc.f32 ? 4 : 2;  // bytes per activation between the products
  sc.x.reserve(M * H * 4);
  sc.xh.reserve(M * H * 2);
  sc.qkv.reserve(M * 3 * H * w);
  sc.ctx.reserve(M * H * w);
  sc.mid.reserve(M * I * w);
  sc.y.reserve(M * H * 4);
}

// Padded: sc.ids = [ids | type_ids | mask], 3 x n x seq.  Packed (M_packed > 0): sc.ids = [ids | type_ids | position ids],
// 3 x M_packed, then cu, n + 1 - the sequences' tokens back to back, no padding anywhere: every per-token kernel (the
// matrix products, LayerNorm, GELU) runs over real tokens only.
static void forward_impl(const EncoderDev &enc, EncoderScratch &sc, int n, int seq, int M_packed, hipStream_t s) {
  const EncoderShape &sh = enc.shape;
  const bool packed = M_packed > 0;
  const int M = packed ? M_packed : n * seq, H = sh.hidden, I = sh.inter, DH = H / sh.heads;
  if (M <= 0) return;
  const int32_t *ids = sc.ids.as<int32_t>(), *types = ids + M, *third = ids + 2 * (size_t)M;
  const int32_t *mask = packed ? nullptr : third, *pos_ids = packed ? third : nullptr, *cu = packed ? ids + 3 * (size_t)M : nullptr;
  float *x = sc.x.as<float>(), *y = sc.y.as<float>();
  uint16_t *xh = sc.xh.as<uint16_t>(), *qkv = sc.qkv.as<uint16_t>(), *ctx = sc.ctx.as<uint16_t>(), *mid = sc.mid.as<uint16_t>();
  const int row_blocks = (M + 3) / 4;
  const float scale = 1.0f / sqrtf((float)DH);
  if (enc.f32) {  // precision f32 (mrk_encoder_load_ex): the fp32 graph's arithmetic; qkv / ctx / mid hold f32 here
    float *qf = sc.qkv.as<float>(), *cf = sc.ctx.as<float>(), *mf = sc.mid.as<float>();
    hipLaunchKernelGGL(embed_ln_kernel<float>, dim3(row_blocks), dim3(256), 0, s, ids, types, pos_ids, M, seq, H, sh.vocab, sh.type_vocab, enc.word32, enc.pos32,
                       enc.type32, enc.embg, enc.embb, sh.eps, x, (_Float16 *)xh);
    for (size_t l = 0; l < enc.layers.size(); ++l) {
      const LayerDev &L = enc.layers[l];
      const LayerDev32 &W = enc.layers32[l];
      launch_gemm_f32<EPI32_NONE>(x, W.wqkv, L.bqkv, nullptr, qf, M, 3 * H, H, s);
      const dim3 ag32((seq + 15) / 16, (sh.heads + ATT_HEADS - 1) / ATT_HEADS, n);
      if (switches().encoder_f32_mfma && DH == 32)
        hipLaunchKernelGGL((attention_f32_mfma_kernel<32>), ag32, dim3(64 * ATT_HEADS), 0, s, (const float *)qf, mask, cu, seq, H, sh.heads, scale, cf);
      else if (switches().encoder_f32_mfma && DH == 64)
        hipLaunchKernelGGL((attention_f32_mfma_kernel<64>), ag32, dim3(64 * ATT_HEADS), 0, s, (const float *)qf, mask, cu, seq, H, sh.heads, scale, cf);
      else
        hipLaunchKernelGGL(attention_f32_kernel, dim3((seq + 3) / 4, sh.heads, n), dim3(256), (size_t)4 * seq * sizeof(float), s, (const float *)qf, mask, cu, seq, H,
                           DH, scale, cf);
      launch_gemm_f32<EPI32_RES>(cf, W.wo, L.bo, x, y, M, H, H, s);
      hipLaunchKernelGGL(ln_kernel, dim3(row_blocks), dim3(256), 0, s, (const float *)y, M, H, L.ln1g, L.ln1b, sh.eps, x, (_Float16 *)xh);
      launch_gemm_f32<EPI32_GELU>(x, W.w1, L.b1, nullptr, mf, M, I, H, s);
      launch_gemm_f32<EPI32_RES>(mf, W.w2, L.b2, x, y, M, H, I, s);
      hipLaunchKernelGGL(ln_kernel, dim3(row_blocks), dim3(256), 0, s, (const float *)y, M, H, L.ln2g, L.ln2b, sh.eps, x, (_Float16 *)xh);
    }
    MRK_HIP(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(embed_ln_kernel<_Float16>, dim3(row_blocks), dim3(256), 0, s, ids, types, pos_ids, M, seq, H, sh.vocab, sh.type_vocab, (const _Float16 *)enc.word,
                     (const _Float16 *)enc.pos, (const _Float16 *)enc.type, enc.embg, enc.embb, sh.eps, x, (_Float16 *)xh);
  for (const LayerDev &L : enc.layers) {
    launch_gemm<EPI_F16>(xh, L.wqkv, L.bqkv, nullptr, qkv, M, 3 * H, H, s);
    dim3 ag((seq + 31) / 32, (sh.heads + ATT_HEADS - 1) / ATT_HEADS, n);  // packed: seq = the longest sequence
    if (DH == 32) hipLaunchKernelGGL((attention_kernel<32>), ag, dim3(64 * ATT_HEADS), 0, s, (const _Float16 *)qkv, mask, cu, seq, H, sh.heads, scale, (_Float16 *)ctx);
    else hipLaunchKernelGGL((attention_kernel<64>), ag, dim3(64 * ATT_HEADS), 0, s, (const _Float16 *)qkv, mask, cu, seq, H, sh.heads, scale, (_Float16 *)ctx);
    launch_gemm<EPI_RES_F32>(ctx, L.wo, L.bo, x, y, M, H, H, s);
    hipLaunchKernelGGL(ln_kernel, dim3(row_blocks), dim3(256), 0, s, (const float *)y, M, H, L.ln1g, L.ln1b, sh.eps, x, (_Float16 *)xh);
    launch_gemm<EPI_GELU_F16>(xh, L.w1, L.b1, nullptr, mid, M, I, H, s);
    launch_gemm<EPI_RES_F32>(mid, L.w2, L.b2, x, y, M, H, I, s);
    hipLaunchKernelGGL(ln_kernel, dim3(row_blocks), dim3(256), 0, s, (const float *)y, M, H, L.ln2g, L.ln2b, sh.eps, x, (_Float16 *)xh);
  }
  MRK_HIP(hipGetLastError());
}

void encoder_forward(const EncoderDev &enc, EncoderScratch &sc, int n, int seq, hipStream_t s) { forward_impl(enc, sc, n, seq, 0, s); }
void encoder_forward_packed(const EncoderDev &enc, EncoderScratch &sc, int n, int max_len, int M, hipStream_t s) { forward_impl(enc, sc, n, max_len, M, s); }

void encoder_meanpool(const EncoderDev &enc, EncoderScratch &sc, int n, int seq, int M_packed, float *d_out, hipStream_t s) {
  if (n <= 0) return;
  const int32_t *base = sc.ids.as<int32_t>();
  const int32_t *mask = M_packed > 0 ? nullptr : base + 2 * (size_t)n * seq, *cu = M_packed > 0 ? base + 3 * (size_t)M_packed : nullptr;
  hipLaunchKernelGGL(meanpool_kernel, dim3(n), dim3(128), 0, s, (const float *)sc.x.as<float>(), mask, cu, seq, enc.shape.hidden, d_out);
  MRK_HIP(hipGetLastError());
}

void encoder_classify(const EncoderDev &enc, EncoderScratch &sc, int n, int seq, int M_packed, float *d_out, hipStream_t s) {
  if (n <= 0) return;
  const int H = enc.shape.hidden;
  const int32_t *cu = M_packed > 0 ? sc.ids.as<int32_t>() + 3 * (size_t)M_packed : nullptr;
  if (enc.f32)
    hipLaunchKernelGGL(classify_kernel<float>, dim3(n), dim3(256), (H + 4) * sizeof(float), s, (const float *)sc.x.as<float>(), cu, seq, H, enc.pool_w32, enc.pool_b,
                       enc.cls_w, enc.cls_b, d_out);
  else
    hipLaunchKernelGGL(classify_kernel<_Float16>, dim3(n), dim3(256), (H + 4) * sizeof(float), s, (const float *)sc.x.as<float>(), cu, seq, H,
                       (const _Float16 *)enc.pool_w, enc.pool_b, enc.cls_w, enc.cls_b, d_out);
  MRK_HIP(hipGetLastError());
}

}  // namespace mrk
