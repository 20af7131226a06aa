// Ordering of requests with more than SORT_MAX_ITEMS candidates (BASELINE config 4: 100 000; bench c4x: 4 M): a SAMPLE
// SORT on (key, index) pairs - key = the java.lang.Double.compare order of the negated score (sort_device.hpp), index =
// the candidate's place in the request.  Pairs are distinct, so ANY correct sort of the pairs is the stable order
// `sortBy(-_.score)` asks for (ml/Ranker.scala:52-67) - which is what lets the scatter below be unordered.
//
//   1. S = 8 x NB pairs are drawn from the request (one per stratum of n / S consecutive candidates, at a hashed place
//      inside it), sorted - by ONE workgroup in LDS when S <= 4096, else by this same sort - and every 8th becomes a
//      splitter: NB buckets of about 1 024 pairs each.  Because the INDEX is part of a pair, a request whose scores are
//      all equal (NoopModel: every score 0.0) splits as evenly as any other.
//   2. classify: every pair finds its bucket by a binary search over the splitters (in LDS), bucket numbers are kept
//      (u16), a workgroup's counts per bucket go to its row of a [workgroups][buckets] table; one lane per bucket then
//      walks its column (counts -> offsets) and the last workgroup of that launch turns the totals into bucket starts.
//   3. scatter: a workgroup's share of bucket b starts at base[b] + table[w][b]; it writes its pairs there (order inside a
//      bucket: arbitrary).  No global atomics anywhere.
//   4. one workgroup per bucket sorts its pairs in LDS (counting for <= 256, a bitonic network up to 4 096) and writes
//      the request's order.  A bucket that outgrew LDS - with 8-fold oversampling the chance is ~1e-7 per bucket - is
//      sorted in place in global memory by the same workgroup (slow, correct; MRK_BIG_SORT_CAP shrinks the LDS limit so
//      that tests reach this path).
// Traffic per candidate: 8 B score in (x 2: classify, scatter) + 2 B bucket (out, in x 2) + 12 B pair out + 12 B pair in
// + 4 B order out: ~50 B, in 5 launches (9 when the sample itself needs the multi-workgroup sort: n > 524 288).  The
// merge sort this replaces (round 2: 1 024-element chunks + log2(n / 1024) merge passes, 24 B per candidate per pass)
// took 13 launches and 1.46 ms for 4 M candidates.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "runtime.hpp"
#include "sort_device.hpp"

namespace mrk {

namespace {

constexpr int BS_THREADS = 256;
constexpr int BS_OVERSAMPLE = 8;       // sample pairs per bucket
constexpr int BS_BUCKET_ITEMS = 1024;  // target bucket size
constexpr int BS_LOCAL_CAP = 4096;     // pairs a bucket's workgroup sorts in LDS
constexpr int BS_MAX_BUCKETS = 16384;
constexpr int BS_LDS_SPLITTERS = 4096; // more splitters than this are searched in global memory (n > 4 M)
constexpr int BS_ONE_WG = 4096;        // pairs ONE workgroup sorts (sample sort, 1 024 lanes)

__device__ __forceinline__ int sample_pos(int i, int n, int samples) {
  const long long lo = (long long)i * n / samples, hi = (long long)(i + 1) * n / samples;  // stratum i: hi > lo (n >= samples)
  uint32_t h = (uint32_t)i * 2654435761u;
  h ^= h >> 15;
  h *= 2246822519u;
  h ^= h >> 13;
  return (int)(lo + (long long)(h % (uint32_t)(hi - lo)));
}

// Bitonic network over nthr x E (key, index) pairs held IN REGISTERS, E per lane, blocked (lane t holds pairs t E ... t E +
// E - 1).  A step of stride j is a register move when j < E, a wavefront shuffle when the partner lane lies in the same
// wavefront (j / E < 64), and a trip through LDS - two barriers - only when it does not.  Round 3's first version kept the
// pairs in LDS and went there for every one of the 55 steps of a 1 024-pair sort: 8-byte keys at stride 2 j are an 8-way
// bank conflict for small j, and a step cost 0.8 us (46 us per bucket; the local sort was 45 % of a 4 M-candidate order).
template <int E>
__device__ __forceinline__ void bitonic_regs(unsigned long long (&k)[E], int (&x)[E], unsigned long long *s_key, int *s_idx) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int total = nthr * E;
  auto cmpex_keep = [](unsigned long long &ka, int &ia, unsigned long long kb, int ib, bool keep_min) {
    const bool b_lt = pair_lt(kb, ib, ka, ia);
    if (b_lt == keep_min) { ka = kb; ia = ib; }   // keep_min: take the other when it is smaller; else take it when it is larger
  };
  for (int size = 2; size <= total; size <<= 1) {
    // strides that cross wavefronts
    for (int j = size >> 1; j >= 64 * E; j >>= 1) {
      const int pt = tid ^ (j / E);
#pragma unroll
      for (int c = 0; c < E; ++c) { s_key[c * nthr + tid] = k[c]; s_idx[c * nthr + tid] = x[c]; }
      __syncthreads();
      const bool lower = (tid & (j / E)) == 0;
#pragma unroll
      for (int c = 0; c < E; ++c) {
        const bool up = ((tid * E + c) & size) == 0;
        cmpex_keep(k[c], x[c], s_key[c * nthr + pt], s_idx[c * nthr + pt], lower == up);
      }
      __syncthreads();
    }
    // strides inside a wavefront: shuffles
    for (int j = min(size >> 1, 32 * E); j >= E; j >>= 1) {
      const int m = j / E;
      const bool lower = (tid & m) == 0;
#pragma unroll
      for (int c = 0; c < E; ++c) {
        const unsigned lo = (unsigned)k[c], hi = (unsigned)(k[c] >> 32);
        const unsigned olo = (unsigned)__shfl_xor((int)lo, m, 64), ohi = (unsigned)__shfl_xor((int)hi, m, 64);
        const int oi = __shfl_xor(x[c], m, 64);
        const bool up = ((tid * E + c) & size) == 0;
        cmpex_keep(k[c], x[c], ((unsigned long long)ohi << 32) | olo, oi, lower == up);
      }
    }
    // strides inside a lane: registers (compile-time indices)
#pragma unroll
    for (int jj = E >> 1; jj > 0; jj >>= 1) {
      if (jj < size) {
#pragma unroll
        for (int c = 0; c < E; ++c) {
          if ((c & jj) == 0) {
            const bool up = ((tid * E + c) & size) == 0;
            const bool b_lt = pair_lt(k[c | jj], x[c | jj], k[c], x[c]);
            if (b_lt == up) {
              const unsigned long long tk = k[c]; k[c] = k[c | jj]; k[c | jj] = tk;
              const int tx = x[c]; x[c] = x[c | jj]; x[c | jj] = tx;
            }
          }
        }
      }
    }
  }
}

// sorts m <= nthr x E pairs of (gk, gi)[0, m) and hands pair number p, in order, to emit(p, key, index)
template <int E, typename Load, typename Emit>
__device__ __forceinline__ void sort_block(int m, unsigned long long *s_key, int *s_idx, Load load, Emit emit) {
  const int tid = threadIdx.x;
  unsigned long long k[E];
  int x[E];
#pragma unroll
  for (int c = 0; c < E; ++c) {
    const int i = tid * E + c;
    k[c] = ~0ull;         // padding sorts last (no pair has key ~0 AND index INT_MAX)
    x[c] = 0x7fffffff;
    if (i < m) load(i, k[c], x[c]);
  }
  bitonic_regs<E>(k, x, s_key, s_idx);
#pragma unroll
  for (int c = 0; c < E; ++c) {
    const int i = tid * E + c;
    if (i < m) emit(i, k[c], x[c]);
  }
}

// step 1 when the sample fits one workgroup: draw, sort, emit the nb - 1 splitters
__global__ void __launch_bounds__(1024)
ss_sample_sort_kernel(SortSrc src, int n, int samples, int nb, unsigned long long *__restrict__ spl_k, int *__restrict__ spl_i, uint32_t *__restrict__ ticket) {
  __shared__ unsigned long long s_key[BS_ONE_WG];
  __shared__ int s_idx[BS_ONE_WG];
  if (threadIdx.x == 0) *ticket = 0u;   // the classify pass's last-workgroup ticket (this launch precedes it on the stream: no memset command)
  auto load = [&](int i, unsigned long long &k, int &x) {
    const int p = sample_pos(i, n, samples);
    k = src_key(src, p);
    x = p;
  };
  auto emit = [&](int i, unsigned long long k, int x) {  // every BS_OVERSAMPLE-th pair is a splitter
    if ((i + 1) % BS_OVERSAMPLE == 0 && (i + 1) / BS_OVERSAMPLE <= nb - 1) {
      spl_k[(i + 1) / BS_OVERSAMPLE - 1] = k;
      spl_i[(i + 1) / BS_OVERSAMPLE - 1] = x;
    }
  };
  if (samples <= 1024) sort_block<1>(samples, s_key, s_idx, load, emit);
  else if (samples <= 2048) sort_block<2>(samples, s_key, s_idx, load, emit);
  else sort_block<4>(samples, s_key, s_idx, load, emit);
}

// step 1 for larger samples: draw ...
__global__ void __launch_bounds__(BS_THREADS)
ss_sample_gather_kernel(SortSrc src, int n, int samples, unsigned long long *__restrict__ smp_k, int *__restrict__ smp_i) {
  const int i = blockIdx.x * BS_THREADS + threadIdx.x;
  if (i >= samples) return;
  const int p = sample_pos(i, n, samples);
  smp_k[i] = src_key(src, p);
  smp_i[i] = p;
}
// ... (the sample is sorted by this same sort: equal keys in sample order = in candidate order, the strata ascend) ... pick
__global__ void __launch_bounds__(BS_THREADS)
ss_pick_splitters_kernel(const unsigned long long *__restrict__ smp_k, const int *__restrict__ smp_i, const int *__restrict__ smp_order,
                         int nb, unsigned long long *__restrict__ spl_k, int *__restrict__ spl_i, uint32_t *__restrict__ ticket) {
  const int j = blockIdx.x * BS_THREADS + threadIdx.x;
  if (j == 0) *ticket = 0u;
  if (j >= nb - 1) return;
  const int e = smp_order[(j + 1) * BS_OVERSAMPLE - 1];
  spl_k[j] = smp_k[e];
  spl_i[j] = smp_i[e];
}

// step 2.  Dynamic LDS: [nb u64 splitter keys][nb i32 splitter indices][nb u32 counters] (LDS_SPL), else the counters only.
// A workgroup leaves its tile's counts per bucket in ITS row of `table` ([n_wg][nb], coalesced) - no global atomics: 512
// workgroups x 4 096 buckets were 1.8 M device-scope atomics per pass, twice (round 3's first version: 250 us of a 4 M order).
template <bool LDS_SPL>
__global__ void __launch_bounds__(BS_THREADS)
ss_classify_kernel(SortSrc src, int n, int tile, int nb, const unsigned long long *__restrict__ spl_k, const int *__restrict__ spl_i,
                   uint16_t *__restrict__ bucket, uint32_t *__restrict__ table) {
  extern __shared__ unsigned long long bs_smem[];
  unsigned long long *s_k = bs_smem;
  int *s_i = (int *)(s_k + (LDS_SPL ? nb : 0));
  uint32_t *s_hist = (uint32_t *)(s_i + (LDS_SPL ? nb : 0));
  const int tid = threadIdx.x;
  if (LDS_SPL)
    for (int j = tid; j < nb - 1; j += BS_THREADS) { s_k[j] = spl_k[j]; s_i[j] = spl_i[j]; }
  for (int j = tid; j < nb; j += BS_THREADS) s_hist[j] = 0u;
  __syncthreads();
  const unsigned long long *kk = LDS_SPL ? s_k : spl_k;
  const int *ki = LDS_SPL ? s_i : spl_i;
  const int lo = blockIdx.x * tile, hi = min(lo + tile, n);
  for (int i = lo + tid; i < hi; i += BS_THREADS) {
    const unsigned long long key = src_key(src, i);
    int b = 0;  // number of splitters below (key, i): nb is a power of two, there are nb - 1 splitters
    for (int step = nb >> 1; step > 0; step >>= 1) {
      const int probe = b + step - 1;
      b += pair_lt(kk[probe], ki[probe], key, i) ? step : 0;
    }
    bucket[i] = (uint16_t)b;
    atomicAdd(&s_hist[b], 1u);
  }
  __syncthreads();
  uint32_t *row = table + (size_t)blockIdx.x * nb;
  for (int j = tid; j < nb; j += BS_THREADS) row[j] = s_hist[j];
}

// step 2b: one lane per bucket walks its column of the table - table[w][b] becomes the number of bucket-b pairs of the
// workgroups before w -, then the last workgroup to finish turns the bucket totals into bucket starts.
__global__ void __launch_bounds__(BS_THREADS)
ss_offsets_kernel(uint32_t *__restrict__ table, int n_wg, int nb, int n, uint32_t *totals, uint32_t *ticket, uint32_t *__restrict__ base) {
  __shared__ uint32_t s_part[BS_THREADS];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  const int b = blockIdx.x * BS_THREADS + tid;
  if (b < nb) {
    uint32_t run = 0;
    int w = 0;
    for (; w + 8 <= n_wg; w += 8) {  // the loads of eight rows in flight together
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = table[(size_t)(w + u) * nb + b];
#pragma unroll
      for (int u = 0; u < 8; ++u) { table[(size_t)(w + u) * nb + b] = run; run += v[u]; }
    }
    for (; w < n_wg; ++w) { const uint32_t v = table[(size_t)w * nb + b]; table[(size_t)w * nb + b] = run; run += v; }
    totals[b] = run;
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  const int per = (nb + BS_THREADS - 1) / BS_THREADS;
  uint32_t sum = 0;
  for (int q = 0; q < per; ++q) {
    const int j = tid * per + q;
    if (j < nb) sum += atomicAdd(&totals[j], 0u);  // written by other workgroups of this launch: a returning atomic reads the coherent value
  }
  s_part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0;
    for (int t = 0; t < BS_THREADS; ++t) { const uint32_t c = s_part[t]; s_part[t] = run; run += c; }
  }
  __syncthreads();
  uint32_t run = s_part[tid];
  for (int q = 0; q < per; ++q) {
    const int j = tid * per + q;
    if (j < nb) {
      base[j] = run;
      run += atomicAdd(&totals[j], 0u);
    }
  }
  if (tid == 0) base[nb] = (uint32_t)n;
}

// step 3.  Dynamic LDS: nb u32.  A workgroup's share of bucket b starts at base[b] + table[w][b]; inside it the order is
// arbitrary (LDS atomics).
__global__ void __launch_bounds__(BS_THREADS)
ss_scatter_kernel(SortSrc src, int n, int tile, int nb, const uint16_t *__restrict__ bucket, const uint32_t *__restrict__ table,
                  const uint32_t *__restrict__ base, unsigned long long *__restrict__ keys, int *__restrict__ idx) {
  extern __shared__ unsigned long long bs_smem[];
  uint32_t *s_off = (uint32_t *)bs_smem;
  const int tid = threadIdx.x;
  const uint32_t *row = table + (size_t)blockIdx.x * nb;
  for (int j = tid; j < nb; j += BS_THREADS) s_off[j] = base[j] + row[j];
  __syncthreads();
  const int lo = blockIdx.x * tile, hi = min(lo + tile, n);
  for (int i = lo + tid; i < hi; i += BS_THREADS) {
    const uint32_t pos = atomicAdd(&s_off[bucket[i]], 1u);
    keys[pos] = src_key(src, i);
    idx[pos] = i;
  }
}

// step 4: one workgroup per bucket
__global__ void __launch_bounds__(BS_THREADS)
ss_local_sort_kernel(const uint32_t *__restrict__ base, unsigned long long *keys, int *idx, int *__restrict__ out_order, int lds_cap) {
  __shared__ unsigned long long s_key[BS_LOCAL_CAP];
  __shared__ int s_idx[BS_LOCAL_CAP];
  const int tid = threadIdx.x;
  const int lo = (int)base[blockIdx.x], m = (int)base[blockIdx.x + 1] - lo;
  if (m <= 0) return;
  if (m <= lds_cap) {
    auto load = [&](int i, unsigned long long &k, int &x) { k = keys[lo + i]; x = idx[lo + i]; };
    auto emit = [&](int i, unsigned long long, int x) { out_order[lo + i] = x; };
    if (m <= BS_THREADS) sort_block<1>(m, s_key, s_idx, load, emit);
    else if (m <= 2 * BS_THREADS) sort_block<2>(m, s_key, s_idx, load, emit);
    else if (m <= 4 * BS_THREADS) sort_block<4>(m, s_key, s_idx, load, emit);
    else if (m <= 8 * BS_THREADS) sort_block<8>(m, s_key, s_idx, load, emit);
    else sort_block<16>(m, s_key, s_idx, load, emit);
    return;
  }
  // A bucket that does not fit LDS: an all-ascending bitonic network in place in global memory.  Every comparator puts
  // the smaller pair at the lower place, so the (virtual) padding above m never moves and comparators that touch it are
  // skipped.  One workgroup: its own writes are visible to it after the barrier.
  unsigned long long *gk = keys + lo;
  int *gi = idx + lo;
  int p2 = 1;
  while (p2 < m) p2 <<= 1;
  auto cmpex = [&](int i, int j) {
    if (j >= m) return;
    const unsigned long long ka = gk[i], kb = gk[j];
    const int ia = gi[i], ib = gi[j];
    if (pair_lt(kb, ib, ka, ia)) { gk[i] = kb; gk[j] = ka; gi[i] = ib; gi[j] = ia; }
  };
  for (int k = 2; k <= p2; k <<= 1) {
    const int half = k >> 1;
    for (int p = tid; p < p2 / 2; p += BS_THREADS) {  // first step of a merge: i against its mirror in the k-block
      const int i = (p / half) * k + (p % half);
      cmpex(i, i ^ (k - 1));
    }
    __threadfence_block();
    __syncthreads();
    for (int j = half >> 1; j > 0; j >>= 1) {
      for (int p = tid; p < p2 / 2; p += BS_THREADS) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        cmpex(i, i | j);
      }
      __threadfence_block();
      __syncthreads();
    }
  }
  for (int i = tid; i < m; i += BS_THREADS) out_order[lo + i] = gi[i];
}

struct Level {
  int n = 0, nb = 0, samples = 0, tile = 0, n_wg = 0;
  unsigned long long *keys = nullptr, *spl_k = nullptr, *smp_k = nullptr;
  int *idx = nullptr, *spl_i = nullptr, *smp_i = nullptr, *smp_order = nullptr;
  uint16_t *bucket = nullptr;
  uint32_t *totals = nullptr, *ticket = nullptr, *base = nullptr, *table = nullptr;
};

int pow2_ceil(long long v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// the levels of a sort of n pairs (level 0: the request; level 1: its sample, when that needs this sort itself) carved
// out of `scratch` (nullptr: sizes only); returns the bytes used
size_t plan_levels(int n, uint8_t *scratch, std::vector<Level> &out) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    uint8_t *p = scratch ? scratch + off : nullptr;
    off = (off + bytes + 255) / 256 * 256;
    return p;
  };
  for (int m = n; m > BS_ONE_WG;) {
    Level lv;
    lv.n = m;
    const int bucket_items = switches().big_sort_bucket >= 64 ? switches().big_sort_bucket : BS_BUCKET_ITEMS;
    lv.nb = std::min(BS_MAX_BUCKETS, std::max(8, pow2_ceil(((long long)m + bucket_items - 1) / bucket_items)));
    lv.samples = lv.nb * BS_OVERSAMPLE;
    lv.tile = std::max(1024, pow2_ceil(((long long)m + 511) / 512));  // at most 512 workgroups per pass
    if (switches().big_sort_tile >= 256) lv.tile = pow2_ceil(switches().big_sort_tile);
    lv.n_wg = (m + lv.tile - 1) / lv.tile;
    lv.keys = (unsigned long long *)take((size_t)m * 8);
    lv.idx = (int *)take((size_t)m * 4);
    lv.bucket = (uint16_t *)take((size_t)m * 2);
    lv.totals = (uint32_t *)take((size_t)lv.nb * 4 + 16);  // [totals][ticket]: zeroed together
    lv.ticket = lv.totals ? lv.totals + lv.nb : nullptr;
    lv.base = (uint32_t *)take((size_t)(lv.nb + 1) * 4);
    lv.table = (uint32_t *)take((size_t)lv.n_wg * lv.nb * 4);
    lv.spl_k = (unsigned long long *)take((size_t)lv.nb * 8);
    lv.spl_i = (int *)take((size_t)lv.nb * 4);
    if (lv.samples > BS_ONE_WG) {
      lv.smp_k = (unsigned long long *)take((size_t)lv.samples * 8);
      lv.smp_i = (int *)take((size_t)lv.samples * 4);
      lv.smp_order = (int *)take((size_t)lv.samples * 4);
    }
    out.push_back(lv);
    m = lv.samples;
  }
  return off;
}

void sort_level(mrk_ctx *ctx, hipStream_t s, const std::vector<Level> &lv, size_t k, const SortSrc &src, int *out_order, int lds_cap) {
  const Level &L = lv[k];
  if (L.samples <= BS_ONE_WG) {
    hipLaunchKernelGGL(ss_sample_sort_kernel, dim3(1), dim3(1024), 0, s, src, L.n, L.samples, L.nb, L.spl_k, L.spl_i, L.ticket);
  } else {
    hipLaunchKernelGGL(ss_sample_gather_kernel, dim3((L.samples + BS_THREADS - 1) / BS_THREADS), dim3(BS_THREADS), 0, s, src, L.n, L.samples,
                       L.smp_k, L.smp_i);
    const SortSrc child{nullptr, L.smp_k, 1, 0};
    sort_level(ctx, s, lv, k + 1, child, L.smp_order, lds_cap);
    hipLaunchKernelGGL(ss_pick_splitters_kernel, dim3((L.nb + BS_THREADS - 1) / BS_THREADS), dim3(BS_THREADS), 0, s, L.smp_k, L.smp_i,
                       L.smp_order, L.nb, L.spl_k, L.spl_i, L.ticket);
  }
  lds_optin(ctx, (const void *)ss_classify_kernel<true>, BS_LDS_SPLITTERS * 16);  // 4 096 buckets: 64 KB of dynamic LDS next to the kernel's static 1 KB
  if (L.nb <= BS_LDS_SPLITTERS)
    hipLaunchKernelGGL(ss_classify_kernel<true>, dim3(L.n_wg), dim3(BS_THREADS), (size_t)L.nb * 16, s, src, L.n, L.tile, L.nb, L.spl_k, L.spl_i,
                       L.bucket, L.table);
  else
    hipLaunchKernelGGL(ss_classify_kernel<false>, dim3(L.n_wg), dim3(BS_THREADS), (size_t)L.nb * 4, s, src, L.n, L.tile, L.nb, L.spl_k, L.spl_i,
                       L.bucket, L.table);
  // (round 5 measured counts -> offsets folded into the classify pass's last workgroup for tables of <= 128 rows - one launch
  //  less: 38.2 vs 35.6 us per 100 000-candidate order, profiles/r05_c_sort_bench.txt: the serial tail of one workgroup costs more
  //  than the launch it saves.  Not kept.  What did pay: no memset command for the ticket - the sample kernel zeroes it.)
  hipLaunchKernelGGL(ss_offsets_kernel, dim3((L.nb + BS_THREADS - 1) / BS_THREADS), dim3(BS_THREADS), 0, s, L.table, L.n_wg, L.nb, L.n, L.totals, L.ticket, L.base);
  hipLaunchKernelGGL(ss_scatter_kernel, dim3(L.n_wg), dim3(BS_THREADS), (size_t)L.nb * 4, s, src, L.n, L.tile, L.nb, L.bucket, L.table, L.base, L.keys,
                     L.idx);
  hipLaunchKernelGGL(ss_local_sort_kernel, dim3(L.nb), dim3(BS_THREADS), 0, s, L.base, L.keys, L.idx, out_order, lds_cap);
  MRK_HIP(hipGetLastError());
}

}  // namespace

// bytes of scratch launch_big_sort needs for a request of n candidates (0 when one workgroup sorts it)
size_t big_sort_scratch_bytes(int n) {
  std::vector<Level> lv;
  return plan_levels(n, nullptr, lv);
}

// out_order[0, n): the indices 0 .. n - 1 in the order of (key, index), key per `src` (sort_device.hpp); n > SORT_MAX_ITEMS.
// Enqueued on `stream`; `scratch` (big_sort_scratch_bytes(n)) must stay untouched until the launches have run.
void launch_big_sort(mrk_ctx *ctx, hipStream_t stream, const SortSrc &src, int n, int *out_order, void *scratch) {
  std::vector<Level> lv;
  plan_levels(n, (uint8_t *)scratch, lv);
  if (lv.empty()) throw StatusError(MRK_ERR_INVALID_ARG, "launch_big_sort: a request one workgroup sorts");
  const int cap = std::min(BS_LOCAL_CAP, std::max(0, switches().big_sort_cap));
  sort_level(ctx, stream, lv, 0, src, out_order, cap);
}

}  // namespace mrk
