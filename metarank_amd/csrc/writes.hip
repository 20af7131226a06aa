// Write path on the device: raw PeriodicIncrements -> per-key bucket rings in HBM -> the
// window sums the read path consumes.  Reference: flow/FeatureValueFlow.scala:44-101 (commitWrite / makeValue),
// fstore/memory/MemPeriodicCounter.scala:16-37 (bucket = ts.toStartOfPeriod(period), counts per bucket),
// model/Feature.scala:142-161 (PeriodicCounterFeature.fromMap: for PeriodRange(p, 0), anchored at the LATEST
// bucket present: start = last - p * period, end = last + period, sum of the buckets in [start, end]).
//
// The reference keeps every bucket for ever; only the max(p) + 1 most recent bucket indices can ever fall
// into a window again (the anchor never moves back), so a ring of that many {bucket start, count} entries,
// indexed by bucket index mod W, is exact: an older bucket that aliases a newer one is outside every future
// window and is dropped.  One lane owns one (slot, column) group, so increments of a batch that hit the
// same key are applied sequentially and deterministically; groups are independent.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "store.hpp"

namespace mrk {

namespace {

struct RingEntry {
  long long ts;      // bucket start (ms); RING_EMPTY = unused
  long long count;
};

__device__ __forceinline__ void recompute_sums(uint8_t *rec, const RingEntry *ring, const RingColDev &c) {
  long long last = 0;
  bool any = false;
  for (uint32_t i = 0; i < c.w; ++i) {
    const long long ts = ring[i].ts;
    if ((unsigned long long)ts == RING_EMPTY) continue;
    last = any ? (ts > last ? ts : last) : ts;
    any = true;
  }
  if (!any) return;  // never incremented: whatever a put stored stays
  const long long end = last + c.period_ms;  // lastTimestamp.minus(period * endOffset = 0).plus(period)
  for (int k = 0; k < c.n_ranges; ++k) {
    const long long start = last - (long long)c.offsets[k] * c.period_ms;
    long long sum = 0;
    for (uint32_t i = 0; i < c.w; ++i) {
      const long long ts = ring[i].ts;
      if ((unsigned long long)ts == RING_EMPTY) continue;
      if (ts <= end && ts >= start) sum += ring[i].count;
    }
    *(long long *)(rec + c.val_off + 8 * k) = sum;
  }
  rec[c.tag_index] = (uint8_t)(1 + c.n_ranges);  // PeriodicCounterValue with n_ranges values (store.hpp: Tag)
}

__global__ void __launch_bounds__(256)
periodic_apply_kernel(uint8_t *rows, uint32_t stride, uint8_t *ring, uint32_t ring_stride, const RingColDev *cols,
                      const IncGroup *groups, int n_groups, const IncUpdate *updates) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n_groups) return;
  const IncGroup grp = groups[g];
  const RingColDev c = cols[grp.col];
  RingEntry *r = (RingEntry *)(ring + (size_t)grp.slot * ring_stride + c.ring_off);
  for (uint32_t u = grp.begin; u < grp.end; ++u) {
    const long long b = updates[u].bucket;
    long long q = b / c.period_ms;  // exact: b is a multiple of the period
    long long idx = q % (long long)c.w;
    if (idx < 0) idx += c.w;
    RingEntry e = r[idx];
    if ((unsigned long long)e.ts != RING_EMPTY && e.ts == b) e.count += updates[u].inc;
    else if ((unsigned long long)e.ts == RING_EMPTY || b > e.ts) { e.ts = b; e.count = updates[u].inc; }
    else continue;  // a bucket at least W periods older than one already seen: outside every future window
    r[idx] = e;
  }
  recompute_sums(rows + (size_t)grp.slot * stride, r, c);
}

// rows [slot_lo, slot_hi) were just re-uploaded from the host mirror: recompute their ring-fed cells
__global__ void __launch_bounds__(256)
periodic_refresh_kernel(uint8_t *rows, uint32_t stride, uint8_t *ring, uint32_t ring_stride, const RingColDev *cols, int n_cols,
                        uint32_t slot_lo, uint32_t slot_hi) {
  const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long n = (unsigned long long)(slot_hi - slot_lo) * (unsigned)n_cols;
  if (t >= n) return;
  const uint32_t slot = slot_lo + (uint32_t)(t / (unsigned)n_cols);
  const RingColDev c = cols[t % (unsigned)n_cols];
  recompute_sums(rows + (size_t)slot * stride, (const RingEntry *)(ring + (size_t)slot * ring_stride + c.ring_off), c);
}

}  // namespace

void launch_periodic_apply(hipStream_t stream, uint8_t *rows, uint32_t stride, uint8_t *ring, uint32_t ring_stride,
                           const RingColDev *cols, const IncGroup *groups, int n_groups, const IncUpdate *updates) {
  if (n_groups <= 0) return;
  hipLaunchKernelGGL(periodic_apply_kernel, dim3((n_groups + 255) / 256), dim3(256), 0, stream, rows, stride, ring, ring_stride, cols,
                     groups, n_groups, updates);
  MRK_HIP(hipGetLastError());
}

void launch_periodic_refresh(hipStream_t stream, uint8_t *rows, uint32_t stride, uint8_t *ring, uint32_t ring_stride,
                             const RingColDev *cols, int n_cols, uint32_t slot_lo, uint32_t slot_hi) {
  if (slot_hi <= slot_lo || n_cols <= 0) return;
  const unsigned long long n = (unsigned long long)(slot_hi - slot_lo) * (unsigned)n_cols;
  hipLaunchKernelGGL(periodic_refresh_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rows, stride, ring, ring_stride,
                     cols, n_cols, slot_lo, slot_hi);
  MRK_HIP(hipGetLastError());
}

}  // namespace mrk
