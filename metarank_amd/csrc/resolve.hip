// Item id -> ITEM-table slot on the device: the first hop of FeatureValueLoader.fromStateBackend
// (fstore/FeatureValueLoader.scala:11-25 builds Key(ItemScope(id), feature) for every candidate; model/Key.scala:7-10).
// The host hands over the UTF-8 bytes of the batch's item ids as they arrived; one lane per candidate hashes its id,
// probes the device mirror of the store's id map (store.hpp SlotMap, device_types.hpp IdTableDev), compares the bytes
// and writes the slot (-1: an item the store has never seen) together with the request the item belongs to.  The host
// touches no per-item data: no hashing, no probing, no per-item arrays to upload besides the ids themselves.
#include <hip/hip_runtime.h>

#include "rank.hpp"
#include "runtime.hpp"

namespace mrk {

namespace {

constexpr int RESOLVE_THREADS = 256;

__global__ void __launch_bounds__(RESOLVE_THREADS)
resolve_ids_kernel(IdTableDev tab, const uint8_t *__restrict__ bytes, const uint32_t *__restrict__ offs, uint32_t bytes_len,
                   const ReqDev *__restrict__ reqs, int n_req, int total, int32_t *__restrict__ item_slot, uint32_t *__restrict__ item_req,
                   int32_t *__restrict__ load_status) {
  const int i = blockIdx.x * RESOLVE_THREADS + threadIdx.x;
  if (i >= total) return;
  // the request of item i: the last request whose first item is <= i (requests without items share a start)
  int lo = 0, hi = n_req - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (reqs[mid].item_begin <= i) lo = mid; else hi = mid - 1;
  }
  item_req[i] = (uint32_t)lo;
  const uint32_t o0 = offs[i], o1 = offs[i + 1];
  if (o1 < o0 || o1 > bytes_len) {  // offsets come off the wire: never read outside the uploaded bytes
    atomicOr(&load_status[lo], ST_BAD_IDS);
    item_slot[i] = -1;
    return;
  }
  const uint8_t *id = bytes + o0;
  const uint32_t len = o1 - o0;
  int32_t slot = -1;
  if (tab.table != nullptr) {
    const uint64_t h = id_hash_bytes(id, len);
    for (uint32_t p = (uint32_t)h & tab.mask;; p = (p + 1) & tab.mask) {
      const IdEntry e = tab.table[p];
      if (e.hash == 0) break;
      if (e.hash == h) {
        const uint8_t *a = tab.arena + e.off;
        bool same = a[len] == 0;
        for (uint32_t k = 0; same && k < len; ++k) same = a[k] == id[k];
        if (same) { slot = (int32_t)e.slot; break; }
      }
    }
  }
  item_slot[i] = slot;
}

}  // namespace

void launch_resolve_ids(hipStream_t stream, const IdTableDev &tab, const uint8_t *d_bytes, const uint32_t *d_offs, uint32_t bytes_len,
                        const ReqDev *d_reqs, int n_req, int total, int32_t *d_item_slot, uint32_t *d_item_req, int32_t *d_load_status) {
  if (total <= 0 || n_req <= 0) return;
  hipLaunchKernelGGL(resolve_ids_kernel, dim3((total + RESOLVE_THREADS - 1) / RESOLVE_THREADS), dim3(RESOLVE_THREADS), 0, stream, tab,
                     d_bytes, d_offs, bytes_len, d_reqs, n_req, total, d_item_slot, d_item_req, d_load_status);
  MRK_HIP(hipGetLastError());
}

}  // namespace mrk
