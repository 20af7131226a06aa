// Launch shapes chosen from a batch's shape: host-side arithmetic only (no HIP), so that the rules - each of them the
// outcome of an A/B measurement cited next to it - can be pinned by CPU tests (tests/test_launch_shapes_cpu.py).
#pragma once
#include <algorithm>
#include <cstddef>

namespace mrk {

// the fused assembly kernel (rank_device.hpp rank_fused_body): lanes per copy of the item lanes, copies that share the
// program's ops (op split), workgroups per request (slices)
struct FusedShape {
  int item_lanes = 64, split = 1, slices = 1;
  int threads() const { return item_lanes * split; }
};

// force_* = the MRK_FUSED_THREADS / MRK_FUSED_SPLIT / MRK_FUSED_SLICES switches (0 = by batch shape)
inline FusedShape fused_launch_shape(int n_req, int max_items, int force_threads = 0, int force_split = 0, int force_slices = 0, int split_max_req = 64) {
  FusedShape s;
  // the largest request rounded up to whole wavefronts, at most 256 lanes (100-item requests: 128 lanes measure 0.288 ms
  // on c2, 64 lanes - one wavefront running two serial rounds - 0.421)
  s.item_lanes = std::min(256, std::max(64, (max_items + 63) / 64 * 64));
  if (force_threads) s.item_lanes = force_threads;
  // A handful of requests cannot fill the chip with one or two wavefronts each: their workgroups get copies of the item
  // lanes that split the program's ops between them (<= 512 lanes per workgroup); p50 of a 100-item request 0.19 -> 0.158 ms
  if (s.item_lanes <= 256) {
    const int fit = s.item_lanes <= 128 ? 4 : 2;
    if (force_split) s.split = std::min(fit, force_split);
    // (round 6: up to 64 requests when they are small - mrk_rank's combined batches, r06_x; batches of 17 ... 64 LARGE requests keep
    //  their slices: 96 x 1 000 candidates 0.52 -> 0.21 ms with 4)
    else if (n_req <= 16 || (n_req <= split_max_req && s.item_lanes <= 128)) s.split = fit;
  }
  // Few LARGE requests (c3: 384 x 1 000 candidates = 1.5 workgroups per CU, each looping 4 times over its 256 lanes): cover
  // a request with several workgroups as long as the launch stays within ONE residency of the chip (4 096 wavefronts at
  // the 4 per SIMD the specialised kernel runs with) - measured (profiles/r02_r_slices.txt): 384 requests 0.62 -> 0.43 ms
  // with 2 slices (0.47 / 0.48 with 3 / 4: a second wave of workgroups pays the pre-pass again for nothing), 96 requests
  // 0.52 -> 0.21 ms with 4
  const int waves = s.item_lanes / 64;
  const int rounds = (max_items + s.item_lanes - 1) / std::max(s.item_lanes, 1);
  if (force_slices) s.slices = std::max(1, std::min(rounds, force_slices));
  else if (s.split == 1 && rounds > 1) s.slices = std::max(1, std::min(rounds, 4096 / std::max(1, n_req * waves)));
  return s;
}

// Wavefronts per 128-row tile of the bit-vector scorer (score_qs.hip launch_wave).  (1) Few tiles (a single request, a
// 100 000-candidate request on 256 CUs): more wavefronts per tile walk the forest that many times faster.  (2) Full
// batches: a tile's slab (V x 256 B) is what limits how many one-wavefront workgroups a CU holds - 15 at V = 41, 6 at
// V = 100 - and the kernel needs ~8 wavefronts per SIMD to keep the VALU fed; NW wavefronts SHARING one slab multiply the
// residency.  Measured on 384 000 rows x 500 trees (profiles/r02_l): V = 41: 1 -> 0.283 ms, 2 -> 0.317, 4 -> 0.218,
// 8 -> 0.220, 16 -> 0.230; 64 columns (V ~ 100): 1 -> 0.547, 4 -> 0.280, 8 -> 0.248.  So: the smallest NW whose
// workgroups fill a CU's wavefront slots, else the NW with the most resident wavefronts.
inline size_t scorer_split_lds(int V, int nw, int leaves, int tile_rows, bool f64) {
  return (size_t)V * 256 + (nw > 1 ? (size_t)8 * nw * ((size_t)leaves * (f64 ? 8 : 4) + tile_rows) : 0);
}
inline int scorer_waves_per_tile(long long n_tiles, int V, bool f64, int n_cus, int leaves, int tile_rows) {
  const long long simds = 4LL * std::max(n_cus, 1);
  int nw = 1, best_waves = 0;
  for (int n : {1, 4, 8}) {  // (16 is kept for single requests: measured 5 % behind 4 on a full batch)
    const size_t lds_n = std::max<size_t>(scorer_split_lds(V, n, leaves, tile_rows, f64), 256);  // (single-leaf trees only: V = 0)
    if (lds_n > 160 * 1024) break;
    const int resident = std::min<int>(32, n * (int)((160 * 1024) / lds_n));  // wavefronts per CU (32 slots)
    if (resident > best_waves) { best_waves = resident; nw = n; }
    if (resident >= 28) break;
  }
  // Few tiles: the kernel is bound by instruction issue and wants ~6 wavefronts per SIMD to keep the VALU fed - the smallest
  // split that gets there, at most 16.  Same box (profiles/r04_h_ab.txt, 500 trees): 100 000 rows = 782 tiles: 4 -> 0.098 ms,
  // 8 -> 0.083, 16 -> 0.083; 20 000 rows = 157 tiles: 4 -> 0.066, 8 -> 0.045, 16 -> 0.032; 400 000 rows = 3 125 tiles: 4 -> 0.231,
  // 8 -> 0.233, 16 -> 0.242 (round 3's rule aimed at 2 wavefronts per SIMD and gave the first two 4 and 8).
  if (V <= 0) return nw;   // a forest of single-leaf trees: nothing to evaluate, one wavefront adds the leaves
  int fill = 1;
  const long long want = 6 * simds;
  while (fill < 16 && n_tiles * fill < want) fill *= 2;
  return std::max(nw, fill);
}

}  // namespace mrk
