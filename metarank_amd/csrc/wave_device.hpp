// Wave-local helpers of the wave-per-section pre-pass (rank_device.hpp prepass_diversity_wave): a prefix scan by ballot and
// the commons-math LEGACY median of up to 64 values held one per lane - lanes of ONE wavefront talking through LDS in
// program order, no workgroup barrier.  Its own header so that tests/native/wave_test.cpp can compile it for the host: 64
// threads stand in for the 64 lanes, the ballot and the LDS ordering point become barriers (both are only ever reached by all
// lanes together), and the medians are compared bit for bit with a sorted-array restatement of the percentile.
#ifndef MRK_WAVE_DEVICE_HPP
#define MRK_WAVE_DEVICE_HPP

namespace mrk {

namespace {

// keeps the compiler from moving this wavefront's LDS accesses across (the LDS unit executes them in order anyway)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exclusive prefix sum of a 0/1 flag over the wavefront + total
__device__ __forceinline__ int wave_scan_flag(bool flag, int &total) {
  const unsigned long long ball = __ballot(flag);
  total = __popcll(ball);
  return __popcll(ball & ((1ull << (threadIdx.x & 63)) - 1ull));
}

// median_of for n_raw <= 64 values, one wavefront (the rank-sort branch: one value per lane)
__device__ __forceinline__ double wave_median_of(double *s_vals, int n_raw) {
  const int lane = threadIdx.x & 63;
  if (n_raw == 1) return s_vals[0];
  const bool mine = lane < n_raw;
  const double v = mine ? s_vals[lane] : 0.0;
  const bool isn = v != v;
  const int n_nan = __popcll(__ballot(mine && isn));
  int rank = 0;
  if (mine && !isn)
    for (int j = 0; j < n_raw; ++j) {
      const double w = s_vals[j];
      rank += (w < v || (w == v && j < lane)) ? 1 : 0;
    }
  wave_lds_sync();
  if (mine && !isn) s_vals[rank] = v;
  wave_lds_sync();
  const int m = n_raw - n_nan;
  if (m <= 0) return __longlong_as_double(0x7ff8000000000000LL);
  const double pos = 0.5 * (double)(m + 1);
  const double fpos = floor(pos);
  const int ipos = (int)fpos;
  const double dif = pos - fpos;
  if (pos < 1.0) return s_vals[0];
  if (pos >= (double)m) return s_vals[m - 1];
  const double lower = s_vals[ipos - 1], upper = s_vals[ipos];
  return __dadd_rn(lower, __dmul_rn(dif, __dsub_rn(upper, lower)));
}

}  // namespace

}  // namespace mrk

#endif  // MRK_WAVE_DEVICE_HPP
