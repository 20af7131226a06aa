// Sort keys of the ordering kernels (rank.hip: requests of <= SORT_MAX_ITEMS candidates in one workgroup; bigsort.hip:
// larger ones).  Reference: ml/Ranker.scala:52-67 (`sortBy(-_.score)`: stable, java.lang.Double.compare on the negated
// score) and ml/onnx/Normalize.scala:25-40 (ascending Ordering.Double).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mrk {

// unsigned keys whose integer order is java.lang.Double.compare's order of the doubles
__device__ __forceinline__ unsigned long long asc_key(double v) {
  unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  if (v != v) bits = 0x7ff8000000000000ULL;  // Double.compare canonicalises NaN: above +Infinity
  return (bits & 0x8000000000000000ULL) ? ~bits : (bits | 0x8000000000000000ULL);
}
__device__ __forceinline__ double asc_value(unsigned long long key) {
  const unsigned long long bits = (key & 0x8000000000000000ULL) ? (key & 0x7fffffffffffffffULL) : ~key;
  return __longlong_as_double((long long)bits);
}
// sortBy(-_.score): ascending Double.compare on the negated score (NaN last, +0.0 before -0.0)
__device__ __forceinline__ unsigned long long sort_key(double score) { return asc_key(-score); }

__device__ __forceinline__ bool pair_lt(unsigned long long ka, int ia, unsigned long long kb, int ib) {
  return ka < kb || (ka == kb && ia < ib);
}

// what the multi-workgroup sort orders: element i of a request is vals[i * stride] (negate: by descending score, else
// ascending), or - the sample arrays of its own recursion - a ready-made key raw[i]
struct SortSrc {
  const double *vals;
  const unsigned long long *raw;
  long long stride;
  int negate;
};
__device__ __forceinline__ unsigned long long src_key(const SortSrc &s, int i) {
  if (s.raw) return s.raw[i];
  const double v = s.vals[(long long)i * s.stride];
  return s.negate ? sort_key(v) : asc_key(v);
}

}  // namespace mrk
