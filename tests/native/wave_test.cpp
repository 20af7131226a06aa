// CPU: the wave-local helpers of the wave-per-section pre-pass (csrc/wave_device.hpp) with 64 host threads standing in
// for the 64 lanes of a wavefront.  Both places where lanes exchange anything - the ballot and the LDS ordering point - are
// reached by all lanes together in this code, so each becomes a barrier; between them the threads run freely, which is a
// WEAKER ordering than a wavefront's lock step (a result that needed lock step anywhere else would show up as a mismatch
// or a data race).  wave_median_of against a sorted-array restatement of commons-math's LEGACY percentile (bit for bit: NaN
// removed, +-0, ties, 0 .. 64 values), wave_scan_flag against a prefix sum.
#include <pthread.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static pthread_barrier_t g_bar;
static thread_local struct { unsigned x; } threadIdx;
static unsigned char g_pred[64];
static unsigned long long ballot_impl(bool p) {
  g_pred[threadIdx.x & 63] = p ? 1 : 0;
  pthread_barrier_wait(&g_bar);
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i) m |= (unsigned long long)g_pred[i] << i;
  pthread_barrier_wait(&g_bar);   // nobody overwrites its predicate before everyone has read
  return m;
}
#define __device__
#define __forceinline__ inline
#define __ballot(p) ballot_impl(p)
#define __popcll(x) __builtin_popcountll(x)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() pthread_barrier_wait(&g_bar)
static inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __dsub_rn(double a, double b) { return a - b; }
static inline double __dmul_rn(double a, double b) { return a * b; }
#include "wave_device.hpp"

static uint64_t rng_state = 0x13198a2e03707344ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

// rank_device.hpp median_of's contract: Percentile(50), LEGACY, NaNStrategy.REMOVED (DiversityFeature.scala:113-126)
static double reference(std::vector<double> v) {
  if (v.size() == 1) return v[0];
  v.erase(std::remove_if(v.begin(), v.end(), [](double x) { return x != x; }), v.end());
  std::stable_sort(v.begin(), v.end());
  const int m = (int)v.size();
  if (m <= 0) return std::nan("");
  const double pos = 0.5 * (m + 1), fpos = std::floor(pos), dif = pos - fpos;
  if (pos < 1.0) return v[0];
  if (pos >= m) return v[m - 1];
  return v[(int)fpos - 1] + dif * (v[(int)fpos] - v[(int)fpos - 1]);
}

struct Case { std::vector<double> vals; std::vector<unsigned char> flags; };
static std::vector<Case> g_cases;
static double g_lds[64];
static double g_got[64];              // per lane: every lane must return the same median
static int g_excl[64], g_total[64];
static long long g_bad = 0;

static void *lane_main(void *arg) {
  threadIdx.x = (unsigned)(uintptr_t)arg;
  const int lane = (int)threadIdx.x;
  for (size_t c = 0; c < g_cases.size(); ++c) {
    const Case &cs = g_cases[c];
    const int n = (int)cs.vals.size();
    if (lane < n) g_lds[lane] = cs.vals[(size_t)lane];   // what the scan loop leaves in LDS: value k at s_vals[k]
    pthread_barrier_wait(&g_bar);
    g_got[lane] = mrk::wave_median_of(g_lds, n);
    int total = -1;
    g_excl[lane] = mrk::wave_scan_flag(cs.flags[(size_t)lane] != 0, total);
    g_total[lane] = total;
    pthread_barrier_wait(&g_bar);
    if (lane == 0) {
      const double exp = reference(cs.vals);
      int run = 0;
      for (int l = 0; l < 64; ++l) {
        if (memcmp(&g_got[l], &exp, 8) != 0 && !(g_got[l] != g_got[l] && exp != exp)) {
          if (++g_bad < 10) printf("case %zu (n %d) lane %d: median %.17g expected %.17g\n", c, n, l, g_got[l], exp);
        }
        if (g_excl[l] != run) { if (++g_bad < 10) printf("case %zu lane %d: scan %d expected %d\n", c, l, g_excl[l], run); }
        run += cs.flags[(size_t)l] ? 1 : 0;
      }
      for (int l = 0; l < 64; ++l) if (g_total[l] != run) { if (++g_bad < 10) printf("case %zu lane %d: total %d expected %d\n", c, l, g_total[l], run); }
    }
    pthread_barrier_wait(&g_bar);
  }
  return nullptr;
}

int main() {
  for (int n = 0; n <= 64; ++n)
    for (int rep = 0; rep < 12; ++rep) {
      Case cs;
      for (int i = 0; i < n; ++i) {
        const uint64_t r = rnd();
        double v = (double)(int64_t)(r % 2001) / 8.0 - 125.0;            // plenty of ties
        if (rep % 3 == 1 && r % 7 == 0) v = std::nan("");
        if (rep % 4 == 2 && r % 5 == 0) v = (r & 64) ? 0.0 : -0.0;
        if (rep == 11) v = std::nan("");                                   // nothing but NaN
        if (rep == 10) v = 3.25;                                           // all equal
        if (rep == 9 && r % 3 == 0) v = (r & 1) ? INFINITY : -INFINITY;
        cs.vals.push_back(v);
      }
      for (int l = 0; l < 64; ++l) cs.flags.push_back(rep == 0 ? 1 : rep == 1 ? 0 : (unsigned char)(rnd() & 1));
      g_cases.push_back(cs);
    }
  pthread_barrier_init(&g_bar, nullptr, 64);
  pthread_t th[64];
  for (uintptr_t l = 0; l < 64; ++l) pthread_create(&th[l], nullptr, lane_main, (void *)l);
  for (int l = 0; l < 64; ++l) pthread_join(th[l], nullptr);
  printf("%zu cases, %lld bad\n", g_cases.size(), g_bad);
  return g_bad ? 1 : 0;
}
