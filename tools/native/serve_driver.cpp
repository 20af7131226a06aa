// A native host of the serving loop, written against include/mrk.h ONLY (what a JVM / Go / C++ host links): the end-to-end
// leg of bench.py without Python between the calls.  Each host thread owns `batches_in_flight` batches and cycles
//   mrk_batch_host_outputs (results of this batch's previous round) -> mrk_batch_load (fresh requests: id bytes uploaded and
//   resolved on the device) -> mrk_batch_run -> mrk_batch_enqueue_fetch
// over `n_sets` pre-marshalled request sets (mrk_request arrays + flat id bytes; marshalling JSON into those structs is the
// host's HTTP layer and is not timed, exactly as in the Python loop it replaces).  Reference: the request loop around
// Ranker.rerank, main/command/Serve.scala:130-150, api/routes/RankApi.scala:25-41.
//   g++ -O2 -shared -fPIC -std=c++17 -I include tools/native/serve_driver.cpp -o tools/native/libserve_driver.so -L metarank_amd -lmrk_hip -pthread
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "mrk.h"

namespace {
using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct Lane {
  std::vector<mrk_batch *> batches;
  long long done = 0;
  int failed = 0;      // first failing status code of an mrk_* call or of a request
  double t[4] = {0, 0, 0, 0};   // seconds in: waiting for results, mrk_batch_load, mrk_batch_run, mrk_batch_enqueue_fetch
};
}  // namespace

extern "C" {

// out[0] elapsed seconds of the timed region, out[1] device batches completed, out[2..5] host seconds per phase summed over the
// threads (wait_results, load, run, enqueue_fetch), out[6] first error code (0 = none), out[7] threads.
// first_scores / first_order (nullable, total_items of set 0): what the loop returned for request set 0 during warm-up.
int mrk_bench_serve_loop(mrk_ctx *ctx, mrk_model *model, const char *model_name, const mrk_request *const *sets, const int *set_n_req,
                         const mrk_item_ids *set_ids, int n_sets, int n_threads, int batches_in_flight, double seconds, double *out,
                         double *first_scores, int32_t *first_order, long long first_items) {
  if (!ctx || !model_name || !sets || !set_n_req || !set_ids || n_sets < 1 || n_threads < 1 || batches_in_flight < 1 || !out) return MRK_ERR_INVALID_ARG;
  std::vector<Lane> lanes((size_t)n_threads);
  for (Lane &ln : lanes)
    for (int i = 0; i < batches_in_flight; ++i) {
      mrk_batch *b = nullptr;
      const int rc = mrk_batch_create(ctx, &b);
      if (rc != MRK_OK) return rc;
      ln.batches.push_back(b);
    }
  auto serve = [&](Lane &ln, int k, long long i, bool keep) {
    mrk_batch *b = ln.batches[(size_t)(i % batches_in_flight)];
    const auto t0 = clk::now();
    if (i >= batches_in_flight) {
      const double *s = nullptr;
      const int32_t *o = nullptr, *st = nullptr;
      const int rc = mrk_batch_host_outputs(b, &s, &o, &st);
      if (rc != MRK_OK && !ln.failed) ln.failed = rc;
      const int prev_set = (int)((((i - batches_in_flight) * n_threads) + k) % n_sets);
      if (rc == MRK_OK) {
        for (int r = 0; r < set_n_req[prev_set]; ++r)
          if (st[r] != 0 && !ln.failed) ln.failed = st[r];
        if (keep && prev_set == 0 && first_scores && first_order) {
          memcpy(first_scores, s, (size_t)first_items * sizeof(double));
          memcpy(first_order, o, (size_t)first_items * sizeof(int32_t));
        }
      }
    }
    const auto t1 = clk::now();
    const int set = (int)(((i * n_threads) + k) % n_sets);
    int rc = mrk_batch_load(b, model_name, sets[set], set_n_req[set], &set_ids[set]);
    const auto t2 = clk::now();
    if (rc == MRK_OK) rc = mrk_batch_run(b, model);
    const auto t3 = clk::now();
    if (rc == MRK_OK) rc = mrk_batch_enqueue_fetch(b);
    const auto t4 = clk::now();
    if (rc != MRK_OK && !ln.failed) ln.failed = rc;
    ln.t[0] += secs(t0, t1); ln.t[1] += secs(t1, t2); ln.t[2] += secs(t2, t3); ln.t[3] += secs(t3, t4);
  };
  // warm-up on the calling thread, lane by lane: every batch has run, set 0's results are kept
  for (int k = 0; k < n_threads; ++k) {
    for (long long i = 0; i < 2LL * batches_in_flight + n_sets; ++i) serve(lanes[(size_t)k], k, i, true);
    for (mrk_batch *b : lanes[(size_t)k].batches) (void)mrk_batch_sync(b);
  }
  std::atomic<int> go{0};
  std::vector<std::thread> th;
  clk::time_point t_start;
  for (int k = 0; k < n_threads; ++k)
    th.emplace_back([&, k] {
      Lane &ln = lanes[(size_t)k];
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      for (double &x : ln.t) x = 0;
      const auto t = clk::now();
      long long i = 2LL * batches_in_flight + n_sets;   // continue the numbering of the warm-up: every batch has results pending
      for (;;) {
        for (int u = 0; u < 16; ++u) { serve(ln, k, i, false); ++i; ++ln.done; }
        if (secs(t, clk::now()) >= seconds || ln.failed) break;
      }
      for (mrk_batch *b : ln.batches) {
        const double *s; const int32_t *o, *st;
        (void)mrk_batch_host_outputs(b, &s, &o, &st);
      }
    });
  t_start = clk::now();
  go.store(1, std::memory_order_release);
  for (std::thread &t : th) t.join();
  const double elapsed = secs(t_start, clk::now());
  memset(out, 0, 8 * sizeof(double));
  out[0] = elapsed;
  for (Lane &ln : lanes) {
    out[1] += (double)ln.done;
    for (int j = 0; j < 4; ++j) out[2 + j] += ln.t[j];
    if (ln.failed && out[6] == 0) out[6] = ln.failed;
    for (mrk_batch *b : ln.batches) mrk_batch_free(b);
  }
  out[7] = n_threads;
  return MRK_OK;
}

}  // extern "C"
