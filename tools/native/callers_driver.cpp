// Closed-loop concurrent callers of the per-request entry points, written against include/mrk.h ONLY: T native threads, each
// calling mrk_rank (or mrk_serve_rank) back to back on pre-marshalled mrk_request structs - the reference's serving model
// (one Ranker.rerank per request fiber on the cats-effect pool, api/routes/RankApi.scala:25-41, ml/Ranker.scala:27-83) without
// an interpreter between the calls.  (tools/concurrent_bench.py drove Python threads until round 5: its numbers at 16+ threads
// were the GIL's - every call re-enters the interpreter - not the library's.)
//   g++ -O2 -shared -fPIC -std=c++17 -I include tools/native/callers_driver.cpp -o tools/native/libcallers_driver.so -L metarank_amd -lmrk_hip -pthread
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "mrk.h"

namespace {
using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }
}  // namespace

// A tool, not the product: a SIGSEGV of the process that loaded this driver prints the native stack before it dies (the
// callers tool was seen to die AFTER its last row, during process exit, where Python's faulthandler is already off).
namespace {
void segv_backtrace(int sig) {
  void *frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "\n== callers_driver: fatal signal, native stack:\n";
  (void)!write(2, msg, sizeof msg - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
struct InstallSegv {
  InstallSegv() {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = segv_backtrace;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
    sigaction(SIGABRT, &sa, nullptr);
  }
} install_segv;
}  // namespace

extern "C" {

// reqs[n_reqs]: the requests the callers cycle through (thread t starts at t * 8); max_items: the largest n_items.
// srv == nullptr: mrk_rank(ctx, model, model_name, ...); else mrk_serve_rank(srv, ...).
// lat_ms[threads * per_thread]: every call's latency; out[0] = elapsed seconds of the timed region, out[1] = first error code,
// out[2] = calls made.  check_scores / check_order (nullable, n_reqs x max_items, filled by the caller from a sequential pass):
// every concurrent result is compared with them bit for bit; out[3] = number of calls whose result differed.
int mrk_bench_callers(mrk_ctx *ctx, mrk_model *model, const char *model_name, mrk_server *srv, const mrk_request *reqs, int n_reqs,
                      int max_items, int threads, int per_thread, double *lat_ms, double *out, const double *check_scores,
                      const int32_t *check_order) {
  if (!reqs || n_reqs < 1 || threads < 1 || per_thread < 1 || !lat_ms || !out || max_items < 1) return MRK_ERR_INVALID_ARG;
  std::atomic<int> go{0}, first_err{0};
  std::atomic<long long> wrong{0};
  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t)
    th.emplace_back([&, t] {
      std::vector<double> scores((size_t)max_items);
      std::vector<int32_t> order((size_t)max_items);
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      for (int k = 0; k < per_thread; ++k) {
        const int ri = (t * 8 + k) % n_reqs;
        const mrk_request *r = &reqs[ri];
        const auto t0 = clk::now();
        const int rc = srv ? mrk_serve_rank(srv, r, scores.data(), order.data()) : mrk_rank(ctx, model, model_name, r, scores.data(), order.data(), nullptr);
        lat_ms[(size_t)t * per_thread + k] = secs(t0, clk::now()) * 1e3;
        if (rc != MRK_OK) {
          int z = 0;
          first_err.compare_exchange_strong(z, rc);
        } else if (check_scores && check_order) {
          const size_t n = (size_t)r->n_items;
          if (memcmp(scores.data(), check_scores + (size_t)ri * max_items, n * 8) != 0 ||
              memcmp(order.data(), check_order + (size_t)ri * max_items, n * 4) != 0)
            wrong.fetch_add(1);
        }
      }
    });
  const auto t_start = clk::now();
  go.store(1, std::memory_order_release);
  for (std::thread &t : th) t.join();
  out[0] = secs(t_start, clk::now());
  out[1] = first_err.load();
  out[2] = (double)threads * per_thread;
  out[3] = (double)wrong.load();
  return MRK_OK;
}

}  // extern "C"
