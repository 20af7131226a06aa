// Host-only timing of resolve_requests() (csrc/features.cpp): see tools/host_bench.py.  Links against libmrk_hip.so's
// internal symbols; no device call is made (the programs are not uploaded, the store is never flushed).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "features.hpp"
#include "store.hpp"

using namespace mrk;

static std::vector<std::string> split_tabs(const std::string &s, size_t from) {
  std::vector<std::string> out;
  size_t at = from;
  while (true) {
    size_t t = s.find('\t', at);
    out.push_back(s.substr(at, t == std::string::npos ? std::string::npos : t - at));
    if (t == std::string::npos) break;
    at = t + 1;
  }
  return out;
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: host_bench <dump> <n_requests> [threads]\n"); return 2; }
  const int n_req = atoi(argv[2]);
  if (argc > 3) setenv("MRK_HOST_THREADS", argv[3], 1);
  std::ifstream in(argv[1]);
  std::string line;
  Store store;
  std::unique_ptr<Registry> reg;
  struct Ev { std::string id, user, session; long long ts; std::vector<std::string> items; std::vector<const char *> ptrs; };
  std::vector<Ev> evs;
  size_t puts = 0;
  double put_s = 0;
  while (std::getline(in, line)) {
    if (line.size() < 2) continue;
    if (line[0] == 'C') {
      reg = load_config(line.data() + 2, line.size() - 2, store, false);
    } else if (line[0] == 'P') {
      std::vector<std::string> f = split_tabs(line, 2);
      const std::string &kind = f[0];
      const char *key = f[1].c_str();
      ++puts;
      const auto load_s = std::chrono::steady_clock::now();
      if (kind == "double") store.put_double(key, atof(f[2].c_str()));
      else if (kind == "string") store.put_string(key, f[2].c_str());
      else if (kind == "counter") store.put_counter(key, atoll(f[2].c_str()));
      else if (kind == "string_list" || kind == "bounded_list") {
        std::vector<const char *> p;
        for (size_t i = 2; i < f.size(); ++i) if (!f[i].empty() || f.size() > 3) p.push_back(f[i].c_str());
        if (kind == "string_list") store.put_string_list(key, p.data(), (int)p.size());
        else store.put_bounded_list(key, p.data(), (int)p.size());
      } else if (kind == "double_list") {
        std::vector<double> v;
        for (size_t i = 2; i < f.size(); ++i) v.push_back(atof(f[i].c_str()));
        store.put_double_list(key, v.data(), (int)v.size());
      } else if (kind == "periodic") {
        std::vector<int64_t> v;
        for (size_t i = 2; i < f.size(); ++i) v.push_back(atoll(f[i].c_str()));
        store.put_periodic(key, v.data(), (int)v.size());
      }
      put_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - load_s).count();
    } else if (line[0] == 'R' && (int)evs.size() < n_req) {
      std::vector<std::string> f = split_tabs(line, 2);
      Ev e;
      e.id = f[0]; e.user = f[1]; e.session = f[2]; e.ts = atoll(f[3].c_str());
      e.items.assign(f.begin() + 4, f.end());
      evs.push_back(std::move(e));
    }
  }
  std::vector<mrk_request> reqs(evs.size());
  size_t total = 0;
  for (size_t r = 0; r < evs.size(); ++r) {
    Ev &e = evs[r];
    for (auto &s : e.items) e.ptrs.push_back(s.c_str());
    mrk_request q;
    memset(&q, 0, sizeof q);
    q.id = e.id.c_str(); q.user = e.user.c_str(); q.session = e.session.c_str(); q.timestamp_ms = e.ts;
    q.n_items = (int)e.ptrs.size(); q.item_ids = e.ptrs.data();
    reqs[r] = q;
    total += e.ptrs.size();
  }
  const Program *prog = reg->program("xgboost");
  fprintf(stderr, "%zu puts in %.2f s (%.2f M puts/s incl. argument conversion), %zu requests, %zu items\n", puts, put_s, puts / put_s / 1e6, reqs.size(), total);
  HostBatch hb;
  const int reps = 5;
  {
    // the serving loop's form of the same batch: flat id bytes, slots resolved by the device (mrk_batch_load with
    // mrk_item_ids) - what is left for the host is O(requests)
    std::string bytes;
    std::vector<uint32_t> offs(1, 0);
    for (const Ev &e : evs)
      for (const std::string &s : e.items) { bytes += s; offs.push_back((uint32_t)bytes.size()); }
    std::vector<mrk_request> flat(reqs);
    for (mrk_request &q : flat) q.item_ids = nullptr;
    mrk_item_ids ids{(const uint8_t *)bytes.data(), offs.data(), bytes.size()};
    resolve_requests(*prog, store, flat.data(), (int)flat.size(), &ids, hb);
    double fbest = 1e30;
    for (int i = 0; i < reps; ++i) {
      auto t0 = std::chrono::steady_clock::now();
      resolve_requests(*prog, store, flat.data(), (int)flat.size(), &ids, hb);
      fbest = std::min(fbest, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("flat ids (device-resolved): %.3f ms per batch of %zu requests -> %.1f M items/s of host work; %zu id bytes; arena %llu entries, max per request %llu\n",
           fbest * 1e3, reqs.size(), total / fbest / 1e6, bytes.size(), (unsigned long long)hb.arena_entries, (unsigned long long)hb.max_req_entries);
  }
  resolve_requests(*prog, store, reqs.data(), (int)reqs.size(), nullptr, hb);  // warm
  double best = 1e30;
  for (int i = 0; i < reps; ++i) {
    auto t0 = std::chrono::steady_clock::now();
    resolve_requests(*prog, store, reqs.data(), (int)reqs.size(), nullptr, hb);
    best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  printf("resolve_requests: %.1f ms per batch of %zu requests (%zu items) -> %.2f M items/s; arena %llu entries, max per request %llu\n",
         best * 1e3, reqs.size(), total, total / best / 1e6, (unsigned long long)hb.arena_entries, (unsigned long long)hb.max_req_entries);
  // a checksum of what the device would receive, to compare runs
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t n) { const unsigned char *c = (const unsigned char *)p; for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 1099511628211ull; };
  mix(hb.item_slot.data(), hb.item_slot.size() * 4);
  mix(hb.reqs.data(), hb.reqs.size() * sizeof(ReqDev));
  for (auto &po : hb.prep_out) { mix(&po.tab_off, 4); mix(&po.tab_cap, 4); }
  mix(hb.consts.data(), hb.consts.size() * 8);
  printf("checksum %016llx\n", h);
  return 0;
}
