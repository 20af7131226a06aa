// CPU robustness test of the other byte formats the library reads from outside (tests/test_bytes_fuzz_cpu.py builds it
// with AddressSanitizer + UBSan; the sources under test are compiled INTO this binary with the sanitizers, everything they
// call is taken from libmrk_hip.so):
//   ckpt     encoder checkpoints - safetensors and ONNX protobuf (csrc/weights.cpp read_checkpoint)
//   request  the reference's binary RankingEventFormat (csrc/codec.cpp DecodedRequest::decode)
//   fv       the binary FeatureValue stream (csrc/codec.cpp load_feature_values) into a store mirror (no device)
//   tok      a HuggingFace tokenizer.json (csrc/tokenizer.cpp), then a few texts through whatever loaded (arbitrary bytes too)
//   config   the feature / model configuration (csrc/features.cpp load_config + json.hpp)
// Every seed must be accepted as it is; its mutants must end in an error or a clean load - never outside a buffer.
//   bytes_fuzz <rounds> kind:path...        (FUZZ_SEED=n: another mutation sequence)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "encoder.hpp"
#include "features.hpp"
#include "store.hpp"
#include "tokenizer.hpp"

using namespace mrk;

namespace mrk {
int load_feature_values(Store &store, const uint8_t *bytes, size_t len, int64_t now_ms);  // codec.cpp
}

static unsigned long long rng_state = 0x2545f4914f6cdd1dull;
static unsigned long long rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static size_t below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }
// a position in a blob of n bytes: half of the time inside its first or last 2 KB, where the big formats keep their structure
static size_t pos(size_t n) {
  const unsigned long long r = rnd() & 3;
  if (n > 4096 && r == 0) return below(2048);
  if (n > 4096 && r == 1) return n - 2048 + below(2048);
  return below(n);
}

static std::vector<uint8_t> slurp(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(2); }
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static const char *CONFIG = R"({"features": [
    {"name": "genre", "type": "string", "scope": "item", "source": "item.genre", "values": ["a", "b", "c"], "encode": "index"},
    {"name": "tags", "type": "string", "scope": "item", "source": "item.tags", "values": ["a", "b"], "encode": "onehot"},
    {"name": "vec", "type": "vector", "scope": "item", "source": "item.vec", "reduce": ["vector8"]},
    {"name": "pop", "type": "number", "scope": "item", "source": "item.pop"},
    {"name": "clicks", "type": "interaction_count", "scope": "item", "interaction": "click"},
    {"name": "ctr", "type": "rate", "scope": "item", "top": "click", "bottom": "impression", "bucket": "1d", "periods": [7, 30]},
    {"name": "profile", "type": "interacted_with", "scope": "session", "interaction": "click", "field": ["item.genres", "item.actors"]}],
    "models": {"m": {"type": "lambdamart", "features": ["genre", "tags", "vec", "pop", "clicks", "ctr", "profile"]}}})";

static bool load(const std::string &kind, const std::vector<uint8_t> &blob) {
  std::vector<uint8_t> exact(blob);  // exactly-sized heap buffer: one byte past the input is an ASan report
  try {
    if (kind == "ckpt") {
      const Checkpoint c = read_checkpoint(exact.data(), exact.size());
      (void)c;
    } else if (kind == "request") {
      size_t at = 0;
      while (at < exact.size()) {
        DecodedRequest d;
        const size_t used = d.decode(exact.data() + at, exact.size() - at);
        if (used == 0) break;
        // what the C ABI hands on: every pointer of the request must be readable
        size_t sum = strlen(d.req.id ? d.req.id : "");
        for (int i = 0; i < d.req.n_items; ++i) sum += strlen(d.req.item_ids[i]);
        (void)sum;
        at += used;
      }
    } else if (kind == "tok") {
      const Tokenizer t = Tokenizer::from_json((const char *)exact.data(), exact.size());
      const std::string second = "caf\xc3\xa9 \xe4\xb8\xad\xe6\x96\x87 \xff\xfe broken";
      for (const char *text : {"red socks", "", "Un\xcc\x81 texte accentu\xc3\xa9 [SEP] [MASK]", "\xf0\x9f\x98\x80\xf0\x9f\x98\x80 \xe1\x84\x80\xe1\x85\xa1\xe1\x86\xa8"}) {
        (void)t.encode(text, nullptr);
        (void)t.encode(text, &second);
      }
    } else if (kind == "config") {
      Store st;
      std::unique_ptr<Registry> reg = load_config((const char *)exact.data(), exact.size(), st, false);
    } else {
      Store st;
      std::unique_ptr<Registry> reg = load_config(CONFIG, strlen(CONFIG), st, false);
      (void)load_feature_values(st, exact.data(), exact.size(), (int64_t)(exact.size() % 3 == 0 ? 1700000000000ll : -1));   // with and without ttl tracking
    }
    return true;
  } catch (const std::bad_alloc &) {
    return false;
  } catch (const std::exception &) {
    return false;
  }
}

static const uint32_t EXTREME[] = {0u, 1u, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0x00010000u, 0xfffffffeu, 0x40000000u};

static void mutate(std::vector<uint8_t> &b) {
  if (b.empty()) return;
  switch (below(8)) {
    case 0:
      for (size_t k = 1 + below(4); k > 0; --k) b[pos(b.size())] ^= (uint8_t)(1u << below(8));
      break;
    case 1:
      for (size_t k = 1 + below(4); k > 0; --k) b[pos(b.size())] = (uint8_t)rnd();
      break;
    case 2:
      b.resize(below(b.size()));
      break;
    case 3: {  // an extreme 32-bit integer, either byte order
      if (b.size() < 4) break;
      const size_t at = pos(b.size() - 3);
      const uint32_t v = EXTREME[below(sizeof EXTREME / sizeof *EXTREME)];
      const bool be = rnd() & 1;
      for (int i = 0; i < 4; ++i) b[at + i] = (uint8_t)(be ? v >> (24 - 8 * i) : v >> (8 * i));
      break;
    }
    case 4: {  // a run of 0xff (varints that never end, lengths of 2^64 - 1)
      const size_t at = pos(b.size()), n = 1 + below(std::min<size_t>(12, b.size() - at));
      memset(b.data() + at, 0xff, n);
      break;
    }
    case 5: {  // delete a span
      const size_t at = pos(b.size()), n = 1 + below(std::min<size_t>(64, b.size() - at));
      b.erase(b.begin() + (long)at, b.begin() + (long)(at + n));
      break;
    }
    case 6: {  // (text formats) a digit run becomes a hostile number
      static const char *const HOSTILE[] = {"4294967295", "-1", "2147483648", "-2147483649", "1e999", "99999999999999999999", "0", "1e-999", "0.5", "1048576"};
      size_t at = below(b.size());
      for (size_t n = 0; n < b.size() && !(b[at] >= '0' && b[at] <= '9'); ++n) at = (at + 1) % b.size();
      if (!(b[at] >= '0' && b[at] <= '9')) break;
      size_t lo = at, hi = at;
      while (lo > 0 && ((b[lo - 1] >= '0' && b[lo - 1] <= '9') || b[lo - 1] == '.' || b[lo - 1] == '-')) --lo;
      while (hi < b.size() && ((b[hi] >= '0' && b[hi] <= '9') || b[hi] == '.' || b[hi] == 'e')) ++hi;
      const char *h = HOSTILE[below(sizeof HOSTILE / sizeof *HOSTILE)];
      std::vector<uint8_t> out(b.begin(), b.begin() + (long)lo);
      out.insert(out.end(), h, h + strlen(h));
      out.insert(out.end(), b.begin() + (long)hi, b.end());
      b.swap(out);
      break;
    }
    default: {  // double a span
      const size_t at = pos(b.size()), n = 1 + below(std::min<size_t>(64, b.size() - at));
      const std::vector<uint8_t> span(b.begin() + (long)at, b.begin() + (long)(at + n));
      b.insert(b.begin() + (long)at, span.begin(), span.end());
      break;
    }
  }
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: bytes_fuzz <rounds> kind:path...\n"); return 2; }
  const int rounds = atoi(argv[1]);
  if (const char *seed = getenv("FUZZ_SEED")) rng_state ^= strtoull(seed, nullptr, 10) * 0x9e3779b97f4a7c15ull;
  long total = 0, accepted = 0;
  for (int a = 2; a < argc; ++a) {
    const std::string arg = argv[a];
    const size_t colon = arg.find(':');
    std::string kind = arg.substr(0, colon);
    const std::string path = arg.substr(colon + 1);
    const std::vector<uint8_t> seed = slurp(path);
    if (kind.rfind("reject-", 0) == 0) {  // a crafted blob (one of the defects the mutation runs found): must end in an error
      kind = kind.substr(7);
      if (load(kind, seed)) { printf("crafted blob %s was ACCEPTED\n", arg.c_str()); return 1; }
      printf("%s: rejected\n", arg.c_str());
      continue;
    }
    if (!load(kind, seed)) { printf("seed %s was rejected\n", arg.c_str()); return 1; }
    long ok = 0;
    for (int r = 0; r < rounds; ++r) {
      std::vector<uint8_t> b(seed);
      for (size_t k = 1 + below(3); k > 0; --k) mutate(b);
      ok += load(kind, b) ? 1 : 0;
    }
    printf("%s: %zu bytes, %d mutants, %ld still accepted\n", arg.c_str(), seed.size(), rounds, ok);
    total += rounds;
    accepted += ok;
  }
  printf("survived %ld mutants (%ld accepted)\n", total, accepted);
  return 0;
}
