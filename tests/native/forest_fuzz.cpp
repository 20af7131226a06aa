// CPU robustness test of the model readers (csrc/forest.cpp), built with AddressSanitizer + UBSan by
// tests/test_forest_fuzz_cpu.py.  `LambdaMARTModel` blobs come from storage (LambdaMARTRanker.scala:192-236): a corrupt or
// crafted blob must end in an error status, never in a read or write outside the parser's buffers.  Every seed model
// (LightGBM text; XGBoost JSON, UBJSON, legacy binary; the Metarank container around them) is loaded as it is - that must
// succeed - and then a few thousand times after a random mutation: flipped bytes, a truncation, a window overwritten with
// extreme integers, a digit run replaced by a hostile number, a span deleted or doubled.  What parses is also packed into
// both device images (pack_forest, pack_forest_qs) - the step that turns indices into offsets.
//   forest_fuzz <rounds> kind:path...      kind = lgbm | xgb | container | reject-<kind> = a crafted blob that must be refused (FUZZ_SEED=n: another mutation sequence)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "forest.hpp"

using namespace mrk;

static unsigned long long rng_state = 0x2545f4914f6cdd1dull;
static unsigned long long rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static size_t below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }

static std::vector<uint8_t> slurp(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(2); }
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// returns true when the blob was accepted
static bool load(const std::string &kind, const std::vector<uint8_t> &blob) {
  // the parsers get an exactly-sized heap buffer: reading one byte past the input is an ASan report
  std::vector<uint8_t> exact(blob);
  try {
    Forest f;
    if (kind == "lgbm") {
      f = parse_lightgbm_text((const char *)exact.data(), exact.size());
    } else if (kind == "xgb") {
      f = parse_xgboost(exact.data(), exact.size());
    } else {
      const Container c = parse_container(exact.data(), exact.size());
      if (c.booster_tag == 0) f = parse_lightgbm_text((const char *)c.inner, c.inner_len);
      else f = parse_xgboost(c.inner, c.inner_len);
    }
    const PackedForest pf = pack_forest(f, 24 * 1024);
    const PackedForestQS qs = pack_forest_qs(f, f.n_features);
    (void)pf;
    (void)qs;
    return true;
  } catch (const std::bad_alloc &) {
    return false;  // (mrk_model_load reports it as out of memory)
  } catch (const std::exception &) {
    return false;
  }
}

static const char *const HOSTILE[] = {"4294967295", "-1", "2147483648", "-2147483649", "1e999", "nan", "99999999999999999999", "0", "", "1e-999", "inf"};
static const uint32_t EXTREME[] = {0u, 1u, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0x00010000u, 0xfffffffeu, 0x40000000u};

static void mutate(std::vector<uint8_t> &b) {
  if (b.empty()) return;
  switch (below(7)) {
    case 0:  // flip bytes
      for (size_t k = 1 + below(4); k > 0; --k) b[below(b.size())] ^= (uint8_t)(1u << below(8));
      break;
    case 1:  // random bytes
      for (size_t k = 1 + below(4); k > 0; --k) b[below(b.size())] = (uint8_t)rnd();
      break;
    case 2:  // truncate
      b.resize(below(b.size()));
      break;
    case 3: {  // an extreme 32-bit integer, either byte order
      if (b.size() < 4) break;
      const size_t at = below(b.size() - 3);
      uint32_t v = EXTREME[below(sizeof EXTREME / sizeof *EXTREME)];
      const bool be = rnd() & 1;
      for (int i = 0; i < 4; ++i) b[at + i] = (uint8_t)(be ? v >> (24 - 8 * i) : v >> (8 * i));
      break;
    }
    case 4: {  // a digit run becomes a hostile number
      size_t at = below(b.size());
      for (size_t n = 0; n < b.size() && !(b[at] >= '0' && b[at] <= '9'); ++n) at = (at + 1) % b.size();
      if (!(b[at] >= '0' && b[at] <= '9')) break;
      size_t lo = at, hi = at;
      while (lo > 0 && ((b[lo - 1] >= '0' && b[lo - 1] <= '9') || b[lo - 1] == '.' || b[lo - 1] == '-' || b[lo - 1] == 'e')) --lo;
      while (hi < b.size() && ((b[hi] >= '0' && b[hi] <= '9') || b[hi] == '.' || b[hi] == 'e' || b[hi] == '-' || b[hi] == '+')) ++hi;
      const char *h = HOSTILE[below(sizeof HOSTILE / sizeof *HOSTILE)];
      std::vector<uint8_t> out(b.begin(), b.begin() + (long)lo);
      out.insert(out.end(), h, h + strlen(h));
      out.insert(out.end(), b.begin() + (long)hi, b.end());
      b.swap(out);
      break;
    }
    case 5: {  // delete a span
      const size_t at = below(b.size()), n = 1 + below(std::min<size_t>(64, b.size() - at));
      b.erase(b.begin() + (long)at, b.begin() + (long)(at + n));
      break;
    }
    default: {  // double a span
      const size_t at = below(b.size()), n = 1 + below(std::min<size_t>(64, b.size() - at));
      std::vector<uint8_t> span(b.begin() + (long)at, b.begin() + (long)(at + n));
      b.insert(b.begin() + (long)at, span.begin(), span.end());
      break;
    }
  }
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: forest_fuzz <rounds> kind:path...\n"); return 2; }
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
