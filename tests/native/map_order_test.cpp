// CPU: the order of an interacted_with feature's columns when it has more than 4 fields - the iteration order of the
// reference's Scala immutable Map (csrc/features.cpp scala_map_key_order).  Prints the order of the keys given as
// arguments, one per line; with "--config" first, loads a config whose `profile` feature has those fields and prints the
// order the feature registry settled on.  features.cpp is compiled INTO this binary; no device.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "features.hpp"
#include "store.hpp"

using namespace mrk;

int main(int argc, char **argv) {
  bool config = argc > 1 && !strcmp(argv[1], "--config");
  std::vector<std::string> keys;
  for (int i = config ? 2 : 1; i < argc; ++i) keys.push_back(argv[i]);
  if (!config) {
    for (auto &k : scala_map_key_order(keys)) printf("%s\n", k.c_str());
    return 0;
  }
  std::string cfg = R"({"features": [{"name": "profile", "type": "interacted_with", "scope": "session", "interaction": "click", "field": [)";
  for (size_t i = 0; i < keys.size(); ++i) cfg += std::string(i ? "," : "") + "\"item." + keys[i] + "\"";
  cfg += R"(]}], "models": {"m": {"type": "lambdamart", "features": ["profile"]}}})";
  Store st;
  std::unique_ptr<Registry> reg = load_config(cfg.c_str(), cfg.size(), st, false);
  for (auto &f : reg->features)
    if (f->name == "profile")
      for (auto &v : f->values) printf("%s\n", v.c_str());
  return 0;
}
