// CPU: numeric `diversity` over more values than the device pre-pass sorts (csrc/features.cpp resolve_requests: the host
// takes the commons-math LEGACY median from its mirror and hands it over as PrepOut.preset).  Reads "<item id> <value|nan|->"
// lines (- = no state), puts the values, resolves ONE request of all the items with `top` = argv[2] and prints
// preset / mode / scalar of the feature's pre-pass entry.  features.cpp + store.cpp compiled INTO this binary; no device.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "features.hpp"
#include "store.hpp"

using namespace mrk;

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  const std::string cfg = std::string(R"({"features": [{"name": "div", "type": "diversity", "source": "item.price", "top": )") + argv[2] +
                          R"(}], "models": {"m": {"type": "lambdamart", "features": ["div"]}}})";
  Store st;
  std::unique_ptr<Registry> reg = load_config(cfg.c_str(), cfg.size(), st, false);
  std::ifstream in(argv[1]);
  std::vector<std::string> ids;
  std::string id, val;
  while (in >> id >> val) {
    ids.push_back(id);
    if (val == "-") continue;
    st.put_double(("item=" + id + "/div").c_str(), val == "nan" ? std::nan("") : strtod(val.c_str(), nullptr));
  }
  std::vector<const char *> ptrs;
  for (auto &s : ids) ptrs.push_back(s.c_str());
  mrk_request q;
  memset(&q, 0, sizeof q);
  q.id = "r";
  q.timestamp_ms = 1661345221008LL;
  q.n_items = (int)ptrs.size();
  q.item_ids = ptrs.data();
  HostBatch hb;
  resolve_requests(*reg->program("m"), st, &q, 1, nullptr, hb);
  const PrepOut &po = hb.prep_out.at(0);
  printf("preset %d mode %d scalar %.17g max_doubles %d\n", (int)po.preset, (int)po.mode, po.scalar, hb.max_doubles);
  return 0;
}
