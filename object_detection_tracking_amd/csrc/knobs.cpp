// The one reader of the ODT_* environment overrides (knobs.hpp).
#include "knobs.hpp"

#include <cstdlib>
#include <mutex>

namespace odt {
namespace {

const char* const kNames[K_COUNT] = {
#define ODT_KNOB_NAME(n) "ODT_" #n,
    ODT_KNOB_LIST(ODT_KNOB_NAME)
#undef ODT_KNOB_NAME
};

struct Table {
  KnobVal v[K_COUNT];
  std::string text[K_COUNT];
  Table() { read(); }
  void read() {
    for (int k = 0; k < K_COUNT; ++k) {
      const char* e = getenv(kNames[k]);
      KnobVal n;
      if (e != nullptr) { n.set = true; n.i = atol(e); n.d = atof(e); n.c0 = e[0]; text[k] = e; } else text[k].clear();
      v[k] = n;
    }
  }
};

std::mutex g_mu;
Table& table() { static Table t; return t; }

}  // namespace

const KnobVal& env_knob(Knob k) { return table().v[k]; }

void knobs_reload() {
  std::lock_guard<std::mutex> lk(g_mu);
  table().read();
}

std::vector<std::string> knobs_active() {
  std::lock_guard<std::mutex> lk(g_mu);
  std::vector<std::string> out;
  const Table& t = table();
  for (int k = 0; k < K_COUNT; ++k)
    if (t.v[k].set) out.push_back(std::string(kNames[k]) + "=" + t.text[k]);
  return out;
}

}  // namespace odt
