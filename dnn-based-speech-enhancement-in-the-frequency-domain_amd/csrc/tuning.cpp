// The process-wide tuning table (tuning.h).
#include "tuning.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace sefd {
namespace {
std::mutex g_mu;
std::map<std::string, std::unique_ptr<std::string>> g_tab;     // values behind stable pointers: tune_str hands out c_str()
std::vector<std::unique_ptr<std::string>> g_retired;           // replaced values stay alive (a caller may still hold the old pointer)
bool g_init = false;

void init_locked() {
  if (g_init) return;
  g_init = true;
  const char* e = std::getenv("SEFD_TUNING");                  // the ONE environment variable of the library, read once
  if (!e) return;
  std::string s(e);
  size_t i = 0;
  while (i < s.size()) {
    size_t j = s.find_first_of(",; ", i);
    if (j == std::string::npos) j = s.size();
    const std::string kv = s.substr(i, j - i);
    const size_t eq = kv.find('=');
    if (eq != std::string::npos && eq > 0) g_tab[kv.substr(0, eq)] = std::make_unique<std::string>(kv.substr(eq + 1));
    i = j + 1;
  }
}
}  // namespace

const char* tune_str(const char* knob) {
  std::lock_guard<std::mutex> lk(g_mu);
  init_locked();
  auto it = g_tab.find(knob);
  return it == g_tab.end() ? nullptr : it->second->c_str();
}

void tune_set(const char* knob, const char* value) {
  std::lock_guard<std::mutex> lk(g_mu);
  init_locked();
  auto it = g_tab.find(knob);
  if (it != g_tab.end()) { g_retired.push_back(std::move(it->second)); g_tab.erase(it); }
  if (value) g_tab[knob] = std::make_unique<std::string>(value);
}

void tune_clear() {
  std::lock_guard<std::mutex> lk(g_mu);
  init_locked();
  for (auto& kv : g_tab) g_retired.push_back(std::move(kv.second));
  g_tab.clear();
}
}  // namespace sefd
