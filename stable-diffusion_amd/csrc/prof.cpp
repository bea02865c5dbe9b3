#include "prof.h"

#include <map>
#include <sstream>
#include <vector>

#include "common.h"

namespace sdmi {
namespace {
struct Rec { std::string name; double flops, flops_exec, bytes; hipEvent_t e0, e1; };
bool g_on = false;
std::vector<Rec> g_recs;
}  // namespace

bool prof_enabled() { return g_on; }

int prof_record_begin(const char* name, double flops, double bytes, hipStream_t s, double flops_exec) {
  Rec r; r.name = name; r.flops = flops; r.flops_exec = flops_exec < 0 ? flops : flops_exec; r.bytes = bytes;
  (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1);
  (void)hipEventRecord(r.e0, s);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}
void prof_record_end(int idx, hipStream_t s) {
  if (idx >= 0 && idx < (int)g_recs.size()) (void)hipEventRecord(g_recs[idx].e1, s);
}

int prof_begin() {
  for (auto& r : g_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_recs.clear();
  g_on = true;
  return 0;
}

int prof_end(std::string* json) {
  g_on = false;
  SDMI_HIP_OK(hipDeviceSynchronize());
  struct Agg { int n = 0; double ms = 0, flops = 0, flops_exec = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : g_recs) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    Agg& a = agg[r.name];
    a.n += 1; a.ms += ms; a.flops += r.flops; a.flops_exec += r.flops_exec; a.bytes += r.bytes;
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
  }
  g_recs.clear();
  std::ostringstream os;
  os << "[";
  bool first = true;
  for (auto& kv : agg) {
    if (!first) os << ",";
    first = false;
    os << "{\"name\":\"" << kv.first << "\",\"launches\":" << kv.second.n << ",\"ms\":" << kv.second.ms
       << ",\"flops\":" << kv.second.flops << ",\"flops_exec\":" << kv.second.flops_exec << ",\"bytes\":" << kv.second.bytes << "}";
  }
  os << "]";
  *json = os.str();
  return 0;
}

}  // namespace sdmi
