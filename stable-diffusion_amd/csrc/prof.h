// Optional per-launch timing (HIP events on the launch stream) used by bench.py's roofline object.
// Off by default: ProfScope is two predictable branches when disabled.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace sdmi {

bool prof_enabled();
void prof_record_begin(const char* name, double flops, double bytes, hipStream_t s);
void prof_record_end(hipStream_t s);

struct ProfScope {
  hipStream_t s; bool on;
  ProfScope(const char* name, double flops, double bytes, hipStream_t stream) : s(stream), on(prof_enabled()) {
    if (on) prof_record_begin(name, flops, bytes, s);
  }
  ~ProfScope() { if (on) prof_record_end(s); }
};

int prof_begin();
int prof_end(std::string* json);

}  // namespace sdmi
