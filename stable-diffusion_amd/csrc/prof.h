// Optional per-launch timing (HIP events on the launch stream) used by bench.py's roofline object.
// Off by default: ProfScope is two predictable branches when disabled.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace sdmi {

bool prof_enabled();
int prof_record_begin(const char* name, double flops, double bytes, hipStream_t s);   // returns the record index
void prof_record_end(int idx, hipStream_t s);

struct ProfScope {
  hipStream_t s; bool on; int idx = -1;
  ProfScope(const char* name, double flops, double bytes, hipStream_t stream) : s(stream), on(prof_enabled()) {
    if (on) idx = prof_record_begin(name, flops, bytes, s);
  }
  void end() { if (on && idx >= 0) { prof_record_end(idx, s); idx = -1; } }   // close early (before a follow-up launch)
  ~ProfScope() { end(); }
};

int prof_begin();
int prof_end(std::string* json);

}  // namespace sdmi
