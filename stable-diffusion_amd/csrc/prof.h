// Optional per-launch timing (HIP events on the launch stream) used by bench.py's roofline object.
// Off by default: ProfScope is two predictable branches when disabled.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace sdmi {

bool prof_enabled();
// `flops` = ALGORITHMIC work of the reference op this launch stands for (2 x MACs of the conv / linear /
// attention product); `flops_exec` = what the kernel executes when that differs (the 3-pass split-fp16 1x1 convs run 3x
// their algorithmic MACs), < 0 = the same.  Returns the record index.
int prof_record_begin(const char* name, double flops, double bytes, hipStream_t s, double flops_exec = -1.0);
void prof_record_end(int idx, hipStream_t s);

struct ProfScope {
  hipStream_t s; bool on; int idx = -1;
  ProfScope(const char* name, double flops, double bytes, hipStream_t stream, double flops_exec = -1.0)
      : s(stream), on(prof_enabled()) {
    if (on) idx = prof_record_begin(name, flops, bytes, s, flops_exec);
  }
  void end() { if (on && idx >= 0) { prof_record_end(idx, s); idx = -1; } }   // close early (before a follow-up launch)
  ~ProfScope() { end(); }
};

int prof_begin();
int prof_end(std::string* json);

}  // namespace sdmi
