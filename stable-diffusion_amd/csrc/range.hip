// fp16 range guard (debug): counts fp16 activation values near / beyond the fp16 range in every buffer the kernels
// write as an MFMA operand.  The synthetic weights of the test / bench environment keep activations O(1); a real
// checkpoint has outlier channels (SURVEY.md 7 "hard parts" iv).  The reference runs the same tensors through
// torch.autocast fp16, so it has the same exposure -- this mode makes it visible per launch instead of as NaN images.
//   SDMI_CHECK_RANGE=1 (or sdmi_range_check(1)): after every launch that writes fp16 activations, scan them;
//   sdmi_range_report(): {"over_6e4": n, "nonfinite": m, "max_abs": x, "first": "<kernel class of the first offender>"}.
// The scan synchronises the stream after each launch: a debugging mode, never on in the timed path.
#include "common.h"

#include <string.h>

#include <atomic>
#include <mutex>
#include <string>

namespace sdmi {
namespace {

struct RangeState {
  std::mutex mu;
  int enabled = -1;                       // -1: read SDMI_CHECK_RANGE on first use
  unsigned long long* dev = nullptr;      // [over, nonfinite, max |x| bits (as float, via atomicMax on the uint)]
  unsigned long long total[3] = {0, 0, 0};
  std::string first;
};
RangeState g_range;

__global__ void __launch_bounds__(256) range_scan_kernel(const f16* p, int64_t n8, int64_t n, unsigned long long* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned over = 0, bad = 0;
  float mx = 0.f;
  if (i < n8) {
    const f16x8 v = *(const f16x8*)(p + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = fabsf((float)v[j]);
      if (!(a <= 65504.f)) ++bad;                  // inf or NaN
      else { if (a > 60000.f) ++over; mx = fmaxf(mx, a); }
    }
  } else if (i == n8) {                            // tail (n not a multiple of 8)
    for (int64_t k = n8 * 8; k < n; ++k) {
      const float a = fabsf((float)p[k]);
      if (!(a <= 65504.f)) ++bad; else { if (a > 60000.f) ++over; mx = fmaxf(mx, a); }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { over += __shfl_xor(over, o); bad += __shfl_xor(bad, o); mx = fmaxf(mx, __shfl_xor(mx, o)); }
  if ((threadIdx.x & 63) == 0) {
    if (over) atomicAdd(out, (unsigned long long)over);
    if (bad) atomicAdd(out + 1, (unsigned long long)bad);
    atomicMax(out + 2, (unsigned long long)__float_as_uint(mx));     // non-negative floats order like their bit patterns
  }
}

}  // namespace

bool range_check_enabled() {
  if (g_range.enabled < 0) {
    const char* e = getenv("SDMI_CHECK_RANGE");
    g_range.enabled = (e && atoi(e) != 0) ? 1 : 0;
  }
  return g_range.enabled == 1;
}

int range_check_set(int enable) {
  std::lock_guard<std::mutex> lk(g_range.mu);
  g_range.enabled = enable ? 1 : 0;
  g_range.total[0] = g_range.total[1] = g_range.total[2] = 0;
  g_range.first.clear();
  return 0;
}

int range_scan(const char* what, const f16* p, int64_t n, hipStream_t stream) {
  if (!p || n <= 0) return 0;
  std::lock_guard<std::mutex> lk(g_range.mu);
  if (!g_range.dev) SDMI_HIP_OK(hipMalloc((void**)&g_range.dev, 3 * sizeof(unsigned long long)));
  SDMI_HIP_OK(hipMemsetAsync(g_range.dev, 0, 3 * sizeof(unsigned long long), stream));
  const int64_t n8 = n / 8;
  SDMI_LAUNCH(range_scan_kernel, dim3((unsigned)((n8 + 1 + 255) / 256)), dim3(256), 0, stream, p, n8, n, g_range.dev);
  SDMI_HIP_OK(hipGetLastError());
  unsigned long long h[3];
  SDMI_HIP_OK(hipMemcpyAsync(h, g_range.dev, sizeof h, hipMemcpyDeviceToHost, stream));
  SDMI_HIP_OK(hipStreamSynchronize(stream));
  g_range.total[0] += h[0];
  g_range.total[1] += h[1];
  if (h[2] > g_range.total[2]) g_range.total[2] = h[2];
  if ((h[0] || h[1]) && g_range.first.empty()) g_range.first = what;
  return 0;
}

int range_report(std::string* json) {
  std::lock_guard<std::mutex> lk(g_range.mu);
  const unsigned bits = (unsigned)g_range.total[2];
  float mx;
  memcpy(&mx, &bits, sizeof mx);
  char buf[512];
  snprintf(buf, sizeof buf, "{\"over_6e4\": %llu, \"nonfinite\": %llu, \"max_abs\": %.6g, \"first\": \"%s\"}", g_range.total[0],
           g_range.total[1], (double)mx, g_range.first.c_str());
  *json = buf;
  return 0;
}

}  // namespace sdmi
