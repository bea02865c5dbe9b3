// Per-seam cost of an N-SPLIT ROW-STRIP CLUSTER: G partner workgroups own one 32-row strip and split the columns of every
// GEMM of a chain; at each GEMM -> GEMM seam they exchange their 32 x N/G fp16 slices so that everybody holds the whole 32 x N operand strip
// of the next GEMM.  This measures exactly that exchange on an MI355X, in the geometry the chain kernels would have: 256 workgroups (one per
// CU), the G partners of a strip on one XCD (blockIdx % 8), a strip of 32 x 640 fp16 (40 KB, G = 4) or 32 x 1280 (80 KB, G = 8 / 16),
//   publish : every wave stores its share of the slice with 16-byte write-through (sc1) stores, drains, one lane stores the epoch flag
//             (the R1 recipe of cdna_hip_programming.md Guideline 16)
//   consume : one wave polls the G flags (relaxed agent loads + s_sleep), ONE agent acquire, then all waves copy the strip into LDS
// (timed with wall_clock64(): the constant-rate counter, hipDeviceAttributeWallClockRate)
// with and without a weight stream between the seams (the chain kernels stream ~10 units of 20 KB per GEMM: "load" = that many KB of LDS-DMA
// reads of a 64 MB buffer per workgroup and round).  Reported: cycles (s_memtime, 100 MHz) from publish start to the strip being in LDS.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/seam.bin tools/ubench/seam.hip && tools/ubench/seam.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 384;                     // 6 waves: the chain kernels' 5 compute waves + a loader
constexpr int ROUNDS = 48;

__global__ void __launch_bounds__(NT) seam_kernel(unsigned char* xbuf, unsigned* flags, const unsigned char* weights, long long* times, int* err,
                                                  int G, int strip_bytes, int load_kb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [strip_bytes] + 32 KB of stream landing zone
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;                // slot 0 .. 31 inside the XCD
  const int cluster = xcd * (32 / G) + slot / G, g = slot % G;
  const int slice = strip_bytes / G;
  unsigned char* const cbase = xbuf + (size_t)cluster * 2 * strip_bytes;     // [parity][G][slice]
  unsigned* const cflag = flags + (size_t)cluster * 2 * 32;                  // [parity][G] (padded to 32 words per parity)
  constexpr int OOB = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(weights + (size_t)(bid & 63) * (1 << 20)), 0, OOB, 0x00020000);
  long long t_acc = 0, t_max = 0;
  for (int r = 0; r < ROUNDS; ++r) {
    const int par = r & 1;
    const unsigned epoch = (unsigned)r + 1;
    // ---- the work between two seams: a weight stream through LDS-DMA (every wave issues, nobody reads: only the traffic matters) ----
    for (int kb = wave; kb < load_kb; kb += NT / 64)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(smem + strip_bytes + (kb & 31) * 1024), 16, lane * 16,
                                               ((r * load_kb + kb) & 1023) * 1024, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t0 = (long long)wall_clock64();
    // ---- publish my slice (values = f(r, g, offset) so that the consumer can check every word) ----
    {
      const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(cbase + (size_t)par * strip_bytes + (size_t)g * slice), 0, slice, 0x00020000);
      for (int o = tid * 16; o < slice; o += NT * 16) {
        const unsigned v = (unsigned)(r * 131 + g * 17) + (unsigned)o;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{v, v + 1, v + 2, v + 3}, rs_x, o, 0, 16);      // aux 16 = sc1 (write-through)
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // EVERY storing wave drains
    __syncthreads();
    if (tid == 0) __hip_atomic_store(cflag + par * 32 + g, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- consume: wave 0 polls the G flags, one acquire, then everybody copies the strip into LDS ----
    if (wave == 0) {
      int budget = 1 << 18;
      bool ok = false;
      while (!ok && --budget > 0) {
        const unsigned f = lane < G ? __hip_atomic_load(cflag + par * 32 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
        ok = __all(f == epoch);
        if (!ok) __builtin_amdgcn_s_sleep(1);
      }
      if (!ok && lane == 0) *err = 1;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    {
      const u32x4* src = (const u32x4*)(cbase + (size_t)par * strip_bytes);
      u32x4 v[8];
      for (int base = 0; base < strip_bytes; base += NT * 16 * 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int o = base + (j * NT + tid) * 16;
          v[j] = o < strip_bytes ? src[o / 16] : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int o = base + (j * NT + tid) * 16;
          if (o < strip_bytes) {
            *(u32x4*)(smem + o) = v[j];
            const int gg = o / slice, oo = o - gg * slice;
            if (v[j][0] != (unsigned)(r * 131 + gg * 17) + (unsigned)oo || v[j][3] != v[j][0] + 3) *err = 2;     // every word checked
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const long long dt = (long long)wall_clock64() - t0;
    if (r >= 8) { t_acc += dt; t_max = dt > t_max ? dt : t_max; }
  }
  if (tid == 0) { times[2 * bid] = t_acc / (ROUNDS - 8); times[2 * bid + 1] = t_max; }
}

int main() {
  unsigned char *xbuf, *weights; unsigned* flags; long long* times; int* err;
  const size_t xbytes = (size_t)256 * 2 * 81920;
  CK(hipMalloc(&xbuf, xbytes)); CK(hipMalloc(&flags, 256 * 2 * 32 * 4)); CK(hipMalloc(&weights, (size_t)65 << 20));
  CK(hipMalloc(&times, 256 * 2 * 8)); CK(hipMalloc(&err, 4));
  CK(hipMemset(weights, 1, (size_t)65 << 20));
  CK(hipFuncSetAttribute((const void*)seam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 81920 + 32768));
  int clk_khz = 0;
  CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
  printf("seam exchange, 256 workgroups x %d threads, %d timed rounds; constant clock %d kHz\n", NT, ROUNDS - 8, clk_khz);
  struct Cfg { int G, strip, load; };
  const Cfg cfgs[] = {{4, 40960, 0}, {4, 40960, 200}, {8, 81920, 0}, {8, 81920, 200}, {16, 81920, 0}, {16, 81920, 200}, {2, 40960, 200}, {1, 40960, 200}};
  for (const Cfg& c : cfgs) {
    std::vector<double> med;
    double worst = 0;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(flags, 0, 256 * 2 * 32 * 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(xbuf, 0xff, xbytes));
      hipLaunchKernelGGL(seam_kernel, dim3(256), dim3(NT), c.strip + 32768, 0, xbuf, flags, weights, times, err, c.G, c.strip, c.load);
      CK(hipDeviceSynchronize());
      long long h[512]; int e;
      CK(hipMemcpy(h, times, sizeof h, hipMemcpyDeviceToHost)); CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
      if (e) { printf("  G=%d: ERROR %d (1 = a poll ran out, 2 = stale data)\n", c.G, e); break; }
      std::vector<long long> a;
      for (int b = 0; b < 256; ++b) { a.push_back(h[2 * b]); worst = std::max(worst, (double)h[2 * b + 1]); }
      std::sort(a.begin(), a.end());
      med.push_back((double)a[128]);
    }
    if (med.empty()) continue;
    std::sort(med.begin(), med.end());
    const double us = med[med.size() / 2] / (clk_khz * 1e-3);
    printf("  G = %2d partners, strip %2d KB (slice %5d B), %3d KB of weight stream between seams:  %.2f us per seam (median workgroup, mean over rounds); worst single round %.2f us\n",
           c.G, c.strip / 1024, c.strip / c.G, c.load, us, worst / (clk_khz * 1e-3));
  }
  return 0;
}
