// Weight-streaming microbenchmark (measurement tool): how fast can 240 workgroups pull a [N = 1280][K = 11520] fp16
// weight matrix (29.5 MB, first touch: 16 different matrices = 472 MB are cycled so neither L2 nor the 256 MB Infinity
// Cache holds them) in the GEMM's access pattern -- k-tiles of 64 rows x 128 bytes --
//   rows:   row-major [N][K] as packed today: the 64 rows of a k-tile are 128-byte pieces 23 KB apart
//   panel:  [N / 64][K / 64][64][64]: the k-tiles of a workgroup's panel are consecutive 8 KB blocks
// with DEPTH k-tiles in flight per workgroup.   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/wstream.bin tools/ubench/wstream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, bool PANEL>
__global__ void __launch_bounds__(256) stream_kernel(const unsigned char* w, unsigned* sink, int N, int K, int splits) {
  const int panel = blockIdx.x / splits, split = blockIdx.x % splits;
  const int nkt = K / 64, per = nkt / splits;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // a wave instruction covers 8 rows x 128 B; 4 waves x 2 instructions = 64 rows
  u32x4 acc = {0, 0, 0, 0};
  u32x4 buf[DEPTH][2];
  auto addr = [&](int kt, int j) -> const u32x4* {
    const int row = wave * 16 + j * 8 + (lane >> 3), chunk = lane & 7;
    if (PANEL) return (const u32x4*)(w + (((size_t)panel * nkt + kt) * 64 + row) * 128 + chunk * 16);
    return (const u32x4*)(w + ((size_t)(panel * 64 + row) * K + kt * 64) * 2 + chunk * 16);
  };
  const int kt0 = split * per;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int j = 0; j < 2; ++j) buf[d][j] = __builtin_nontemporal_load(addr(kt0 + d, j));
  for (int t = 0; t < per; t += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc ^= buf[d][j];
        const int nk = t + d + DEPTH;
        if (nk < per) buf[d][j] = __builtin_nontemporal_load(addr(kt0 + nk, j));
      }
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int DEPTH, bool PANEL>
static void run(const std::vector<unsigned char*>& bufs, unsigned* sink, int N, int K, int splits) {
  const int blocks = N / 64 * splits;
  hipLaunchKernelGGL((stream_kernel<DEPTH, PANEL>), dim3(blocks), dim3(256), 0, 0, bufs[0], sink, N, K, splits);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  for (size_t i = 0; i < bufs.size(); ++i)
    hipLaunchKernelGGL((stream_kernel<DEPTH, PANEL>), dim3(blocks), dim3(256), 0, 0, bufs[i], sink, N, K, splits);
  hipEventRecord(e1, 0); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / bufs.size(), gb = (double)N * K * 2 / 1e9;
  printf("%-6s depth %2d splits %2d blocks %4d : %6.1f us per matrix = %5.2f TB/s\n", PANEL ? "panel" : "rows", DEPTH, splits, blocks, us, gb / us * 1e3);
}

int main() {
  const int N = 1280, K = 11520;
  std::vector<unsigned char*> bufs(16);
  for (auto& b : bufs) { hipMalloc(&b, (size_t)N * K * 2); hipMemset(b, 1, (size_t)N * K * 2); }
  unsigned* sink; hipMalloc(&sink, 64);
  for (int splits : {6, 12, 36}) {
    run<2, false>(bufs, sink, N, K, splits); run<2, true>(bufs, sink, N, K, splits);
    run<5, false>(bufs, sink, N, K, splits); run<5, true>(bufs, sink, N, K, splits);
    run<10, false>(bufs, sink, N, K, splits); run<10, true>(bufs, sink, N, K, splits);
  }
  return 0;
}
