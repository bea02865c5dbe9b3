// L2 -> LDS streaming rate per CU (measurement tool): every workgroup (256 threads) pulls 16 KB "k-tiles" from its own
// L2-resident region by MUBUF LDS-DMA, DEPTH tiles in flight, no barriers, no LDS reads -- the ceiling of the GEMM's
// operand path as a function of bytes in flight and of workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/l2stream.bin tools/ubench/l2stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N>
__device__ __forceinline__ void wait_vm() {
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if (N == 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int DEPTH>
__global__ void __launch_bounds__(256) l2stream_kernel(const unsigned char* src, int region_bytes, int tiles, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * region_bytes), 0, region_bytes, 0x00020000);
  const int tiles_in_region = region_bytes / 16384;
  auto issue = [&](int t) {
    const int stage = t % DEPTH, rt = t % tiles_in_region;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      auto dst = (__attribute__((address_space(3))) void*)(smem + stage * 16384 + (wave * 4 + j) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, rt * 16384 + (wave * 4 + j) * 1024 + lane * 16, 0, 0, 0);
    }
  };
  for (int t = 0; t < DEPTH - 1; ++t) issue(t);
  for (int t = 0; t < tiles; ++t) {
    issue(t + DEPTH - 1);
    wait_vm<4 * (DEPTH - 1)>();
  }
  wait_vm<0>();
  __syncthreads();
  if (((unsigned*)smem)[tid] == 0x12345678u) sink[0] = 1;
}

template <int DEPTH>
static void run(const unsigned char* src, unsigned* sink, int blocks, int region) {
  const int tiles = 2048;
  hipFuncSetAttribute((const void*)l2stream_kernel<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 16384);
  hipLaunchKernelGGL((l2stream_kernel<DEPTH>), dim3(blocks), dim3(256), DEPTH * 16384, 0, src, region, 64, sink);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((l2stream_kernel<DEPTH>), dim3(blocks), dim3(256), DEPTH * 16384, 0, src, region, tiles, sink);
  hipEventRecord(e1, 0); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * (tiles + DEPTH - 1) * 16384;
  printf("depth %d (%3d KB in flight / workgroup) workgroups %4d (%.1f per CU) region %4d KB : %6.2f TB/s total, %6.1f GB/s per CU, %5.1f GB/s per workgroup\n",
         DEPTH, (DEPTH - 1) * 16, blocks, blocks / 256.0, region / 1024, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / 256, bytes / (ms * 1e-3) / 1e9 / blocks);
}

int main() {
  unsigned char* src; hipMalloc(&src, (size_t)64 << 20); hipMemset(src, 1, (size_t)64 << 20);
  unsigned* sink; hipMalloc(&sink, 64);
  for (int region : {64 << 10, 16 << 10}) {      // per-workgroup region: 64 KB (L2 hits) / 16 KB (one tile: L1 + L2 hits)
    for (int blocks : {256, 512, 768}) {
      run<2>(src, sink, blocks, region);
      run<3>(src, sink, blocks, region);
      if (blocks <= 512) run<4>(src, sink, blocks, region);
      if (blocks <= 256) { run<6>(src, sink, blocks, region); run<8>(src, sink, blocks, region); }
    }
  }
  return 0;
}
