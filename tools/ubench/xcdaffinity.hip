// Does data written by one kernel stay readable from the SAME XCD's L2 by the next kernel?  (measurement tool)
// Kernel W: workgroup b writes region (b + shift) % nblocks.  Kernel R: workgroup b (one wave) chases dependent loads, a new 128-byte line per hop, through
// region b.  shift 0 -> the reader runs on the writer's XCD (workgroups are dealt round-robin: b % 8), shift 1 -> another
// XCD, shift 8 -> the same XCD, another CU.   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/xcdaffinity.bin tools/ubench/xcdaffinity.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int REGION = 16384;      // ints per region (64 KB)

__global__ void writer(int* buf, int shift, int salt) {
  int* r = buf + (size_t)((blockIdx.x + shift) % gridDim.x) * REGION;
  for (int i = threadIdx.x; i < REGION; i += blockDim.x) r[i] = (i + 32 * 97 + (salt & 1) * 32 * 2) & (REGION - 1);     // next hop: another 128-byte line
}
__global__ void reader(const int* buf, int* out, int hops) {
  const int* r = buf + (size_t)blockIdx.x * REGION;
  int idx = threadIdx.x;                       // (64 lanes, adjacent ints: one line per hop and wave)
  for (int h = 0; h < hops; ++h) idx = r[idx & (REGION - 1)];
  if (idx == 0x7fffffff) out[0] = idx;
}

int main() {
  const int blocks = 256;
  int *buf, *out; hipMalloc(&buf, (size_t)blocks * REGION * 4); hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int hops : {0, 64, 256}) {
    for (int shift : {0, 8, 1, 4}) {
      float tot = 0;
      const int reps = 50;
      for (int i = 0; i < reps + 5; ++i) {
        hipLaunchKernelGGL(writer, dim3(blocks), dim3(256), 0, 0, buf, shift, i);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(reader, dim3(blocks), dim3(64), 0, 0, buf, out, hops);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i >= 5) tot += ms;
      }
      printf("hops %2d  writer shift %d (%s): reader %6.2f us\n", hops, shift,
             shift == 0 ? "same workgroup id: same XCD" : (shift % 8 == 0 ? "same XCD, other workgroup" : "another XCD"), tot * 1e3 / reps);
    }
  }
  return 0;
}
