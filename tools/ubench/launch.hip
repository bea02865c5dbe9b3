// Per-launch cost of back-to-back kernels in one stream (measurement tool): what a UNet call of ~420 dependent launches
// pays before any work is done.   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/launch.bin tools/ubench/launch.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void empty_kernel(float* p) { if (p == nullptr) __builtin_trap(); }
__global__ void touch_kernel(const float* in, float* out, int n_per_block) {      // streams n_per_block floats per block
  const float* s = in + (size_t)blockIdx.x * n_per_block;
  float* d = out + (size_t)blockIdx.x * n_per_block;
  for (int i = threadIdx.x; i < n_per_block; i += blockDim.x) d[i] = s[i] + 1.0f;
}

template <class F>
static double per_launch_us(int n, F launch) {
  for (int i = 0; i < 20; ++i) launch(i);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) launch(i);
  hipEventRecord(e1, 0); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / n;
}

int main() {
  float *a, *b; hipMalloc(&a, 64 << 20); hipMalloc(&b, 64 << 20); hipMemset(a, 0, 64 << 20); hipMemset(b, 0, 64 << 20);
  const int N = 2000;
  printf("empty kernel,    1 workgroup  x 64 threads : %6.2f us per launch\n", per_launch_us(N, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, a); }));
  printf("empty kernel,  256 workgroups x 256 threads: %6.2f us per launch\n", per_launch_us(N, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, a); }));
  printf("empty kernel, 2048 workgroups x 256 threads: %6.2f us per launch\n", per_launch_us(N, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(2048), dim3(256), 0, 0, a); }));
  for (int kb : {4, 64, 1024}) {          // dependent chain: each launch reads what the previous one wrote (ping-pong)
    const int blocks = 256, n_per_block = kb * 1024 / 4 / 1;      // kb KB per workgroup
    if ((size_t)blocks * n_per_block * 4 > ((size_t)64 << 20)) continue;
    const double us = per_launch_us(N, [&](int i) {
      hipLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n_per_block);
    });
    printf("dependent copy chain, 256 workgroups x %4d KB (%6.1f MB per launch): %6.2f us per launch\n", kb, blocks * kb / 1024.0, us);
  }
  return 0;
}
