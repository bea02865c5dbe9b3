// Where does the dispatcher put the workgroups of a grid that is smaller than the chip's capacity?  (measurement tool)
// Every workgroup records its XCC / SE / SH / CU ids (s_getreg HW_ID, XCC_ID) and spins ~20 us so that the whole grid is
// resident at once; the host prints how many CUs hold 0 / 1 / 2 / 3+ workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/placement.bin tools/ubench/placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(256) where_kernel(unsigned* out, long long spin) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    smem[0] = 1;
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
}

int main() {
  unsigned* d; hipMalloc(&d, 8192 * 8);
  for (int lds_kb : {48, 64, 100}) {
    hipFuncSetAttribute((const void*)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    for (int blocks : {160, 320, 640}) {
      hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), lds_kb * 1024, 0, d, 2000LL);     // 20 us at 100 MHz
      hipDeviceSynchronize();
      std::vector<unsigned> h(2 * blocks); hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
      std::map<unsigned, int> per_cu; std::map<unsigned, int> per_xcc;
      for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++; per_xcc[xcc]++;
      }
      int hist[8] = {0};
      for (auto& kv : per_cu) hist[kv.second < 7 ? kv.second : 7]++;
      printf("LDS %3d KB / workgroup, %3d workgroups: %3zu distinct CUs;  CUs holding 1: %3d  2: %3d  3: %3d  4+: %3d;  per XCC:", lds_kb, blocks,
             per_cu.size(), hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
      for (auto& kv : per_xcc) printf(" %d", kv.second);
      printf("\n");
    }
  }
  return 0;
}
